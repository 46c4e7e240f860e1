timeout 900 python bench.py --no-cpu-baseline > gpurun_out/x11_bench.log 2> gpurun_out/x11_bench.err; tail -1 gpurun_out/x11_bench.log > gpurun_out/x11_line.json; wc -c gpurun_out/x11_line.json; python -c "
import json
d = json.load(open('gpurun_out/x11_line.json'))
print(d['value'], d['ms_per_step'], d['config']['stage_ms'], d['roofline']['frac'], d['roofline']['frac_in_step']); print(d['extra']); print(d['roofline_spconv'])"
tail -3 gpurun_out/x11_bench.err
