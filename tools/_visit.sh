timeout 900 python bench.py > gpurun_out/x6_bench.log 2> gpurun_out/x6_bench.err; tail -1 gpurun_out/x6_bench.log > gpurun_out/x6_line.json; wc -c gpurun_out/x6_line.json; python -c "
import json
d = json.load(open('gpurun_out/x6_line.json'))
print(d['value'], d['ms_per_step'], d['config']['stage_ms'], d['roofline']['frac'], d['roofline']['frac_in_step']); print(d['extra']); print(d['config']['overlap'])"
tail -3 gpurun_out/x6_bench.err
timeout 300 python bench.py --batch 1 --no-cpu-baseline --no-extras | tail -1 | cut -c1-400
