#!/bin/bash
# short verification visit: whole GPU suite, the --amp train step with a kernel trace, the default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/v_tests.log 2>&1
echo "== tests rc=$?"; grep -E "^E |FAILED|passed|failed" gpurun_out/v_tests.log | tail -4 | cut -c1-300
rm -rf gpurun_out/prof_vt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_vt -o b -- python $R/bench.py --mode train-step --amp --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_vt_run.log 2>&1)
grep "^{" gpurun_out/prof_vt_run.log | tail -1 > gpurun_out/v_train_amp_traced.json
python tools/rocprof_summary.py gpurun_out/prof_vt > gpurun_out/v_train_step_amp_kernel_trace_stats.txt 2>&1
grep -E "sparse_bn|wgrad|bev_fused" gpurun_out/v_train_step_amp_kernel_trace_stats.txt | cut -c1-170
timeout 400 python bench.py --mode train-step --amp --no-cpu-baseline > gpurun_out/v_train_amp.log 2>&1
grep "^{" gpurun_out/v_train_amp.log | tail -1 > gpurun_out/v_train_amp.json
python -c "import json; d=json.load(open('gpurun_out/v_train_amp.json')); print('train --amp', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['config']['stage_ms'].items()})"
timeout 900 python bench.py > gpurun_out/v_bench.log 2>gpurun_out/v_bench.err
echo "== bench rc=$?"; grep "^{" gpurun_out/v_bench.log | tail -1 > gpurun_out/v_bench_line.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/v_bench_line.json'))
print(round(d['value'], 1), round(d['ms_per_step'], 3), d['config'].get('stage_ms'), d['roofline'])
print(json.dumps(d.get('extra'))[:900])
print(d['cpu_baseline'])
PY
find gpurun_out -name "*.db" -delete
