#!/bin/bash
# round 4, seventh GPU visit: wide filter-gradient variant A/B, column backward second cut, train step + trace
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_ddp.py tests/test_gpu_fused_columns.py tests/test_gpu_shims.py -x -q > gpurun_out/r4g_tests.log 2>&1
echo "== tests rc=$?"; tail -5 gpurun_out/r4g_tests.log | cut -c1-400
for wide in 1 0; do
  BEVAMD_SPCONV_WGRAD_WIDE=$wide timeout 400 python bench.py --mode train-step --no-cpu-baseline --amp > gpurun_out/r4g_train_wide${wide}.log 2>&1
  echo "== train --amp wide=$wide rc=$?"; tail -1 gpurun_out/r4g_train_wide${wide}.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['config']['stage_ms'].items()}, d['roofline']['frac'])" || tail -15 gpurun_out/r4g_train_wide${wide}.log
done
rm -rf gpurun_out/prof_r4g
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4g -o b -- python $GRAFT_REPO_ROOT/bench.py --mode train-step --amp --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r4g_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_r4g > gpurun_out/r4g_train_amp_kernel_trace_stats.txt 2>&1
head -24 gpurun_out/r4g_train_amp_kernel_trace_stats.txt | cut -c1-150
find gpurun_out/prof_r4g -name "*.db" -delete
