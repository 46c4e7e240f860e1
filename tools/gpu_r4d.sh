#!/bin/bash
# round 4, fourth GPU visit: native training BatchNorm — parity tests, train step with / without it, kernel trace of the --amp step
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sparse_bn.py tests/test_gpu_ddp.py tests/test_gpu_modules.py tests/test_gpu_spconv_fused.py -x -q > gpurun_out/r4d_tests.log 2>&1
echo "== tests rc=$?"; tail -12 gpurun_out/r4d_tests.log | cut -c1-400
for nb in 1 0; do
for mode in "--amp" ""; do
  BEVAMD_NATIVE_BN=$nb timeout 400 python bench.py --mode train-step --no-cpu-baseline $mode > gpurun_out/r4d_train_nb${nb}${mode}.log 2>&1
  echo "== train native_bn=$nb $mode rc=$?"; tail -1 gpurun_out/r4d_train_nb${nb}${mode}.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['config']['stage_ms'].items()}, d['roofline']['frac'])" || tail -15 gpurun_out/r4d_train_nb${nb}${mode}.log
done; done
rm -rf gpurun_out/prof_r4d
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4d -o b -- python $GRAFT_REPO_ROOT/bench.py --mode train-step --amp --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r4d_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_r4d > gpurun_out/r4d_train_amp_kernel_trace_stats.txt 2>&1
head -40 gpurun_out/r4d_train_amp_kernel_trace_stats.txt | cut -c1-150
find gpurun_out/prof_r4d -name "*.db" -delete
