#!/bin/bash
# round 4, sixth GPU visit: column backward of the fused pooling, BatchNorm reductions (third cut), train step
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sparse_bn.py tests/test_gpu_fused_columns.py tests/test_gpu_bev_pool.py tests/test_gpu_modules.py -x -q > gpurun_out/r4f_tests.log 2>&1
echo "== tests rc=$?"; tail -8 gpurun_out/r4f_tests.log | cut -c1-400
for mode in "--amp" ""; do
  timeout 400 python bench.py --mode train-step --no-cpu-baseline $mode > gpurun_out/r4f_train${mode}.log 2>&1
  echo "== train $mode rc=$?"; tail -1 gpurun_out/r4f_train${mode}.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['config']['stage_ms'].items()}, d['roofline']['frac'])" || tail -15 gpurun_out/r4f_train${mode}.log
done
rm -rf gpurun_out/prof_r4f
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4f -o b -- python $GRAFT_REPO_ROOT/bench.py --mode train-step --amp --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r4f_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_r4f > gpurun_out/r4f_train_amp_kernel_trace_stats.txt 2>&1
head -22 gpurun_out/r4f_train_amp_kernel_trace_stats.txt | cut -c1-150
find gpurun_out/prof_r4f -name "*.db" -delete
