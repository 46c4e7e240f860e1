#!/bin/bash
# round 4, fifth GPU visit: BatchNorm kernels (second cut), voxelizer-written encoder rows, train step A/B, bench, whole suite
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sparse_bn.py tests/test_gpu_keyorder.py tests/test_gpu_voxelize.py -x -q > gpurun_out/r4e_tests.log 2>&1
echo "== tests rc=$?"; tail -8 gpurun_out/r4e_tests.log | cut -c1-400
for nb in 1 0; do
  BEVAMD_NATIVE_BN=$nb timeout 400 python bench.py --mode train-step --no-cpu-baseline --amp > gpurun_out/r4e_train_nb${nb}.log 2>&1
  echo "== train --amp native_bn=$nb rc=$?"; tail -1 gpurun_out/r4e_train_nb${nb}.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['config']['stage_ms'].items()}, d['roofline']['frac'])" || tail -15 gpurun_out/r4e_train_nb${nb}.log
done
rm -rf gpurun_out/prof_r4e
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4e -o b -- python $GRAFT_REPO_ROOT/bench.py --mode train-step --amp --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r4e_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_r4e > gpurun_out/r4e_train_amp_kernel_trace_stats.txt 2>&1
head -16 gpurun_out/r4e_train_amp_kernel_trace_stats.txt | cut -c1-150
find gpurun_out/prof_r4e -name "*.db" -delete
bash tools/bench_pair.sh "X=1" --no-extras
bash tools/bench_pair.sh "X=1" --no-extras
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4e_all_tests.log 2>&1
echo "== all tests rc=$?"; tail -4 gpurun_out/r4e_all_tests.log | cut -c1-300
