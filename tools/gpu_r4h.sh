#!/bin/bash
# round 4, eighth GPU visit: wide filter gradient (fixed), train step A/B + trace; inference schedule A/B (voxelizer beside the camera stages)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_ddp.py tests/test_gpu_shims.py tests/test_gpu_modules.py -x -q > gpurun_out/r4h_tests.log 2>&1
echo "== tests rc=$?"; tail -4 gpurun_out/r4h_tests.log | cut -c1-300
for wide in 1 0; do
  BEVAMD_SPCONV_WGRAD_WIDE=$wide timeout 400 python bench.py --mode train-step --no-cpu-baseline --amp > gpurun_out/r4h_train_wide${wide}.log 2>&1
  echo "== train --amp wide=$wide rc=$?"; tail -1 gpurun_out/r4h_train_wide${wide}.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['config']['stage_ms'].items()}, d['roofline']['frac'])" || tail -5 gpurun_out/r4h_train_wide${wide}.log
done
timeout 400 python bench.py --mode train-step --no-cpu-baseline > gpurun_out/r4h_train_fp32.log 2>&1
echo "== train fp32 rc=$?"; tail -1 gpurun_out/r4h_train_fp32.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['config']['stage_ms'].items()})"
rm -rf gpurun_out/prof_r4h
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r4h -o b -- python $GRAFT_REPO_ROOT/bench.py --mode train-step --amp --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r4h_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_r4h > gpurun_out/r4h_train_amp_kernel_trace_stats.txt 2>&1
head -20 gpurun_out/r4h_train_amp_kernel_trace_stats.txt | cut -c1-150
find gpurun_out/prof_r4h -name "*.db" -delete
bash tools/bench_pair.sh "X=1" --no-extras
bash tools/bench_pair.sh "X=1" --no-extras --overlap voxel
bash tools/bench_pair.sh "X=1" --no-extras
bash tools/bench_pair.sh "X=1" --no-extras --overlap voxel
