#!/bin/bash
# round 3, visit A: the key-order tests, then bench lines in both voxel orders + kernel trace of the key-order step
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_keyorder.py -q --maxfail=30 -x -k "voxelizer or sorted_index" > gpurun_out/r3a_t1.log 2>&1
echo "== t1 rc=$?"; tail -15 gpurun_out/r3a_t1.log
timeout 900 python -m pytest tests/test_gpu_keyorder.py -q --maxfail=12 -k "metadata" > gpurun_out/r3a_t2.log 2>&1
echo "== t2 rc=$?"; tail -15 gpurun_out/r3a_t2.log
timeout 900 python -m pytest tests/test_gpu_keyorder.py -q --maxfail=12 -k "narrow" > gpurun_out/r3a_t3.log 2>&1
echo "== t3 rc=$?"; tail -25 gpurun_out/r3a_t3.log
timeout 900 python -m pytest tests/test_gpu_keyorder.py -q --maxfail=5 -k "encoder" > gpurun_out/r3a_t4.log 2>&1
echo "== t4 rc=$?"; tail -25 gpurun_out/r3a_t4.log
timeout 300 python bench.py --no-cpu-baseline --voxel-order first > gpurun_out/r3a_bench_first.log 2>&1
echo "== bench first rc=$?"; tail -1 gpurun_out/r3a_bench_first.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('stage_ms'), d['config'].get('lidar_branch_eager_ms'))"
timeout 300 python bench.py --no-cpu-baseline --voxel-order key > gpurun_out/r3a_bench_key.log 2>&1
echo "== bench key rc=$?"; tail -1 gpurun_out/r3a_bench_key.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('stage_ms'), d['config'].get('lidar_branch_eager_ms'))" || tail -20 gpurun_out/r3a_bench_key.log
rm -rf gpurun_out/prof_r3a
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3a -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r3a_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_r3a > gpurun_out/prof_r3a_summary.txt 2>&1
head -40 gpurun_out/prof_r3a_summary.txt | cut -c1-150
find gpurun_out/prof_r3a -name "*.db" -size +20M -delete
