#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
tag=${1:-r3c}
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/${tag}_all.log 2>&1
echo "== tests rc=$?"; tail -6 gpurun_out/${tag}_all.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench.log 2>&1
echo "== bench rc=$?"; tail -1 gpurun_out/${tag}_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('stage_ms'), d['config'].get('lidar_branch_eager_ms'))" || tail -20 gpurun_out/${tag}_bench.log
rm -rf gpurun_out/prof_$tag
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_$tag > gpurun_out/prof_${tag}_summary.txt 2>&1
head -34 gpurun_out/prof_${tag}_summary.txt | cut -c1-150
find gpurun_out/prof_$tag -name "*.db" -size +20M -delete
