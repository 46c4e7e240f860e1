#!/bin/bash
# One GPU-box visit: GPU test suite, bench line, rocprofv3 kernel-trace summary (all outputs under gpurun_out/).
#   tools/gpu_round.sh <tag> [pytest-args...]
tag=${1:-x}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/tests_$tag.log 2>&1
echo "== tests rc=$?"; tail -5 gpurun_out/tests_$tag.log
timeout 600 python bench.py > gpurun_out/bench_$tag.log 2>&1
echo "== bench rc=$?"; tail -3 gpurun_out/bench_$tag.log | cut -c1-2500
rm -rf gpurun_out/prof_$tag
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_$tag > gpurun_out/prof_${tag}_summary.txt 2>&1
head -45 gpurun_out/prof_${tag}_summary.txt | cut -c1-150
# keep the merge-back small
find gpurun_out/prof_$tag -name "*.db" -size +20M -delete
