#!/bin/bash
# kernel trace of the bench + the timeline of one step (tools/graph_timeline.py); extra bench args / env pass through
mkdir -p gpurun_out; export TMPDIR=/tmp
tag=${1:-tl}; shift
rm -rf gpurun_out/prof_$tag
(cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_run.log 2>&1)
tail -1 gpurun_out/prof_${tag}_run.log | cut -c1-200
python tools/graph_timeline.py gpurun_out/prof_$tag > gpurun_out/timeline_$tag.txt 2>&1
head -5 gpurun_out/timeline_$tag.txt
find gpurun_out/prof_$tag -name "*.db" -delete
