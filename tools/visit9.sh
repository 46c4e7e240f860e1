#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_tf32
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_tf32 -o b -- python $R/bench.py --mode train-step --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_tf32_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_tf32 > gpurun_out/r05_train_step_fp32_kernel_trace_stats.txt 2>&1
head -30 gpurun_out/r05_train_step_fp32_kernel_trace_stats.txt | cut -c1-150
find gpurun_out -name "*.db" -delete
