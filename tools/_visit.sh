export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_v3
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v3 -o b -- python $R/bench.py --mode train-step --amp --steps 5 --warmup 2 --no-cpu-baseline --train-inputs static > $R/gpurun_out/prof_v3_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_v3 > gpurun_out/v3_train_trace.txt 2>&1
head -70 gpurun_out/v3_train_trace.txt | cut -c1-150
find gpurun_out -name "*.db" -size +30M -delete
