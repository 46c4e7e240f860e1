export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export BEVAMD_BENCH_BEVPOOL_AT=64
rm -rf gpurun_out/prof_x
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_x -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_x_run.log 2>&1)
cd tools && python step_window.py ../gpurun_out/prof_x 8 10.5 > ../gpurun_out/y18_window.txt 2>&1; cd ..
find gpurun_out -name "*.db" -delete
