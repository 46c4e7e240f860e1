export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for s in BEVAMD_X=0 BEVAMD_BENCH_BEVPOOL_FIRST=1; do
env $s python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$s'.ljust(44), round(d['ms_per_step'], 3))"
done; done
export BEVAMD_BENCH_BEVPOOL_FIRST=1
rm -rf gpurun_out/prof_x
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_x_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_x 2>/dev/null | grep -i "by setting" | cut -c1-400
python tools/graph_timeline.py gpurun_out/prof_x > gpurun_out/y30_timeline.txt 2>&1
find gpurun_out -name "*.db" -delete
