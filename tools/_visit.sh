export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_spconv_fused.py -x -q -m gpu 2>&1 | tail -3
rm -rf gpurun_out/prof_x
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_x_run.log 2>&1)
python tools/rocprof_summary.py gpurun_out/prof_x 2>/dev/null | grep -i "dense_bev\|stream_kernel<1, 128" | cut -c1-150
grep "^{" gpurun_out/prof_x_run.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
python bench.py --batch 1 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b1', d['ms_per_step'])"
find gpurun_out -name "*.db" -delete
