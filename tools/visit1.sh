#!/bin/bash
# round 5, visit 1: changed tests, reference-GPU goldens, the compact bench line, schedule A/B, timeline of the new default
mkdir -p gpurun_out/golden
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fused_columns.py tests/test_gpu_voxelize.py tests/test_gpu_vtransform.py tests/test_gpu_keyorder.py -x -q > gpurun_out/v1_tests.log 2>&1
echo "== tests rc=$?"; grep -E "^E |FAILED|passed|failed|Error" gpurun_out/v1_tests.log | tail -6 | cut -c1-300
timeout 300 python tests/golden/make_voxel_gpu_golden.py gpurun_out/golden > gpurun_out/v1_golden_voxel.log 2>&1; echo "== voxel golden rc=$?"; tail -8 gpurun_out/v1_golden_voxel.log | cut -c1-250
timeout 300 python tests/golden/make_spconv_gpu_golden.py gpurun_out/golden > gpurun_out/v1_golden_spconv.log 2>&1; echo "== spconv golden rc=$?"; tail -10 gpurun_out/v1_golden_spconv.log | cut -c1-250
timeout 600 python bench.py > gpurun_out/v1_bench.log 2>gpurun_out/v1_bench.err; echo "== bench rc=$?"
grep "^{" gpurun_out/v1_bench.log | tail -1 > gpurun_out/v1_bench_line.json; wc -c gpurun_out/v1_bench_line.json; cat gpurun_out/v1_bench_line.json; tail -3 gpurun_out/v1_bench.err
for rep in 1 2; do
  for setting in "chain:1" "chain:0" "voxel:1" "head:1"; do
    ov=${setting%%:*}; gate=${setting##*:}
    BEVAMD_BENCH_CHAIN_GATE=$gate timeout 300 python bench.py --no-cpu-baseline --no-extras --overlap $ov 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$setting'.ljust(10), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['config']['stage_ms'].items()}, round(d['roofline']['frac'], 3), d['roofline'].get('kernel_ms_in_step'))"
  done
done
rm -rf gpurun_out/prof_v1
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v1 -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_v1_run.log 2>&1)
python tools/graph_timeline.py gpurun_out/prof_v1 > gpurun_out/v1_step_timeline.txt 2>&1
python tools/rocprof_summary.py gpurun_out/prof_v1 > gpurun_out/v1_kernel_trace_stats.txt 2>&1
head -40 gpurun_out/v1_step_timeline.txt | cut -c1-200
find gpurun_out -name "*.db" -delete
