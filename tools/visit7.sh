#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fused_columns.py tests/test_gpu_bev_pool.py tests/test_gpu_sparse_bn.py tests/test_gpu_modules.py -x -q > gpurun_out/v7_tests.log 2>&1
echo "== tests rc=$?"; grep -E "^E |FAILED|passed|failed|Error" gpurun_out/v7_tests.log | tail -8 | cut -c1-300
timeout 300 python tools/time_fused_pool.py 8 2>&1 | tail -6
timeout 300 python tools/time_fused_pool.py 1 2>&1 | tail -6
# counters of the two passes
out=$R/gpurun_out/pmc_fused; rm -rf $out; mkdir -p $out
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $out/g$i -o r -- python $R/tools/time_fused_pool.py 8 > $out/g${i}_run.log 2>&1)
done
for d in $out/g*/; do python tools/rocprof_summary.py $d 2>/dev/null | grep -E "bev_fused_(cols|reduce)" ; done | cut -c1-200
find $out -name "*.db" -delete; find $out -name "*agent_info.csv" -delete
