#!/bin/bash
# round 4, first GPU visit: new tests first (fast feedback), then the whole suite, the fused-pool timing, the bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fused_columns.py tests/test_gpu_keyorder.py -x -q > gpurun_out/r4a_new_tests.log 2>&1
echo "== new tests rc=$?"; tail -15 gpurun_out/r4a_new_tests.log
timeout 200 python tools/time_fused_pool.py 8 > gpurun_out/r4a_fused_time.log 2>&1; cat gpurun_out/r4a_fused_time.log | tail -8
timeout 200 python tools/time_fused_pool.py 1 > gpurun_out/r4a_fused_time1.log 2>&1; cat gpurun_out/r4a_fused_time1.log | tail -8
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4a_tests.log 2>&1
echo "== all tests rc=$?"; tail -5 gpurun_out/r4a_tests.log
timeout 600 python bench.py > gpurun_out/r4a_bench.log 2>&1
echo "== bench rc=$?"; tail -1 gpurun_out/r4a_bench.log | cut -c1-1800
