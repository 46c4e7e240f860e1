#!/bin/bash
# round 4, third GPU visit: A/B of the strided-layer tables (inputs vs outputs side), depth-tile sizes of the column pooling, a step timeline
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused_columns.py tests/test_gpu_spconv_fused.py tests/test_gpu_flagship_oracle.py tests/test_gpu_flagship_batch8.py tests/test_gpu_spconv.py -x -q > gpurun_out/r4c_tests.log 2>&1
echo "== tests rc=$?"; tail -4 gpurun_out/r4c_tests.log | cut -c1-300
for dh in 60 40 32 20; do echo "DH=$dh"; BEVAMD_FUSED_COLS_DH=$dh timeout 200 python tools/time_fused_pool.py 8 2>&1 | grep "^columns:"; done
bash tools/bench_pair.sh "BEVAMD_SPCONV_DOWN_NBR=inputs" --no-extras
bash tools/bench_pair.sh "BEVAMD_SPCONV_DOWN_NBR=outputs" --no-extras
bash tools/bench_pair.sh "BEVAMD_SPCONV_DOWN_NBR=inputs" --no-extras
bash tools/bench_pair.sh "BEVAMD_SPCONV_DOWN_NBR=outputs" --no-extras
bash tools/gpu_timeline.sh r4c --no-extras
cp gpurun_out/timeline_r4c.txt gpurun_out/r4c_timeline.txt
