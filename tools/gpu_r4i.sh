#!/bin/bash
# round 4, ninth GPU visit: compact slot metadata for the narrow-row kernels (level 1), pipelined output-side tables
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_keyorder.py tests/test_gpu_flagship_oracle.py tests/test_gpu_spconv_fused.py tests/test_gpu_spconv_slab.py -x -q > gpurun_out/r4i_tests.log 2>&1
echo "== tests rc=$?"; tail -5 gpurun_out/r4i_tests.log | cut -c1-300
bash tools/bench_pair.sh "BEVAMD_SPCONV_SLAB_COMPACT=0" --no-extras
bash tools/bench_pair.sh "BEVAMD_SPCONV_SLAB_COMPACT=1" --no-extras
bash tools/bench_pair.sh "BEVAMD_SPCONV_SLAB_COMPACT=0" --no-extras
bash tools/bench_pair.sh "BEVAMD_SPCONV_SLAB_COMPACT=1" --no-extras
bash tools/gpu_timeline.sh r4i --no-extras
cp gpurun_out/timeline_r4i.txt gpurun_out/r4i_timeline.txt
