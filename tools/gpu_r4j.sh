#!/bin/bash
# round 4, tenth GPU visit: whole suite after the ext-library split, per-row table kernel (third cut), step timeline
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4j_tests.log 2>&1
echo "== tests rc=$?"; tail -4 gpurun_out/r4j_tests.log | cut -c1-300
bash tools/bench_pair.sh "BEVAMD_SPCONV_DOWN_NBR=inputs" --no-extras
bash tools/bench_pair.sh "BEVAMD_SPCONV_DOWN_NBR=outputs" --no-extras
bash tools/bench_pair.sh "BEVAMD_SPCONV_DOWN_NBR=outputs" --no-extras
bash tools/gpu_timeline.sh r4j --no-extras
cp gpurun_out/timeline_r4j.txt gpurun_out/r4j_timeline.txt
grep -E "sp_nbr_rows|sp_slab_from|sp_mark|sp_rank" gpurun_out/r4j_timeline.txt | cut -c1-100
