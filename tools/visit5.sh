#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_spconv_slab.py tests/test_gpu_flagship_oracle.py tests/test_gpu_keyorder.py tests/test_gpu_flagship_batch8.py -x -q > gpurun_out/v5_tests.log 2>&1
echo "== tests rc=$?"; grep -E "^E |FAILED|passed|failed|Error" gpurun_out/v5_tests.log | tail -8 | cut -c1-300
timeout 600 python tools/time_slab_variant.py 64:1644222 64:1644228 2>&1 | tail -2
for rep in 1 2; do
  for ov in chain lidar head voxel; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --overlap $ov 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$ov'.ljust(10), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['config']['stage_ms'].items()}, round(d['roofline']['frac'], 3))"
  done
done
timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('batch1', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['config']['stage_ms'].items()})"
