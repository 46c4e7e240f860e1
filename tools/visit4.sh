#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_spconv_slab.py -x -q > gpurun_out/v4_tests.log 2>&1
echo "== tests rc=$?"; grep -E "^E |FAILED|passed|failed|Error" gpurun_out/v4_tests.log | tail -8 | cut -c1-300
timeout 600 python tools/time_slab_variant.py 64:1644222 64:1644228 64:1648128 64:1648118 64:1648228 128:1644220 128:1644228 128:1648128 128:1648148 2>&1 | tail -10
