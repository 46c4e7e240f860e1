#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_spconv_f32x3.py -x -q > gpurun_out/v8_tests_a.log 2>&1
echo "== f32x3 tests rc=$?"; grep -E "^E |FAILED|passed|failed|Error" gpurun_out/v8_tests_a.log | tail -12 | cut -c1-300
timeout 1200 python -m pytest tests/test_gpu_spconv.py tests/test_gpu_modules.py tests/test_gpu_ddp.py tests/test_gpu_spconv_ext.py tests/test_gpu_shims.py -x -q > gpurun_out/v8_tests_b.log 2>&1
echo "== other tests rc=$?"; grep -E "^E |FAILED|passed|failed|Error" gpurun_out/v8_tests_b.log | tail -8 | cut -c1-300
for mode in 0 auto; do
  BEVAMD_SPCONV_F32X3=$mode timeout 400 python bench.py --mode train-step --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('fp32 train-step F32X3=$mode', round(d['ms_per_step'], 2), {k: round(v, 2) for k, v in d['config']['stage_ms'].items()})"
done
