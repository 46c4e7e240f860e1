#!/bin/bash
# round 4, eleventh GPU visit: whole suite (two-level BatchNorm reduction, fused rank apply+emit, row-based tilings, shims on the
# goldens' own gradients), train step, bench
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r4k_tests.log 2>&1
echo "== tests rc=$?"; grep -E "^E |FAILED|passed|failed" gpurun_out/r4k_tests.log | tail -8 | cut -c1-300
for mode in "--amp" ""; do
  timeout 400 python bench.py --mode train-step --no-cpu-baseline $mode > gpurun_out/r4k_train${mode}.log 2>&1
  echo "== train $mode rc=$?"; tail -1 gpurun_out/r4k_train${mode}.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['config']['stage_ms'].items()}, d['roofline']['frac'])" || tail -5 gpurun_out/r4k_train${mode}.log
done
bash tools/bench_pair.sh "X=1" --no-extras
bash tools/bench_pair.sh "X=1" --no-extras
