#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_observed.json
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/v6_tests.log 2>&1
echo "== tests rc=$?"; grep -E "^E |FAILED|passed|failed|Error" gpurun_out/v6_tests.log | tail -8 | cut -c1-300
timeout 900 python bench.py > gpurun_out/v6_bench.log 2>gpurun_out/v6_bench.err; echo "== bench rc=$?"
grep "^{" gpurun_out/v6_bench.log | tail -1 > gpurun_out/v6_bench_line.json; wc -c gpurun_out/v6_bench_line.json; cat gpurun_out/v6_bench_line.json; tail -3 gpurun_out/v6_bench.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('batch1', round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['config']['stage_ms'].items()}, d['config']['overlap'][:40])"
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
