#!/bin/bash
# one GPU visit of the development loop: [tests] [bench] [extras given as arguments are run verbatim at the end]
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
if [ "${TESTS:-all}" != "none" ]; then
  if [ "${TESTS:-all}" = "all" ]; then SEL="tests"; else SEL="$TESTS"; fi
  timeout 1200 python -m pytest $SEL -m gpu -x -q > gpurun_out/v_tests.log 2>&1
  echo "== tests rc=$?"; grep -E "^E |FAILED|passed|failed|Error" gpurun_out/v_tests.log | tail -6 | cut -c1-300
fi
for i in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline ${BENCH_ARGS:---no-extras} > gpurun_out/v_bench_$i.log 2>gpurun_out/v_bench_$i.err
  grep "^{" gpurun_out/v_bench_$i.log | tail -1 > gpurun_out/v_bench_line_$i.json
  python - <<PY
import json
d = json.load(open('gpurun_out/v_bench_line_$i.json'))
print('bench $i', round(d['value'], 1), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['config'].get('stage_ms', {}).items()}, round(d['roofline']['frac'], 3))
PY
done
timeout 300 python bench.py --no-cpu-baseline --no-extras --batch 1 > gpurun_out/v_bench_b1.log 2>&1
grep "^{" gpurun_out/v_bench_b1.log | tail -1 > gpurun_out/v_bench_line_b1.json
python - <<PY
import json
d = json.load(open('gpurun_out/v_bench_line_b1.json'))
print('batch 1', round(d['value'], 1), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['config'].get('stage_ms', {}).items()})
PY
if [ -n "$TIMELINE" ]; then
  rm -rf gpurun_out/prof_v gpurun_out/prof_vb1
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_v -o b -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_v_run.log 2>&1)
  python tools/graph_timeline.py gpurun_out/prof_v > gpurun_out/v_step_timeline.txt 2>&1
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_vb1 -o b -- python $R/bench.py --batch 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_vb1_run.log 2>&1)
  python tools/graph_timeline.py gpurun_out/prof_vb1 > gpurun_out/v_step_timeline_batch1.txt 2>&1
  head -3 gpurun_out/v_step_timeline.txt | tail -1; head -3 gpurun_out/v_step_timeline_batch1.txt | tail -1
fi
for cmd in "$@"; do
  echo "== $cmd"; bash -c "$cmd" 2>&1 | tail -${TAILN:-15}
done
find gpurun_out -name "*.db" -delete
