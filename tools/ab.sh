#!/bin/bash
# A/B of environment settings on ONE box: tools/ab.sh "<bench args>" "VAR=a" "VAR=b" ...  (each setting twice, interleaved)
ARGS="$1"; shift
for rep in 1 2; do
  for setting in "$@"; do
    env $setting python bench.py --no-cpu-baseline --no-extras $ARGS 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$setting'.ljust(44), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['config']['stage_ms'].items()})"
  done
done
