#!/bin/bash
# tools/bench_pair.sh "<env assignments>" [bench args...]: one bench line (no CPU baseline) with the given environment; prints value / ms / stages
envs="$1"; shift
out=$(env $envs timeout 300 python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1)
echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$envs', round(d['value'],1), round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['config'].get('stage_ms',{}).items()}, {k: round(v,3) for k,v in d['config'].get('lidar_branch_eager_ms',{}).items()})" || echo "$out" | tail -5
