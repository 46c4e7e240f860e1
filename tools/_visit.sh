for rep in 1 2 3; do
for s in BEVAMD_BENCH_BEVPOOL_ALONE=0 BEVAMD_BENCH_BEVPOOL_ALONE=1; do
env $s python bench.py --no-cpu-baseline --no-extras --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$s'.ljust(36), round(d['ms_per_step'], 3), {k: round(v, 3) for k, v in d['config']['stage_ms'].items()}, round(d['roofline']['frac_in_step'],3))"
done; done
