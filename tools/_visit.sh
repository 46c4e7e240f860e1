for rep in 1 2 3; do
for s in BEVAMD_X=0 BEVAMD_SPCONV_SLAB_VARIANTS=32:4100128; do
env $s python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$s'.ljust(44), round(d['ms_per_step'], 3))"
done; done
