for ov in auto ahead; do python bench.py --no-cpu-baseline --no-extras --batch 1 --overlap $ov 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('batch 1 overlap $ov'.ljust(30), round(d['ms_per_step'], 3), round(d['value'],1))"; done
python bench.py --no-cpu-baseline --no-extras --batch 2 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('batch 2 auto', round(d['ms_per_step'],3), d['config']['overlap'][:40])"
