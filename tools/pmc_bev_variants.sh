#!/bin/bash
# HBM bytes per bev_pool forward VARIANT: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (kernel trace only)
# over tools/sweep_bev_pool.py in its three-launches-per-variant mode; the template arguments in the kernel name tell the
# variants apart.  Usage (GPU box): tools/pmc_bev_variants.sh TAG [BATCHES] [VARIANTS]
tag=${1:-x}
export TMPDIR=/tmp BEVAMD_SWEEP_ONCE=1 BEVAMD_SWEEP_BATCHES=${2:-8} BEVAMD_SWEEP_VARIANTS=${3:-1,2,8,9,10,11}
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_bevvar_$tag
rm -rf $out; mkdir -p $out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $out/$c -o r -- python $GRAFT_REPO_ROOT/tools/sweep_bev_pool.py > $out/${c}_run.log 2>&1
  echo "$c rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bev_pool_fwd_cells" in r["Kernel_Name"]:
            acc[re.sub(r"\(.*", "", r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("(last 3 dispatches per kernel name = the largest batch of the sweep; FETCH_SIZE in KB, x2 on gfx950 for 16-byte lanes)")
for k in sorted(acc):
    cs = acc[k]
    print("==", k)
    for c, v in cs.items():
        tail = v[-3:]
        print("   %-12s n=%d  last3 mean = %.1f KB" % (c, len(v), sum(tail) / len(tail)))
PY
find $out -name "*agent_info.csv" -delete
