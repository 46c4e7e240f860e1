#!/bin/bash
# HBM traffic of the bev_pool kernels from rocprofv3 PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (kernel-trace only, no other trace domain), over the default bench command.
tag=${1:-x}
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_bev_$tag
rm -rf $out; mkdir -p $out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $c -d $out/$c -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/${c}_run.log 2>&1
  echo "$c rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("$out/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bev_pool" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$out/*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "bev_pool" in r["Kernel_Name"]:
            dur[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in acc.items():
    print("==", k, "n=", {c: len(v) for c, v in cs.items()}, "avg_us=%.1f" % (sum(dur[k]) / max(len(dur[k]), 1)))
    for c, v in cs.items():
        print("   %-12s mean per dispatch = %.1f KB" % (c, sum(v) / len(v)))
PY
find $out -name "*agent_info.csv" -delete
