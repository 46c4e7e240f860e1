#!/bin/bash
# Raw PMC means per kernel:  tools/pmc_raw.sh <tag> <kernel-name regex> "<counters pass 1>;<counters pass 2>;..." -- <command>
tag=$1; filt=$2; groups=$3; shift 4
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcr_$tag
rm -rf $out; mkdir -p $out
i=0
IFS=';' read -ra GRPS <<< "$groups"
for grp in "${GRPS[@]}"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $out/g$i -o r -- "$@" > $out/g${i}_run.log 2>&1)
done
cd $GRAFT_REPO_ROOT
python - "$out" "$filt" <<'PY' | tee gpurun_out/pmcr_${tag}.txt
import csv, glob, os, re, sys
from collections import defaultdict
d, filt = sys.argv[1], re.compile(sys.argv[2])
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
def short(n):
    n = n.split("(")[0]
    for p in ("void ", "bevamd::", "slab::", "tile::"): n = n.replace(p, "")
    return n.strip()
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if filt.search(k): acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if filt.search(k): dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in acc:
    print(f"== {k}  dispatches/pass ~{max(len(v) for v in acc[k].values())}  us(mean under counters) {sum(dur[k]) / max(len(dur[k]), 1):.1f}")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:34s} {sum(v) / len(v):16.0f}")
PY
find $out -name "*agent_info.csv" -delete
