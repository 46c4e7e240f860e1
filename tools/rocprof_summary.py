"""Summarise a rocprofv3 run directory (rocpd .db and/or csv) as a per-kernel table.
    python tools/rocprof_summary.py <dir> [pmc]   ->  text on stdout (copy into profiles/)"""
import csv
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def from_db(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) "
                      "from kernels group by name order by sum(end-start) desc").fetchall()
    return [(r[0], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e6) for r in rows]


def from_csv(path):
    acc = defaultdict(list)
    with open(path) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name") or row.get("Name")
            s, e = row.get("Start_Timestamp"), row.get("End_Timestamp")
            if name and s and e:
                acc[name].append((int(e) - int(s)) / 1e3)
    out = [(k, len(v), sum(v) / len(v), min(v), max(v), sum(v) / 1e3) for k, v in acc.items()]
    return sorted(out, key=lambda r: -r[5])


def pmc_from_csv(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path) as fh:
        for row in csv.DictReader(fh):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    d = sys.argv[1]
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    kcsv = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = from_db(dbs[0]) if dbs else (from_csv(kcsv[0]) if kcsv else [])
    tot = sum(r[5] for r in rows) or 1.0
    print(f"{'kernel':78s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for r in rows:
        print(f"{r[0][:78]:78s} {r[1]:6d} {r[2]:10.2f} {r[3]:10.2f} {r[4]:10.2f} {r[5]:10.3f} {100 * r[5] / tot:6.2f}")
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        print(f"\n# PMC ({os.path.relpath(f, d)}): mean counter value per dispatch")
        for k, cs in pmc_from_csv(f).items():
            for c, v in cs.items():
                print(f"{k[:78]:78s} {c:16s} n={len(v):4d} mean={sum(v) / len(v):14.1f}")


if __name__ == "__main__":
    main()
