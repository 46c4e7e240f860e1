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


def solo_split(path, needle="bev_pool_fwd_cells_vec_kernel"):
    """The roofline kernel runs in two settings in one bench process (round 5): inside the timed step, where it shares the machine
    with the LiDAR branch, and SOLO right after the timed region (20 + 2 back-to-back launches: what roofline.kernel_ms reports).
    A launch is solo when no other kernel overlaps it in time.  -> (n_solo, avg_us_solo, n_shared, avg_us_shared) or None"""
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    mine = [(s, e) for n, s, e in rows if needle in n]
    if not mine:
        return None
    others = [(s, e) for n, s, e in rows if needle not in n]
    import bisect

    starts = [s for s, _ in others]
    # running maximum of the end times: overlap test against everything that started before
    maxend, m = [], 0
    for _, e in others:
        m = max(m, e)
        maxend.append(m)
    solo, shared = [], []
    for s, e in mine:
        i = bisect.bisect_left(starts, e)          # kernels that started before this one ended
        j = bisect.bisect_left(starts, s)          # ... of which these started before it did
        overl = (i > j) or (j > 0 and maxend[j - 1] > s)
        (shared if overl else solo).append((e - s) / 1e3)
    avg = lambda v: sum(v) / len(v) if v else float("nan")
    return len(solo), avg(solo), len(shared), avg(shared)


def main():
    d = sys.argv[1]
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    kcsv = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = from_db(dbs[0]) if dbs else (from_csv(kcsv[0]) if kcsv else [])
    tot = sum(r[5] for r in rows) or 1.0
    print(f"{'kernel':78s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for r in rows:
        print(f"{r[0][:78]:78s} {r[1]:6d} {r[2]:10.2f} {r[3]:10.2f} {r[4]:10.2f} {r[5]:10.3f} {100 * r[5] / tot:6.2f}")
    if dbs:
        sp = solo_split(dbs[0])
        # round 6: the roofline kernel's kernel-trace figures as a small JSON that bench.py quotes in its line
        # (roofline.rocprof_kernel_us) — python tools/rocprof_summary.py <dir> --roofline-json <out.json> <round>
        if "--roofline-json" in sys.argv and sp:
            import json

            i = sys.argv.index("--roofline-json")
            names = [r[0] for r in rows if "bev_pool_fwd_cells_vec_kernel" in r[0]]
            json.dump({"kernel": names[0] if names else None, "round": int(sys.argv[i + 2]) if len(sys.argv) > i + 2 else None,
                       "source": "rocprofv3 --kernel-trace --stats of `python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras` "
                                 "(tools/gpu_round_artefacts.sh), split by tools/rocprof_summary.py::solo_split",
                       "solo_launches": sp[0], "solo_avg_us": sp[1], "in_step_launches": sp[2], "in_step_avg_us": sp[3]},
                      open(sys.argv[i + 1], "w"), indent=1)
        if sp and sp[0] and sp[2]:
            print(f"\n# bev_pool_fwd_cells_vec_kernel by setting: SOLO launches (nothing else on the GPU; what roofline.kernel_ms reports) "
                  f"n={sp[0]} avg {sp[1]:.2f} us; launches inside the step (LiDAR branch beside them; roofline.kernel_ms_in_step) "
                  f"n={sp[2]} avg {sp[3]:.2f} us — the table's average above mixes the two")
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        print(f"\n# PMC ({os.path.relpath(f, d)}): mean counter value per dispatch")
        for k, cs in pmc_from_csv(f).items():
            for c, v in cs.items():
                print(f"{k[:78]:78s} {c:16s} n={len(v):4d} mean={sum(v) / len(v):14.1f}")


if __name__ == "__main__":
    main()
