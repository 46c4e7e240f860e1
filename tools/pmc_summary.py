"""Per-kernel summary of the PMC passes of tools/pmc_bench.sh:  python tools/pmc_summary.py <dir> <out.json>  -> text on stdout.

Per kernel (mean per dispatch over all dispatches of the pass): HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE (KB units of rocprofv3;
the x2 is the gfx950 correction of MI355X_MICROARCH.md §HBM for wide coalesced reads: FETCH_SIZE tallies 128-byte requests at
64 B), L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS), MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x cycles of the
launch), LDS bank-conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, waiting share = SQ_WAIT_ANY / SQ_WAVE_CYCLES,
launch cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs), microseconds from the kernel trace of the same passes."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "bevamd::"):
        n = n.replace(p, "")
    return n.strip()


def main():
    d, out_json = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = {}
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        rec = {"dispatches": max(len(v) for v in cs.values()), "us": sum(dur[k]) / max(len(dur[k]), 1)}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            rec["fetch_kb_raw"], rec["write_kb"] = m["FETCH_SIZE"], m["WRITE_SIZE"]
            rec["hbm_mb"] = (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 / 1e6
            rec["hbm_gbs"] = rec["hbm_mb"] / max(rec["us"], 1e-9) * 1e3 if rec["us"] else None
        if "TCC_HIT_sum" in m:
            rec["l2_hit"] = m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0.0), 1.0)
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            rec["mfma_busy"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * cyc)
        if m.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_conflict_of_active"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
            if cyc:
                rec["lds_active_of_cu_cycles"] = m["SQ_LDS_IDX_ACTIVE"] / (256 * cyc)
        if m.get("SQ_WAVE_CYCLES"):
            rec["waiting"] = m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
        rows[k] = rec
    json.dump({"note": __doc__.split("\n\n")[1], "kernels": rows}, open(out_json, "w"), indent=1)
    order = sorted(rows, key=lambda k: -rows[k]["us"] * rows[k]["dispatches"])
    print(f"{'kernel':70s} {'n':>5s} {'us':>8s} {'HBM MB':>9s} {'GB/s':>7s} {'L2hit':>6s} {'MFMA%':>6s} {'LDScf%':>7s} {'wait%':>6s}")
    f = lambda v, s=1.0, w=7, p=1: (f"{v * s:{w}.{p}f}" if v is not None else " " * (w - 1) + "-")
    for k in order:
        r = rows[k]
        if "bevamd" not in k and not any(t in k for t in ("spconv", "bev_", "sp_", "vox_", "radix", "scan_", "depth_raster", "slab")):
            continue
        print(f"{k[:70]:70s} {r['dispatches']:5d} {r['us']:8.1f} {f(r.get('hbm_mb'), 1, 9)} {f(r.get('hbm_gbs'), 1, 7, 0)} "
              f"{f(r.get('l2_hit'), 100, 6)} {f(r.get('mfma_busy'), 100, 6)} {f(r.get('lds_conflict_of_active'), 100, 7)} {f(r.get('waiting'), 100, 6)}")


if __name__ == "__main__":
    main()
