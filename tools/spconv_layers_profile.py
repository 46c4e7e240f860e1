"""Per-LAYER figures of the 21 SparseEncoder convolutions from stored rocprofv3 runs (counter bytes next to
the ideal bytes, in-graph microseconds from a kernel trace instead of eager events):

    python tools/spconv_layers_profile.py <graph kernel-trace dir (.db)> <tools/pmc_bench.sh dir (--no-graph passes)> <bench line .json> <out.json>

The convolution kernels of one encoder pass always run in the same order, so the k-th convolution kernel of a pass IS layer k:
the dispatches are cut into groups of 21 in start order (graph trace: the LAST 20 groups = replays of the timed steps) and
averaged per position.  HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md §HBM); metadata bytes =
what the kernel reads besides features and filters: 54 B per output row of 16-bit slots (slab kernels) or K x 4 B per output row
of int32 neighbour table (gather kernels)."""
import csv
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

NL = 21


def is_conv(name):
    return ("spconv_sla" in name or "spconv_stream" in name or "spconv_resident" in name) and "filter_image" not in name


def short(name):
    n = name.split("(")[0]
    for p in ("void ", "bevamd::", "slab::", "tile::"):
        n = n.replace(p, "")
    return n.strip()


def groups(seq):
    n = len(seq) // NL
    return [seq[i * NL:(i + 1) * NL] for i in range(n)]


def main():
    trace_dir, pmc_dir, bench_json, out = sys.argv[1:5]
    text = open(bench_json).read().strip()
    try:
        line = json.loads(text)                       # the full result of a run (profiles/bench_last_full.json, round 5)
    except ValueError:
        line = json.loads(text.split("\n")[-1])       # a bench line of rounds 1-4
    layers = line["roofline_spconv"]["layers"]
    assert len(layers) == NL
    db = sqlite3.connect(glob.glob(os.path.join(trace_dir, "**", "*.db"), recursive=True)[0])
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    conv = [(short(n), (e - s) / 1e3) for n, s, e in rows if is_conv(n)]
    gs = groups(conv)[-20:]
    in_graph = [sum(g[i][1] for g in gs) / len(gs) for i in range(NL)]
    names = [gs[-1][i][0] for i in range(NL)]
    per = [defaultdict(list) for _ in range(NL)]
    eager = [[] for _ in range(NL)]
    for f in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
        by_disp = {}
        for r in csv.DictReader(open(f)):
            if is_conv(r["Kernel_Name"]):
                by_disp.setdefault(int(r["Dispatch_Id"]), []).append(r)
        seq = [by_disp[k] for k in sorted(by_disp)]
        for g in groups(seq):
            for i, recs in enumerate(g):
                for r in recs:
                    per[i][r["Counter_Name"]].append(float(r["Counter_Value"]))
                eager[i].append((int(recs[0]["End_Timestamp"]) - int(recs[0]["Start_Timestamp"])) / 1e3)
    res = []
    for i, l in enumerate(layers):
        m = {c: sum(v) / len(v) for c, v in per[i].items()}
        slab = l["kernel"] == "slab"
        meta = l["rows_out"] * (54 if slab else int(l["layer"].split("K=")[1]) * 4)
        rec = dict(layer=l["layer"], kernel=names[i], variant=l["variant"], rows_in=l["rows_in"], rows_out=l["rows_out"], pairs=l["pairs"],
                   in_graph_us=round(in_graph[i], 1), eager_us_under_counters=round(sum(eager[i]) / max(len(eager[i]), 1), 1),
                   ideal_mb=l["ideal_mb"], metadata_mb=round(meta / 1e6, 1))
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            rec["hbm_mb_counters"] = round((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 / 1e6, 1)
            rec["hbm_over_ideal"] = round(rec["hbm_mb_counters"] / max(l["ideal_mb"], 1e-9), 2)
        if "TCC_HIT_sum" in m:
            rec["l2_hit"] = round(m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0), 1), 3)
        cyc = m.get("GRBM_GUI_ACTIVE", 0) / 8
        if cyc and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            rec["mfma_busy"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc), 3)
        rec["tflops_in_graph"] = round(l["gflop"] / max(in_graph[i], 1e-9) * 1e-3 * 1e3, 1)
        res.append(rec)
    tot = sum(r["in_graph_us"] for r in res)
    json.dump({"note": __doc__.split("\n\n")[2], "frames_per_step": line["config"].get("frames_per_step_per_gpu"),
               "in_graph_total_us": round(tot, 1), "layers": res}, open(out, "w"), indent=1)
    print(f"{'layer':26s} {'kernel':52s} {'graph us':>8s} {'HBM MB':>8s} {'ideal':>7s} {'meta':>6s} {'L2hit':>6s} {'MFMA':>6s}")
    for r in res:
        print(f"{r['layer']:26s} {r['kernel'][:52]:52s} {r['in_graph_us']:8.1f} {r.get('hbm_mb_counters', 0):8.1f} {r['ideal_mb']:7.1f} "
              f"{r['metadata_mb']:6.1f} {r.get('l2_hit', 0):6.3f} {r.get('mfma_busy', 0):6.3f}")
    print("in-graph total of the 21 convolutions:", round(tot, 1), "us")


if __name__ == "__main__":
    main()
