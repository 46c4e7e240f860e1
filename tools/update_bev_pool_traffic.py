"""Refresh profiles/bev_pool_traffic.json (what bench.py's roofline.traffic reads) from a tools/pmc_bench.sh summary:
    python tools/update_bev_pool_traffic.py <pmcb_infer.json> <round> [frames per launch = 8]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, rnd = sys.argv[1], int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 8
k = [v for n, v in json.load(open(src))["kernels"].items() if n.startswith("bev_pool_fwd_cells_vec_kernel")][0]
hbm = (2.0 * k["fetch_kb_raw"] + k["write_kb"]) * 1024
alg = 4988319168 if frames == 8 else None
out = {
    "kernel": "bev_pool_fwd_cells_vec_kernel", "round": rnd,
    "source": [f"profiles/r{rnd:02d}_pmc_infer_per_kernel.txt / .json (tools/pmc_bench.sh infer --no-graph: FETCH_SIZE and WRITE_SIZE in separate "
               f"rocprofv3 --pmc passes over the default bench command, {frames} frames per launch, {k['dispatches']} dispatches)"],
    "frames_per_launch_when_measured": frames,
    "FETCH_SIZE_KB_per_launch_raw": k["fetch_kb_raw"], "WRITE_SIZE_KB_per_launch_raw": k["write_kb"],
    "correction": "FETCH_SIZE x2 on gfx950 for 16 B/lane coalesced reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
    "hbm_bytes_per_launch": hbm, "hbm_bytes_per_frame": hbm / frames,
    "kernel_us_under_counters": k["us"], "l2_hit": k.get("l2_hit"),
}
if alg:
    out["algorithmic_bytes_per_launch"] = alg
    out["traffic_over_algorithmic"] = hbm / alg
json.dump(out, open(os.path.join(ROOT, "profiles", "bev_pool_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
