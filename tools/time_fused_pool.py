"""Fused depth (x) context pooling, B flagship frames: column formulation (pass 1 + pass 2) vs the cell-centric kernels
(camera-sector schedule / frame-major walk).  tools/time_fused_pool.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bevfusion_amd import synth, bev_pool as bp
from bevfusion_amd.bev_pool import BevPoolPlan
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
cfg = synth.CL_CONFIG
inp = synth.bev_pool_inputs(cfg, batch=B, seed=0, with_feats=False)
H, W, D = (int(v) for v in inp["nx"]); C = inp["channels"]
plan = BevPoolPlan.from_geometry(torch.from_numpy(inp["geom"]).to(dev), B, inp["origin"], inp["dx"], inp["nx"])
fh, fw = cfg["feature_size"]; ncam = cfg["num_cameras"]
dbins = plan.n // (B * ncam * fh * fw)
g = torch.Generator(device=dev).manual_seed(1)
depth = torch.softmax(torch.randn((B * ncam, dbins, fh, fw), generator=g, device=dev), 1).reshape(-1)
ctx = torch.randn((B * ncam * fh * fw, C), generator=g, device=dev)
outs = {}
ONLY_COLUMNS = os.environ.get("BEVAMD_TIME_FUSED_ONLY_COLUMNS", "0") == "1"   # A/B runs: skip the cell-centric kernels
MODES = (("cells frame-major", "cells", False), ("cells scheduled", "cells", True), ("columns", "columns!", True))
for name, mode, flag in (MODES[2:] if ONLY_COLUMNS else MODES):
    bp._FUSED_SCHEDULE = flag
    for _ in range(3): o = plan.launch_fused(depth, ctx, dbins, fh, fw, mode=mode)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): o = plan.launch_fused(depth, ctx, dbins, fh, fw, mode=mode)
    b.record(); b.synchronize()
    outs[name] = o.clone()
    print(f"{name}: {a.elapsed_time(b)/30*1e3:.1f} us per {B} frames = {a.elapsed_time(b)/30*1e3/B:.1f} us/frame")
if ONLY_COLUMNS:
    print("checksum", float(outs["columns"].double().sum()), float(outs["columns"].abs().max()))
    sys.exit(0)
print("cells variants bit-identical:", torch.equal(outs["cells frame-major"], outs["cells scheduled"]))
print("columns vs cells max |diff|:", float((outs["columns"] - outs["cells scheduled"]).abs().max()), "max |out|:", float(outs["columns"].abs().max()))
cols = plan.fused_columns(dbins, fh, fw, C, force=True)
print(f"runs {cols.nruns} for {cols.n_kept} kept points ({cols.n_kept / max(cols.nruns, 1):.1f} points per run); partial rows {cols.nruns * C * 4 / 1e6:.1f} MB")
