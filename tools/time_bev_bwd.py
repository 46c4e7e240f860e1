"""Time the bev_pool backward kernels alone (HIP events, 30 launches each) at B flagship frames: tools/time_bev_bwd.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bevfusion_amd import synth, _capi
from bevfusion_amd.bev_pool import BevPoolPlan
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
cfg = synth.CL_CONFIG
inp = synth.bev_pool_inputs(cfg, batch=B, seed=0, with_feats=False)
H, W, D = (int(v) for v in inp["nx"]); C = inp["channels"]
plan = BevPoolPlan.from_geometry(torch.from_numpy(inp["geom"]).to(dev), B, inp["origin"], inp["dx"], inp["nx"])
g = torch.randn((B, D, H, W, C), device=dev)
lib = _capi.load()
x = torch.empty((plan.n, C), device=dev)
cop = plan.cell_of_point()
def rows():
    lib.bevamd_bev_pool_backward_rows(_capi.ptr(g), _capi.ptr(plan.order), _capi.ptr(plan.ranks_sorted), _capi.ptr(x), plan.n, C, B, D, H, W, _capi.stream_ptr(dev))
def points():
    lib.bevamd_bev_pool_backward_points(_capi.ptr(g), _capi.ptr(cop), _capi.ptr(x), plan.n, C, B, D, H, W, _capi.stream_ptr(dev))
bytes_ = B * D * H * W * C * 4 + plan.n_kept() * C * 4
for name, fn in (("rows", rows), ("points", points)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30): fn()
    b.record(); b.synchronize()
    ms = a.elapsed_time(b) / 30
    print(f"{name}: {ms*1e3:.1f} us  algorithmic {bytes_/1e6:.0f} MB -> {bytes_/ms/1e6:.0f} GB/s = {bytes_/ms/1e6/8000:.3f} of 8 TB/s; written {plan.n*C*4/1e6:.0f} MB -> {plan.n*C*4/ms/1e6:.0f} GB/s")
