"""Is the bev_pool kernel's time sensitive to WHERE its buffers sit?  Same launch, feature / output buffers carved out of one
big allocation at different byte skews; and repeated launches over a minute (thermal drift)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from bevfusion_amd import synth
from bevfusion_amd.bev_pool import BevPoolPlan
B = 8
dev = torch.device("cuda:0")
cfg = synth.CL_CONFIG
inp = synth.bev_pool_inputs(cfg, batch=B, seed=0, with_feats=False)
H, W, D = (int(v) for v in inp["nx"]); C = inp["channels"]
plan = BevPoolPlan.from_geometry(torch.from_numpy(inp["geom"]).to(dev), B, inp["origin"], inp["dx"], inp["nx"])
n = plan.n
fbytes, obytes = n * C * 4, B * D * H * W * C * 4
pool = torch.empty(fbytes + obytes + (64 << 20), dtype=torch.uint8, device=dev)
pool[: fbytes + (16 << 20)].view(torch.float32).normal_()
def timeit(feats, out, reps=10):
    for _ in range(2): plan.launch_forward(feats, out)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): plan.launch_forward(feats, out)
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("pool base % 2MiB =", pool.data_ptr() % (2 << 20))
for fs in (0, 256, 4096, 65536, 1 << 20):
    for os_ in (0, 256, 4096, 65536, 1 << 20, 3 << 20):
        feats = pool[fs: fs + fbytes].view(torch.float32).view(n, C)
        o0 = fbytes + (8 << 20) + os_
        out = pool[o0: o0 + obytes].view(torch.float32).view(B, D, H, W, C)
        print(f"feats skew {fs:8d} out skew {os_:8d}: {timeit(feats, out):7.1f} us", flush=True)
feats = pool[:fbytes].view(torch.float32).view(n, C)
out = pool[fbytes + (8 << 20): fbytes + (8 << 20) + obytes].view(torch.float32).view(B, D, H, W, C)
t0 = time.time()
while time.time() - t0 < 40:
    print(f"t={time.time() - t0:5.1f}s: {timeit(feats, out, 2000):7.1f} us", flush=True)
