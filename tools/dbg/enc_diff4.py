import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_keyorder import flagship_encoder, linear_key, CFG
from bevfusion_amd import synth
from bevfusion_amd.spconv import fused
from bevfusion_amd.voxel import voxelize_batch_device
dev = torch.device("cuda:0")
B = 8
pts = [torch.from_numpy(synth.lidar_points(seed=70 + b, sweeps=10 if b < 2 else 3)).to(dev) for b in range(B)]
vs, pr, mp, mv = CFG["voxel_size"], CFG["point_cloud_range"], CFG["max_num_points"], CFG["max_voxels"][1]
enc = flagship_encoder(dev, torch.float16)
f0, c0, _, t0 = voxelize_batch_device(pts, vs, pr, mp, mv)
f1, c1, _, t1 = voxelize_batch_device(pts, vs, pr, mp, mv, order="key")
rec = []
orig = fused._conv
def spy(conv, x, bn=None, relu=False, residual=None):
    y = orig(conv, x, bn, relu, residual)
    rec[-1].append((conv.in_channels, conv.out_channels, bool(conv.subm), y))
    return y
fused._conv = spy
def run(f, c, t, order, prof):
    rec.append([])
    fused.LAYER_PROFILE = [] if prof else None
    with torch.no_grad():
        out = enc(f, c, B, num_voxels=t, coors_order=order)
    fused.LAYER_PROFILE = None
    torch.cuda.synchronize()
    return out, rec[-1]
ref, _ = run(f0, c0, t0, None, False)
ref2, _ = run(f0, c0, t0, None, False)
a1, good = run(f1, c1, t1, "linear", False)
got, layers = run(f1, c1, t1, "linear", True)
print("plain equal", torch.equal(a1, ref), "profiled equal", torch.equal(got, ref))
n_live = int(t1.item())
for i, (a, b) in enumerate(zip(good, layers)):
    la = a[3].level
    na = int(la.n_dev.item()) if la.n_dev is not None else la.n_cap
    fa = a[3].features[:na]; fb = b[3].features[:na]
    bad = (fa != fb).any(1)
    print(i, a[:3], "rows", na, "bad rows", int(bad.sum()), "nan", int(torch.isnan(fb.float()).any(1).sum()))
    if bad.any():
        r = bad.nonzero().flatten()
        print("   rows", r[:12].tolist(), "last", int(r[-1]), "blocks", sorted(set((r // 256).tolist()))[:16], "nblocks", len(set((r // 256).tolist())))
        lb = b[3].level
        print("   index kinds", la.index_kind, lb.index_kind, "status", fused.geometry_status(lb))
        break
# dense tail only
print("dense diff cells", int((got != ref).any(1).sum()))
