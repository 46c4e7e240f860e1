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
f1, c1, _, t1 = voxelize_batch_device(pts, vs, pr, mp, mv, order="key")
rec = []
orig = fused._conv
def spy(conv, x, bn=None, relu=False, residual=None):
    y = orig(conv, x, bn, relu, residual)
    rec[-1].append((conv.in_channels, conv.out_channels, bool(conv.subm), y))
    return y
fused._conv = spy
outs = []
for prof in (False, True):
    rec.append([])
    fused.LAYER_PROFILE = [] if prof else None
    with torch.no_grad():
        outs.append(enc(f1, c1, B, num_voxels=t1, coors_order="linear"))
    fused.LAYER_PROFILE = None
    torch.cuda.synchronize()
print("dense equal", torch.equal(outs[0], outs[1]))
for i, (a, b) in enumerate(zip(rec[0], rec[1])):
    la = a[3].level
    na = int(la.n_dev.item()) if la.n_dev is not None else la.n_cap
    fa = a[3].features[:na]; fb = b[3].features[:na]
    nd = int((fa != fb).sum())
    print(i, a[:3], "rows", na, "mismatch", nd, "nan", int(torch.isnan(fb).sum()))
    if nd:
        r = (fa != fb).any(1).nonzero().flatten()
        print("  rows", r[:10].tolist(), "...", r[-3:].tolist(), "count", r.numel(), "blocks", sorted(set((r // 256).tolist()))[:20])
        break
