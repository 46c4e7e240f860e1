import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from test_gpu_keyorder import flagship_encoder, linear_key, CFG
from bevfusion_amd import synth
from bevfusion_amd.spconv import fused
from bevfusion_amd.voxel import voxelize_batch_device
dev = torch.device("cuda:0")
B = int(os.environ.get("B", 8))
pts = [torch.from_numpy(synth.lidar_points(seed=70 + b, sweeps=10 if b < 2 else 3)).to(dev) for b in range(B)]
vs, pr, mp, mv = CFG["voxel_size"], CFG["point_cloud_range"], CFG["max_num_points"], CFG["max_voxels"][1]
enc = flagship_encoder(dev, torch.float16)
rec = []
orig = fused._conv
def spy(conv, x, bn=None, relu=False, residual=None):
    y = orig(conv, x, bn, relu, residual)
    rec[-1].append((conv.in_channels, conv.out_channels, bool(conv.subm), y))
    return y
fused._conv = spy
outs = []
for order in ("first", "key"):
    f, c, _, t = voxelize_batch_device(pts, vs, pr, mp, mv, order=order)
    rec.append([])
    with torch.no_grad():
        outs.append(enc(f, c, B, num_voxels=t, coors_order="linear" if order == "key" else None))
    torch.cuda.synchronize()
print("dense equal:", torch.equal(outs[0], outs[1]), "max diff", float((outs[0].float() - outs[1].float()).abs().max()),
      "n diff", int((outs[0] != outs[1]).sum()))
for i, (a, b) in enumerate(zip(rec[0], rec[1])):
    la, lb = a[3].level, b[3].level
    na = int(la.n_dev.item()) if la.n_dev is not None else la.n_cap
    nb = int(lb.n_dev.item()) if lb.n_dev is not None else lb.n_cap
    ka = linear_key(la.indices[:na].cpu().numpy(), la.shape); kb = linear_key(lb.indices[:nb].cpu().numpy(), lb.shape)
    pa, pb = np.argsort(ka, kind="stable"), np.argsort(kb, kind="stable")
    fa = a[3].features[:na].cpu().numpy()[pa]; fb = b[3].features[:nb].cpu().numpy()[pb]
    same_keys = na == nb and np.array_equal(ka[pa], kb[pb])
    nd = int((fa != fb).sum()) if same_keys else -1
    print(i, a[:3], "rows", na, nb, "keys equal", same_keys, "feature mismatches", nd,
          "maxdiff", float(np.abs(fa.astype(np.float32) - fb.astype(np.float32)).max()) if same_keys else None)
    if nd > 0:
        r, cc = np.argwhere(fa != fb)[0]
        print("   first at sorted row", r, "col", cc, fa[r, cc], fb[r, cc], "row in key-order set:", pb[r], "block", pb[r] // 256)
        break
