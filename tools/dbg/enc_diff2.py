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
def cmp(name, a, b):
    print(name, "equal", torch.equal(a, b), "ndiff", int((a != b).sum()), "max", float((a.float() - b.float()).abs().max()), flush=True)
with torch.no_grad():
    ref = enc(f0, c0, B, num_voxels=t0)
    ref2 = enc(f0, c0, B, num_voxels=t0)
    cmp("first vs first again", ref, ref2)
    a1 = enc(f1, c1, B, num_voxels=t1, coors_order="linear")
    cmp("key plain", a1, ref)
    fused.LAYER_PROFILE = []
    got = enc(f1, c1, B, num_voxels=t1, coors_order="linear")
    fused.LAYER_PROFILE = None
    cmp("key profiled", got, ref)
    again = enc(f1, c1, B, num_voxels=t1, coors_order="linear")
    cmp("key again", again, ref)
    lvl = enc.prepare_geometry(c1, B, num_voxels=t1, coors_order="linear")
    prepared = enc(f1, c1, B, num_voxels=t1, geometry=lvl)
    cmp("key prepared", prepared, ref)
    print("status", fused.geometry_status(lvl))
    fused.LAYER_PROFILE = []
    gotf = enc(f0, c0, B, num_voxels=t0)
    fused.LAYER_PROFILE = None
    cmp("first profiled", gotf, ref)
