"""Time the batched voxelizer (bevamd_voxelize_mean_batch) against one call per sample on 8 flagship sweeps:
    python tools/vox_time.py"""
import torch, time, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevfusion_amd import synth
from bevfusion_amd import voxel as V
dev = torch.device("cuda:0")
cfg = synth.CL_CONFIG
pts = [torch.from_numpy(synth.lidar_points(seed=s)).to(dev) for s in range(8)]
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
args = (cfg["voxel_size"], cfg["point_cloud_range"], 10, 120000)
print("batched packed   ms", t(lambda: V._voxelize_mean_batch(pts, *args, packed=True)))
print("lanes + compact  ms", t(lambda: (V._voxelize_mean_lanes(pts, *args))))
V._VOXEL_BATCHED = False
print("old device path  ms", t(lambda: V.voxelize_batch_device(pts, *args)))
