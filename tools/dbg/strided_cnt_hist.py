"""Staged-range lengths of the STRIDED layers' slab metadata (sorted-key route) per block size, flagship encoder levels."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevfusion_amd import synth
from bevfusion_amd.spconv import fused
from bevfusion_amd.voxel import voxelize_batch_device
frames = int(os.environ.get("FRAMES", "8")); dev = torch.device("cuda", 0); cfg = synth.CL_CONFIG
pts = [torch.from_numpy(synth.lidar_points(seed=b)).to(dev) for b in range(frames)]
vf, vc, _, cnt = voxelize_batch_device(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1], order="key")
lvl = fused.Level(vc.int().contiguous(), vc.shape[0], cnt.reshape(-1)[:1].int().contiguous(), frames, list(cfg["sparse_shape"]), linear_order=True)
stages = [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 0))]
for i, (ks, st, pd) in enumerate(stages):
    out, _ = lvl.downsample(ks, st, pd, want_nbr=False)
    torch.cuda.synchronize()
    m = int(out.n_dev.item())
    for bm in (64, 128, 256):
        import time
        meta = lvl.down_slab(ks, st, pd, bm); torch.cuda.synchronize()
        t0 = time.perf_counter()
        nblk = (m + bm - 1) // bm
        c = meta.hdr.cpu().numpy().view(np.int32).reshape(-1)[: nblk * 6].reshape(nblk, 3, 2)[:, :, 1] & 0x3FFFFFFF
        c = c[c > 0]
        print(f"strided stage {i} in_rows~{int(lvl.n_dev.item()) if lvl.n_dev is not None else lvl.n_cap} out_rows={m} bm={bm}: mean={c.mean():.0f} p50/90/99/99.9={np.percentile(c,[50,90,99,99.9]).round(0).tolist()} max={c.max()} "
              + " ".join(f">{t}:{(c > t).mean()*100:.1f}%" for t in (192, 256, 320, 384, 512, 640)), flush=True)
    lvl = out
