"""Distribution of the per-(block, plane) staged-range lengths of the slab metadata on the flagship encoder levels."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bevfusion_amd import synth
from bevfusion_amd.spconv import ops as sops
from bevfusion_amd.voxel import voxelize_batch
frames = int(os.environ.get("FRAMES", "8"))
dev = torch.device("cuda", 0)
cfg = synth.CL_CONFIG
pts = [torch.from_numpy(synth.lidar_points(seed=b)).to(dev) for b in range(frames)]
vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
shape = list(cfg["sparse_shape"]); ind = vc.int().contiguous()
stages = [(16, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (32, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 0))]
for cin, cout, ks, st, pd in stages:
    rbs = sops.build_rulebook(ind, frames, shape, list(ks), list(st), list(pd), 1, False)
    ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
    rb = sops.build_rulebook(ind, frames, shape, 3, 1, 1, 1, True)
    for bm in (64, 128, 256):
        meta = sops.slab_build(rb.nbr, rb.num_out, None, bm)
        nblk = (rb.num_out + bm - 1) // bm
        cnt = meta.hdr.cpu().numpy().view(np.int32).reshape(-1)[: nblk * 6].reshape(nblk, 3, 2)[:, :, 1] & 0x3FFFFFFF
        c = cnt[cnt > 0]
        qs = np.percentile(c, [50, 75, 90, 95, 99, 99.9])
        print(f"cout={cout} shape={shape} rows={rb.num_out} bm={bm}: planes={c.size} mean={c.mean():.1f} pct50/75/90/95/99/99.9={qs.round(0).tolist()} max={c.max()} "
              + " ".join(f">{t}:{(c > t).mean() * 100:.1f}%" for t in (96, 112, 128, 160, 192, 256, 384)))
