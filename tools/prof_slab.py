"""Run ONE SubM layer shape of the flagship encoder a few times through a slab variant (or, variant < 0, the shipped gather
kernel) — the workload of the rocprofv3 --pmc passes in tools/pmc_slab.sh.
    python tools/prof_slab.py <cin 32|64|128> <variant> [frames=8] [reps=3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevfusion_amd import synth  # noqa: E402
from bevfusion_amd.spconv import ops as sops  # noqa: E402
from bevfusion_amd.spconv.fused import _variant_for  # noqa: E402
from bevfusion_amd.voxel import voxelize_batch  # noqa: E402

cin, variant = int(sys.argv[1]), int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda", 0)
cfg = synth.CL_CONFIG
pts = [torch.from_numpy(synth.lidar_points(seed=b)).to(dev) for b in range(frames)]
vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
shape = list(cfg["sparse_shape"])
ind = vc.int().contiguous()
stages = [(32, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (64, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (128, (3, 3, 3), (2, 2, 2), (1, 1, 0))]
for cout, ks, st, pd in stages:
    rbs = sops.build_rulebook(ind, frames, shape, list(ks), list(st), list(pd), 1, False)
    ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
    if cout == cin:
        break
rb = sops.build_rulebook(ind, frames, shape, 3, 1, 1, 1, True)
f = torch.randn(ind.shape[0], cin, device=dev).half()
w = (torch.randn(27, cin, cin, device=dev) / (27 * cin) ** 0.5).half()
img = sops.make_filter_image(w.view(27, 1, 1, cin, cin))
if variant >= 0:
    meta = sops.slab_build(rb.nbr, rb.num_out, None, sops.slab_block_rows(cin, variant))
for _ in range(reps):
    if variant >= 0:
        sops.sparse_conv_slab(f, img, meta, rb.num_out, cin, cin, variant=variant)
    else:
        sops.sparse_conv_tiled(f, img, rb.nbr, rb.num_out, 27, cin, cin, variant=_variant_for(frames, 27, cin, cin))
torch.cuda.synchronize()
print("done", rb.num_out)
