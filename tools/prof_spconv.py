"""Run each SparseEncoder conv layer shape a few times with the auto variant (for rocprofv3 --pmc passes)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevfusion_amd import synth  # noqa: E402
from bevfusion_amd.spconv import ops as sops  # noqa: E402
from bevfusion_amd.voxel import voxelize_batch  # noqa: E402

dev = torch.device("cuda", 0)
cfg = synth.CL_CONFIG
pts = torch.from_numpy(synth.lidar_points(seed=0)).to(dev)
vf, vc, _ = voxelize_batch([pts], cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
shape = list(cfg["sparse_shape"])
ind = vc.int().contiguous()
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
stages = [(16, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (32, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
          (64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 0))]
rb = sops.build_rulebook(ind, 1, shape, 3, 1, 1, 1, True)
todo = [(rb, ind.shape[0], 16, 16)]
for cin, cout, ks, st, pd in stages:
    rbs = sops.build_rulebook(ind, 1, shape, list(ks), list(st), list(pd), 1, False)
    ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
    rb = sops.build_rulebook(ind, 1, shape, 3, 1, 1, 1, True)
    todo.append((rb, ind.shape[0], cout, cout))
for rb, n_in, cin, cout in todo:
    f = torch.randn(n_in, cin, device=dev).half()
    w = (torch.randn(27, cin, cout, device=dev) / (27 * cin) ** 0.5).half()
    img = sops.make_filter_image(w.view(27, 1, 1, cin, cout))
    for _ in range(reps):
        sops.sparse_conv_tiled(f, img, rb.nbr, rb.num_out, 27, cin, cout, variant=variant)
torch.cuda.synchronize()
print("done")
