import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from bevfusion_amd import synth
from bevfusion_amd.spconv import ops as sops
from bevfusion_amd.voxel import voxelize_batch
sys.path.insert(0, "/root/repo/tools")
from sweep_spconv import timeit
dev = torch.device("cuda", 0)
cfg = synth.CL_CONFIG
pts = torch.from_numpy(synth.lidar_points(seed=0)).to(dev)
vf, vc, _ = voxelize_batch([pts], cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
shape = list(cfg["sparse_shape"]); ind = vc.int().contiguous()
for i, (ks, st, pd) in enumerate([((3,3,3),(2,2,2),(1,1,1)), ((3,3,3),(2,2,2),(1,1,1)), ((3,3,3),(2,2,2),(1,1,0))]):
    rbs = sops.build_rulebook(ind, 1, shape, list(ks), list(st), list(pd), 1, False)
    ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
    cin = [32, 64, 128][i]
    n = ind.shape[0]
    def morton_key(c):
        x, y, z = c[:, 1].long(), c[:, 2].long(), c[:, 3].long()
        return (((x >> 2) * 4096 + (y >> 2)) * 16 + (x & 3) * 4 + (y & 3)) * 64 + z
    orders = {"sorted (x,y,z)": torch.arange(n, device=dev), "random": torch.randperm(n, device=dev),
              "4x4 bricks": torch.argsort(morton_key(ind))}
    for name, perm in orders.items():
        ii = ind[perm].contiguous()
        rb = sops.build_rulebook(ii, 1, shape, 3, 1, 1, 1, True)
        f = torch.randn(n, cin, device=dev).half()
        w = (torch.randn(27, cin, cin, device=dev) / (27 * cin) ** 0.5).half()
        img = sops.make_filter_image(w.view(27, 1, 1, cin, cin))
        t, _ = timeit(lambda: sops.sparse_conv_tiled(f, img, rb.nbr, rb.num_out, 27, cin, cin))
        print(f"level {i+2} {cin}->{cin} rows {n:7d} order {name:16s}: {t:6.1f} us")
