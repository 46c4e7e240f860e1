"""Time the kept slab variants on the flagship SubM layer shapes with and without the fused epilogue operands
(bias / folded BatchNorm / ReLU / residual):   python tools/slab_epilogue_cost.py [frames=8]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # sweep_spconv.timeit
from bevfusion_amd import synth  # noqa: E402
from bevfusion_amd.spconv import ops as sops  # noqa: E402
from bevfusion_amd.spconv.fused import _SLAB_DEFAULT  # noqa: E402
from bevfusion_amd.voxel import voxelize_batch  # noqa: E402
from sweep_spconv import timeit  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
cfg = synth.CL_CONFIG
pts = [torch.from_numpy(synth.lidar_points(seed=b)).to(dev) for b in range(frames)]
vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
shape, ind = list(cfg["sparse_shape"]), vc.int().contiguous()
for cin, ks, st, pd in [(32, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (64, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (128, (3, 3, 3), (2, 2, 2), (1, 1, 0))]:
    rbs = sops.build_rulebook(ind, frames, shape, list(ks), list(st), list(pd), 1, False)
    ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
    rb = sops.build_rulebook(ind, frames, shape, 3, 1, 1, 1, True)
    n = ind.shape[0]
    f = torch.randn(n, cin, device=dev).half()
    res = torch.randn(n, cin, device=dev).half()
    w = (torch.randn(27, cin, cin, device=dev) / (27 * cin) ** 0.5).half()
    img = sops.make_filter_image(w.view(27, 1, 1, cin, cin))
    sc, sh = torch.rand(cin, device=dev) + 0.5, torch.randn(cin, device=dev)
    v = _SLAB_DEFAULT[cin]
    meta = sops.slab_build(rb.nbr, rb.num_out, None, sops.slab_block_rows(cin, v))
    out = torch.empty(n, cin, device=dev, dtype=torch.half)
    for name, kw in [("plain", {}), ("bn+relu", dict(bn_scale=sc, bn_shift=sh, relu=True)),
                     ("bn+relu+residual", dict(bn_scale=sc, bn_shift=sh, relu=True, residual=res))]:
        t, _ = timeit(lambda: sops.sparse_conv_slab(f, img, meta, n, cin, cin, variant=v, out=out, **kw))
        print(f"{cin:4d}->{cin:<4d} rows={n:8d} variant {v}: {name:18s} {t:7.1f} us")
