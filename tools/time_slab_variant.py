"""Time chosen slab variants on the SubM layers of the flagship encoder (8 frames by default): A/B harness for kernel experiments.
    [BEVAMD_LIB=bevfusion_amd/lib/exp/<name>.so] python tools/time_slab_variant.py 32:2324410 64:1644222 128:1644220"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevfusion_amd import synth  # noqa: E402
from bevfusion_amd.spconv import ops as sops  # noqa: E402
from bevfusion_amd.voxel import voxelize_batch  # noqa: E402
from tools.sweep_spconv import timeit  # noqa: E402


def main():
    want = {}
    for a in sys.argv[1:]:
        c, v = a.split(":")
        want.setdefault(int(c), []).append(int(v))
    frames = int(os.environ.get("FRAMES", "8"))
    dev = torch.device("cuda", 0)
    cfg = synth.CL_CONFIG
    pts = [torch.from_numpy(synth.lidar_points(seed=b)).to(dev) for b in range(frames)]
    vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
    shape = list(cfg["sparse_shape"])
    ind = vc.int().contiguous()
    stages = [(16, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (32, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 0))]
    tag = os.path.basename(os.environ.get("BEVAMD_LIB", "shipped"))
    if 16 in want:   # level 1: the narrow-row kernels (variant 0 | 3000256 | 3000128)
        rb = sops.build_rulebook(ind, frames, shape, 3, 1, 1, 1, True)
        f = torch.randn(ind.shape[0], 16, device=dev).half()
        w = (torch.randn(27, 16, 16, device=dev) / (27 * 16) ** 0.5).half()
        img = sops.make_filter_image(w.view(27, 1, 1, 16, 16))
        for v in want[16]:
            meta = sops.slab_build(rb.nbr, rb.num_out, None, sops.slab_block_rows(16, v))
            t = min(timeit(lambda: sops.sparse_conv_slab(f, img, meta, rb.num_out, 16, 16, variant=v))[0] for _ in range(3))
            print(f"{tag:24s}  16->16  rows={rb.num_out:8d} variant {v}: {t:7.1f} us", flush=True)
    for cin, cout, ks, st, pd in stages:
        rbs = sops.build_rulebook(ind, frames, shape, list(ks), list(st), list(pd), 1, False)
        ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
        if cout not in want:
            continue
        rb = sops.build_rulebook(ind, frames, shape, 3, 1, 1, 1, True)
        c = cout
        f = torch.randn(ind.shape[0], c, device=dev).half()
        w = (torch.randn(27, c, c, device=dev) / (27 * c) ** 0.5).half()
        img = sops.make_filter_image(w.view(27, 1, 1, c, c))
        # EPI=0 bare product, 1 folded BatchNorm + ReLU (first convolution of a block), 2 ... + residual (the second), 3 = also through
        # the device-side row count, as the encoder's sync-free route runs it
        epi = int(os.environ.get("EPI", "0"))
        kw = {}
        if epi >= 1:
            kw = dict(bn_scale=torch.rand(c, device=dev) + 0.5, bn_shift=torch.randn(c, device=dev), relu=True)
        if epi >= 2:
            kw["residual"] = torch.randn(ind.shape[0], c, device=dev).half()
        if epi >= 3:
            kw["num_out_dev"] = torch.tensor([rb.num_out], dtype=torch.int32, device=dev)
        out = torch.empty(ind.shape[0], c, device=dev, dtype=torch.float16)
        for v in want[c]:
            meta = sops.slab_build(rb.nbr, rb.num_out, None, sops.slab_block_rows(c, v))
            t = min(timeit(lambda: sops.sparse_conv_slab(f, img, meta, rb.num_out, c, c, variant=v, out=out, **kw))[0] for _ in range(3))
            print(f"{tag:24s} {c:3d}->{c:<3d} rows={rb.num_out:8d} epi {epi} variant {v}: {t:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
