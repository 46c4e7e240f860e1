"""Does the relative placement of the three row streams of a 32-channel SubM layer (features, residual, output) change its time?
(round 6: the second layer of the HIP-graph replay took 255 us instead of 143 on the wave-pair kernel at one placement.)
    python tools/slab_alias_probe.py <variant> [<variant> ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevfusion_amd import synth  # noqa: E402
from bevfusion_amd.spconv import ops as sops  # noqa: E402
from bevfusion_amd.voxel import voxelize_batch  # noqa: E402
from tools.sweep_spconv import timeit  # noqa: E402

MiB = 1 << 20


def main():
    variants = [int(a) for a in sys.argv[1:]] or [4000112, 4100128]
    dev = torch.device("cuda", 0)
    cfg = synth.CL_CONFIG
    frames = 8
    pts = [torch.from_numpy(synth.lidar_points(seed=b)).to(dev) for b in range(frames)]
    vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
    shape = list(cfg["sparse_shape"])
    ind = vc.int().contiguous()
    rbs = sops.build_rulebook(ind, frames, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], 1, False)
    ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
    rb = sops.build_rulebook(ind, frames, shape, 3, 1, 1, 1, True)
    n, c = ind.shape[0], 32
    arena = torch.empty(36 * 1024 * MiB, dtype=torch.uint8, device=dev)
    base = 17 * 1024 * MiB

    def view(off_mib):
        o = base + off_mib * MiB
        return arena[o:o + n * c * 2].view(torch.float16).view(n, c)

    w = (torch.randn(27, c, c, device=dev) / (27 * c) ** 0.5).half()
    img = sops.make_filter_image(w.view(27, 1, 1, c, c))
    kw = dict(bn_scale=torch.rand(c, device=dev) + 0.5, bn_shift=torch.randn(c, device=dev), relu=True,
              num_out_dev=torch.tensor([rb.num_out], dtype=torch.int32, device=dev))
    src = torch.randn(n, c, device=dev).half()
    placements = [(628, -16714), (-16714, -17342), (2284, -1466), (628, -16384), (628, -16712), (628, -16716), (628, -8192),
                  (628, -330), (628, 1256), (630, -16714), (1256, -16714), (None, -16714), (None, 628), (628, -16714 + 4096),
                  (628, -4096 - 330), (628, -2048 - 330), (628, -1024 - 330), (628, -512 - 330), (628, -256 - 330)]
    for v in variants:
        meta = sops.slab_build(rb.nbr, rb.num_out, None, sops.slab_block_rows(c, v))
        for res_off, out_off in placements:
            f = view(0)
            f.copy_(src)
            r = None
            if res_off is not None:
                r = view(res_off)
                r.copy_(src)
            o = view(out_off)
            t = min(timeit(lambda: sops.sparse_conv_slab(f, img, meta, rb.num_out, c, c, variant=v, out=o, residual=r, **kw))[0]
                    for _ in range(3))
            print(f"variant {v}  residual at {res_off} MiB, output at {out_off} MiB: {t:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
