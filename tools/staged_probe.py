"""Staged-rows sparse-conv variants (kind 3) vs the shipped ones on the flagship frame's sorted levels: bitwise equality and time.
    python tools/staged_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevfusion_amd import synth  # noqa: E402
from bevfusion_amd.spconv import ops as sops  # noqa: E402
from bevfusion_amd.voxel import voxelize_batch  # noqa: E402
from tools.sweep_spconv import timeit  # noqa: E402

STG = (3111, 3211, 3121)


def range_stats(rb, bm):
    """rows a staged tile of `bm` output rows would copy (sum over the 3 kernel x-planes of max-min+1 of the neighbour ids)"""
    K, m = rb.nbr.shape[0], rb.num_out
    span = K // 3 if K % 3 == 0 else K
    nt = (m + bm - 1) // bm
    nb = torch.full((K, nt * bm), -1, dtype=torch.int32, device=rb.nbr.device)
    nb[:, :m] = rb.nbr[:, :m]
    tot = torch.zeros(nt, dtype=torch.int64, device=nb.device)
    for g in range(K // span):
        v = nb[g * span:(g + 1) * span].reshape(span, nt, bm).permute(1, 0, 2).reshape(nt, -1).long()
        lo = torch.where(v >= 0, v, torch.full_like(v, 1 << 40)).min(1).values
        hi = v.max(1).values
        tot += torch.where(hi >= 0, hi - lo + 1, torch.zeros_like(hi))
    q = torch.quantile(tot.float(), torch.tensor([0.5, 0.9, 0.99, 1.0], device=tot.device))
    return f"bm={bm}: rows/tile median {q[0]:.0f} p90 {q[1]:.0f} p99 {q[2]:.0f} max {q[3]:.0f}; >5x: {float((tot > 5 * bm).float().mean()) * 100:.1f}% >3x: {float((tot > 3 * bm).float().mean()) * 100:.1f}%"


def main():
    dev = torch.device("cuda", 0)
    dt = torch.float16
    cfg = synth.CL_CONFIG
    pts = torch.from_numpy(synth.lidar_points(seed=0)).to(dev)
    vf, vc, _ = voxelize_batch([pts], cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
    shape = list(cfg["sparse_shape"])
    ind = vc.int().contiguous()
    stages = [(16, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (32, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
              (64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 0)), (128, 128, (1, 1, 3), (1, 1, 2), (0, 0, 0))]
    layers = []
    for i, (cin, cout, ks, st, pd) in enumerate(stages):
        rbs = sops.build_rulebook(ind, 1, shape, list(ks), list(st), list(pd), 1, False)
        if cin >= 32:
            layers.append((f"spconv{i + 1} {cin}->{cout} k{ks}", rbs, ind.shape[0], cin, cout))
        ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
        if i < 3:
            rb = sops.build_rulebook(ind, 1, shape, 3, 1, 1, 1, True)
            layers.append((f"subm{i + 2} {cout}->{cout}", rb, ind.shape[0], cout, cout))
    for name, rb, n_in, cin, cout in layers:
        K = rb.nbr.shape[0]
        torch.manual_seed(0)
        f = torch.randn(n_in, cin, device=dev).to(dt)
        w = (torch.randn(K, cin, cout, device=dev) / (K * cin) ** 0.5).to(dt)
        img = sops.make_filter_image(w.view(K, 1, 1, cin, cout))
        res = torch.randn(rb.num_out, cout, device=dev).to(dt)
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        run = lambda v: sops.sparse_conv_tiled(f, img, rb.nbr, rb.num_out, K, cin, cout, bn_scale=sc, bn_shift=sh, residual=res,
                                               relu=True, variant=v)
        ref = run(0)
        t0, _ = timeit(lambda: run(0))
        print(f"{name:34s} rows_in={n_in:7d} rows_out={rb.num_out:7d}  shipped {t0:7.1f} us")
        print("    " + range_stats(rb, 64) + "\n    " + range_stats(rb, 128))
        for v in STG:
            try:
                out = run(v)
            except RuntimeError as e:
                continue
            same = torch.equal(out, ref)
            t, _ = timeit(lambda: run(v))
            extra = ""
            for d in (1, 2, 3):      # 1: no main loop, 2: no staging copy, 3: neither
                os.environ["BEVAMD_STAGED_DBG"] = str(d)
                td, _ = timeit(lambda: run(v))
                extra += f"  dbg{d}={td:6.1f}"
            os.environ.pop("BEVAMD_STAGED_DBG")
            print(f"    variant {v}: {t:7.1f} us   bit-identical={same}  max|diff|={float((out.float() - ref.float()).abs().max()):.3g}{extra}")


if __name__ == "__main__":
    main()
