"""Time every tiled sparse-conv variant on the SparseEncoder's layer shapes (synthetic flagship frame).
    python tools/sweep_spconv.py [--dtype fp16|bf16]   -> table on stdout (copy into profiles/)"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from bevfusion_amd import synth  # noqa: E402
from bevfusion_amd.spconv import ops as sops  # noqa: E402
from bevfusion_amd.voxel import voxelize_batch  # noqa: E402

RES = (1221, 1222, 1223, 1421, 1422, 1211)
STR = (2111, 2112, 2113, 2121, 2122, 2123, 2211, 2212, 2213, 2221, 2222)


_BUSY = None


def timeit(fn, iters=20):
    """us per launch with the host taken out of the picture: a ~5 ms matmul keeps the GPU busy while the host
    enqueues all `iters` launches, the events then bracket back-to-back kernel execution."""
    global _BUSY
    if _BUSY is None:
        _BUSY = torch.randn(8192, 8192, device="cuda")
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _BUSY @ _BUSY
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    t = s.elapsed_time(e) * 1e3 / iters
    return t, t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="fp16")
    ap.add_argument("--frames", type=int, default=1, help="frames in the batch (row counts scale with it)")
    ap.add_argument("--small", action="store_true", help="only the layers with < 32 input channels (and no round-0 kernel)")
    ap.add_argument("--quick", action="store_true", help="skip the round-0 kernel and the layers with < 32 input channels")
    ap.add_argument("--slab", action="store_true", help="SubM layers of the sorted levels: every slab variant next to the shipped gather variant")
    ap.add_argument("--cap-factor", type=float, default=0.0, help="with --slab: also time each variant launched for cap-factor x the live rows (device row count), as the sync-free encoder launches it")
    ap.add_argument("--ablate", action="store_true", help="time the 64->64 / 128->128 SubM layers with parts of the kernel compiled out (needs `python -m bevfusion_amd.build --profiling`)")
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    dev = torch.device("cuda", 0)
    cfg = synth.CL_CONFIG
    pts = [torch.from_numpy(synth.lidar_points(seed=b)).to(dev) for b in range(args.frames)]
    vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
    shape = list(cfg["sparse_shape"])
    ind = vc.int().contiguous()
    NB = args.frames
    layers = []   # (name, rulebook, n_in, cin, cout)
    stages = [(16, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (32, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
              (64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 0)), (128, 128, (1, 1, 3), (1, 1, 2), (0, 0, 0))]
    c = 16
    rb = sops.build_rulebook(ind, NB, shape, 3, 1, 1, 1, True)
    layers.append(("subm1 5(8)->16", rb, ind.shape[0], 8, 16))
    for i, (cin, cout, ks, st, pd) in enumerate(stages):
        layers.append((f"subm{i + 1} {cin}->{cin}", rb, ind.shape[0], cin, cin))
        rbs = sops.build_rulebook(ind, NB, shape, list(ks), list(st), list(pd), 1, False)
        layers.append((f"spconv{i + 1} {cin}->{cout} k{ks} s{st}", rbs, ind.shape[0], cin, cout))
        ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
        if i < 3:
            rb = sops.build_rulebook(ind, NB, shape, 3, 1, 1, 1, True)
    print(f"# tiled sparse conv sweep, dtype={args.dtype}; time = us per launch, 20 back-to-back launches behind a busy GPU (host overhead excluded)")
    for name, rb, n_in, cin, cout in layers:
        if (args.quick and cin < 32) or (args.small and cin >= 32):
            continue
        K = rb.nbr.shape[0]
        f = torch.randn(n_in, cin, device=dev).to(dt)
        w = (torch.randn(K, cin, cout, device=dev) / (K * cin) ** 0.5).to(dt)
        img = sops.make_filter_image(w.view(K, 1, 1, cin, cout))
        pairs = int((rb.nbr[:, :rb.num_out] >= 0).sum())
        gflop = 2.0 * pairs * cin * cout / 1e9
        print(f"{name:38s} rows_in={n_in:7d} rows_out={rb.num_out:7d} pairs={pairs:8d} ({gflop:6.2f} GFLOP)")
        img_bytes = img.numel() * 2
        if args.slab:
            if not (name.startswith("subm") and cin == cout and cin >= 32):
                continue
            from bevfusion_amd.spconv.fused import _variant_for
            gv = _variant_for(NB, K, cin, cout)
            med, _ = timeit(lambda: sops.sparse_conv_tiled(f, img, rb.nbr, rb.num_out, K, cin, cout, variant=gv))
            base = sops.sparse_conv_tiled(f, img, rb.nbr, rb.num_out, K, cin, cout, variant=gv)
            print(f"    gather variant {gv:5d}: {med:8.1f} us  {gflop / med * 1e3:8.1f} TFLOP/s eff")
            for bm in (128, 256):
                t, _ = timeit(lambda: sops.slab_build(rb.nbr, rb.num_out, None, bm))
                print(f"    slab_build (BM={bm}): {t:8.1f} us (once per level)")
            for v in sops.slab_variants(cin):
                meta = sops.slab_build(rb.nbr, rb.num_out, None, sops.slab_block_rows(cin, v))
                try:
                    med, _ = timeit(lambda: sops.sparse_conv_slab(f, img, meta, rb.num_out, cin, cout, variant=v))
                except RuntimeError as e:
                    print(f"    slab variant {v}: {e}")
                    continue
                out = sops.sparse_conv_slab(f, img, meta, rb.num_out, cin, cout, variant=v)
                same = bool(torch.equal(out, base))
                md = float((out.float() - base.float()).abs().max())
                print(f"    slab variant {v:5d}: {med:8.1f} us  {gflop / med * 1e3:8.1f} TFLOP/s eff   identical={same} max|diff|={md:.3g}")
                if args.cap_factor > 1.0:
                    cap = int(rb.num_out * args.cap_factor)
                    nbr_pad = torch.full((27, cap), -1, dtype=torch.int32, device=dev)
                    nbr_pad[:, :rb.num_out] = rb.nbr[:, :rb.num_out]
                    m_dev = torch.tensor([rb.num_out], dtype=torch.int32, device=dev)
                    meta_c = sops.slab_build(nbr_pad, cap, m_dev, sops.slab_block_rows(cin, v))
                    out_c = torch.empty((cap, cout), dtype=dt, device=dev)
                    med_c, _ = timeit(lambda: sops.sparse_conv_slab(f, img, meta_c, cap, cin, cout, variant=v, num_out_dev=m_dev, out=out_c))
                    print(f"         launched for {cap} rows (x{args.cap_factor:g}): {med_c:8.1f} us")
                    del nbr_pad, meta_c, out_c
            continue
        if args.ablate:
            if (cin, cout) not in ((64, 64), (128, 128)) or K != 27:
                continue
            names = {0: "full", 1: "no MFMA", 2: "no gather", 4: "no filter staging", 8: "no bpermute", 16: "no LDS fragment reads",
                     32: "no nbr preload", 3: "no MFMA+gather", 6: "no gather+staging", 7: "no MFMA+gather+staging",
                     24: "no bpermute+LDS reads", 31: "only nbr preload+epilogue+barriers", 63: "skeleton", 62: "MFMA only",
                     64: "no epilogue stores", 127: "skeleton, no stores", 255: "skeleton, no stores, no barriers",
                     191: "skeleton, no barriers"}
            for mask, nm in names.items():
                med, _ = timeit(lambda: sops.sparse_conv_tiled(f, img, rb.nbr, rb.num_out, K, cin, cout, variant=9000 + mask))
                print(f"    ablation {mask:2d} {nm:40s}: {med:8.1f} us")
            continue
        for v in (0,) + RES + STR:
            try:
                med, mn = timeit(lambda: sops.sparse_conv_tiled(f, img, rb.nbr, rb.num_out, K, cin, cout, variant=v))
            except RuntimeError:
                continue   # not built for this shape
            print(f"    variant {v:4d}: {med:8.1f} us ({mn:8.1f})  {gflop / med * 1e3:8.1f} TFLOP/s eff")
        if args.quick or args.small:
            continue
        # old kernel for comparison
        from bevfusion_amd import _capi
        prep = sops.prepare_filters(w.view(K, 1, 1, cin, cout))
        lib = _capi.load()
        out = torch.empty(rb.num_out, cout, dtype=dt, device=dev)

        def old():
            lib.bevamd_spconv_conv_forward(_capi.ptr(f), 1 if dt == torch.float16 else 2, _capi.ptr(prep), _capi.ptr(rb.nbr),
                                           rb.nbr.shape[1], rb.num_out, None, K, cin, cout, _capi.ptr(out), None, None, None,
                                           None, 0, _capi.stream_ptr(dev))
        med, mn = timeit(old)
        print(f"    r0 wave-kernel: {med:8.1f} us ({mn:8.1f})")


if __name__ == "__main__":
    main()
