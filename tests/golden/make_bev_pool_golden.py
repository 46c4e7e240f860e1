"""Generates tests/golden/bev_pool_ref_small.npz with the REFERENCE's own bev_pool kernel.

The reference has no CPU bev_pool (bev_pool_cpu.cpp is the CUDA host wrapper), so this runs on the
GPU box: oracle/_ref/bev_pool_ext/bev_pool_ext.so is the reference's extension, hipified from
/root/reference at build time by oracle/ref_build.py (sources never enter this repository).

    gpurun -- python tests/golden/make_bev_pool_golden.py gpurun_out/golden
then copy gpurun_out/golden/bev_pool_ref_small.npz into tests/golden/.

Also times the reference kernel on the flagship-size input (printed, for DESIGN.md's table).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from oracle import ref_build  # noqa: E402


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    ext = ref_build.load_ref("bev_pool_ext")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(20240807)
    B, D, H, W, C, n = 2, 2, 12, 10, 80, 6000
    coords = np.stack([rng.integers(0, H, n), rng.integers(0, W, n), rng.integers(0, D, n), rng.integers(0, B, n)], 1)
    coords[:700] = coords[0]  # one long interval
    feats = rng.standard_normal((n, C)).astype(np.float32)
    pro = oracle.bev_pool_prologue(coords.astype(np.int64), B, D, H, W)
    x = np.ascontiguousarray(feats[pro["order"]])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = ext.bev_pool_forward(t(x), t(pro["geom_sorted"]), t(pro["interval_lengths"]), t(pro["interval_starts"]),
                               B, D, H, W)
    og = rng.standard_normal((B, D, H, W, C)).astype(np.float32)
    xg = ext.bev_pool_backward(t(og), t(pro["geom_sorted"]), t(pro["interval_lengths"]), t(pro["interval_starts"]),
                               B, D, H, W)
    torch.cuda.synchronize()
    np.savez_compressed(os.path.join(out_dir, "bev_pool_ref_small.npz"), x=x, geom=pro["geom_sorted"],
                        interval_starts=pro["interval_starts"], interval_lengths=pro["interval_lengths"],
                        out=out.cpu().numpy(), out_grad=og, x_grad=xg.cpu().numpy(), B=B, D=D, H=H, W=W)
    print("wrote", os.path.join(out_dir, "bev_pool_ref_small.npz"))

    # reference kernel timing at the flagship size (sorted inputs, reference contract)
    from bevfusion_amd import synth
    from bevfusion_amd.bev_pool import bev_pool_ext

    inp = synth.bev_pool_inputs(seed=0)
    Hh, Ww, Dd = (int(v) for v in inp["nx"])
    cc, kept = oracle.bev_cell_index(inp["geom"], 1, inp["origin"], inp["dx"], inp["nx"])
    pro = oracle.bev_pool_prologue(cc[kept], 1, Dd, Hh, Ww)
    xs = t(inp["feats"][kept][pro["order"]])
    g, L, S = t(pro["geom_sorted"]), t(pro["interval_lengths"]), t(pro["interval_starts"])

    def timeit(fn, iters=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    # NOTE: the reference launches on the NULL stream; with torch's default stream that is the same stream.
    t_ref = timeit(lambda: ext.bev_pool_forward(xs, g, L, S, 1, Dd, Hh, Ww))
    t_ours = timeit(lambda: bev_pool_ext.bev_pool_forward(xs, g, L, S, 1, Dd, Hh, Ww))
    o_ref = ext.bev_pool_forward(xs, g, L, S, 1, Dd, Hh, Ww)
    o_ours = bev_pool_ext.bev_pool_forward(xs, g, L, S, 1, Dd, Hh, Ww)
    print(f"flagship sorted-input forward: reference kernel (hipified) {t_ref:.3f} ms, ours {t_ours:.3f} ms, "
          f"max|diff| {float((o_ref - o_ours).abs().max()):.3e}")
    og = torch.randn_like(o_ref)
    t_refb = timeit(lambda: ext.bev_pool_backward(og, g, L, S, 1, Dd, Hh, Ww))
    t_oursb = timeit(lambda: bev_pool_ext.bev_pool_backward(og, g, L, S, 1, Dd, Hh, Ww))
    print(f"flagship sorted-input backward: reference {t_refb:.3f} ms, ours {t_oursb:.3f} ms, equal "
          f"{bool(torch.equal(ext.bev_pool_backward(og, g, L, S, 1, Dd, Hh, Ww), bev_pool_ext.bev_pool_backward(og, g, L, S, 1, Dd, Hh, Ww)))}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/golden")
