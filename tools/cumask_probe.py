"""CU-masked HIP streams (hipExtStreamCreateWithCUMask): how do bev_pool and a wide SubM layer run on a SUBSET of the compute units,
alone and side by side?  (round 6: would partitioning the machine beat letting the two queues fight for wave slots?)
    python tools/cumask_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bevfusion_amd import synth  # noqa: E402
from bevfusion_amd.bev_pool import BevPoolPlan  # noqa: E402
from bevfusion_amd.spconv import ops as sops  # noqa: E402
from bevfusion_amd.voxel import voxelize_batch  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) that are enabled"""
    words = [0] * 8
    for b in bits:
        words[b >> 5] |= 1 << (b & 31)
    arr = (ctypes.c_uint32 * 8)(*words)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


def time_on(stream, fn, iters=10):
    with torch.cuda.stream(stream):
        for _ in range(2):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            fn()
        e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    dev = torch.device("cuda", 0)
    frames = 8
    cfg = synth.CL_CONFIG
    inp = synth.bev_pool_inputs(cfg, batch=1, seed=0, with_feats=False)
    H, W, D = (int(v) for v in inp["nx"])
    C = inp["channels"]
    geom = torch.from_numpy(inp["geom"]).to(dev).repeat(frames, 1)
    feats = torch.randn((geom.shape[0], C), device=dev, dtype=torch.float32)
    plan = BevPoolPlan.from_geometry(geom, frames, inp["origin"], inp["dx"], inp["nx"], want_intervals=True)
    bev = torch.empty((frames, D, H, W, C), dtype=torch.float32, device=dev)
    # a 64-channel SubM layer of the flagship encoder (level 3)
    pts = [torch.from_numpy(synth.lidar_points(seed=b)).to(dev) for b in range(frames)]
    vf, vc, _ = voxelize_batch(pts, cfg["voxel_size"], cfg["point_cloud_range"], cfg["max_num_points"], cfg["max_voxels"][1])
    shape = list(cfg["sparse_shape"])
    ind = vc.int().contiguous()
    for ks, st, pd in (((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1))):
        rbs = sops.build_rulebook(ind, frames, shape, list(ks), list(st), list(pd), 1, False)
        ind, shape = rbs.out_indices.contiguous(), rbs.out_spatial_shape
    rb = sops.build_rulebook(ind, frames, shape, 3, 1, 1, 1, True)
    c, v = 64, 1644228
    f = torch.randn(ind.shape[0], c, device=dev).half()
    w = (torch.randn(27, c, c, device=dev) / (27 * c) ** 0.5).half()
    img = sops.make_filter_image(w.view(27, 1, 1, c, c))
    meta = sops.slab_build(rb.nbr, rb.num_out, None, sops.slab_block_rows(c, v))
    out = torch.empty(ind.shape[0], c, device=dev, dtype=torch.float16)
    torch.cuda.synchronize()

    def pool():
        plan.launch_forward(feats, bev)

    def conv():
        sops.sparse_conv_slab(f, img, meta, rb.num_out, c, c, variant=v, out=out)

    # bit i of the mask = compute unit i / 8 of XCC i % 8 (an XCC whose bits are all clear runs UNMASKED): n CUs per XCC = the low 8 n bits
    def per_xcc(lo, hi):
        return [8 * cu + x for cu in range(lo, hi) for x in range(8)]

    g_pool, g_conv = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        pool(); conv()
    torch.cuda.synchronize()
    with torch.cuda.graph(g_pool):
        pool()
    with torch.cuda.graph(g_conv):
        conv()
    for n in (32, 24, 20, 16, 12, 8):
        st = masked_stream(per_xcc(0, n))
        print(f"{n:2d} CUs per XCC: bev_pool {time_on(st, pool):7.1f} us (as a graph {time_on(st, g_pool.replay):7.1f})   64-ch layer "
              f"{time_on(st, conv):7.1f} us (as a graph {time_on(st, g_conv.replay):7.1f})", flush=True)

    def together(sa, sb, n_conv=5):
        torch.cuda.synchronize()
        s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream()
        s0.record(cur)
        sa.wait_event(s0)
        sb.wait_event(s0)
        with torch.cuda.stream(sa):
            pool()
            ea.record()
        with torch.cuda.stream(sb):
            for _ in range(n_conv):
                conv()
            eb.record()
        cur.wait_event(ea)
        cur.wait_event(eb)
        e0.record(cur)
        e0.synchronize()
        return s0.elapsed_time(ea) * 1e3, s0.elapsed_time(eb) * 1e3, s0.elapsed_time(e0) * 1e3

    plain_a, plain_b = torch.cuda.Stream(), torch.cuda.Stream()
    for rep in range(3):
        a, b, t = together(plain_a, plain_b)
        print(f"unmasked, two streams:          bev_pool done {a:7.1f}  5 layers done {b:7.1f}  both {t:7.1f} us", flush=True)
    for n in (8, 12, 16, 20, 24):
        sa, sb = masked_stream(per_xcc(0, n)), masked_stream(per_xcc(n, 32))
        for rep in range(2):
            a, b, t = together(sa, sb)
            print(f"bev_pool on {n:2d} CUs per XCC, layers on {32 - n:2d}: bev_pool done {a:7.1f}  5 layers done {b:7.1f}  both {t:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
