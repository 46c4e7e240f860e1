"""Column formulation of the fused depth (x) context -> BEV pooling (csrc/bev_pool_fused_cols.hip, round 4) through the C ABI.

Reference op: models/vtransforms/depth_lss.py:92-97 (outer product) + base.py:141-176 (range mask, bev_pool).  Bar: <= 1e-4 abs
against a float64 segment sum of the explicit outer product (BASELINE.json north_star), and agreement with the cell-centric
kernels on the same plan.  Covered: the plan (row masks, run numbering, frame-major CSR) against numpy; random cells (every row
its own run: the long-tailed worst case, forced), camera geometry with pitched / rolled cameras and rotated / flipped image
augmentation (a few runs per column), several frames, depth halves (D > 60), bf16 context, empty plans, whole-column drops,
graph capture."""
import math

import numpy as np
import pytest
import torch

from bevfusion_amd import _capi, synth
from bevfusion_amd.bev_pool import BevPoolPlan

pytestmark = pytest.mark.gpu


def float64_reference(depth, ctx, coords, cams, D, fh, fw, B, Dz, H, W):
    n = cams * D * fh * fw
    p = np.arange(n)
    pix = (p // (D * fh * fw)) * fh * fw + (p % (D * fh * fw)) % (fh * fw)
    rows = depth.reshape(-1, 1).astype(np.float64) * ctx.astype(np.float64)[pix]
    ok = ((coords[:, 0] >= 0) & (coords[:, 0] < H) & (coords[:, 1] >= 0) & (coords[:, 1] < W) & (coords[:, 2] >= 0)
          & (coords[:, 2] < Dz))
    want = np.zeros((B, Dz, H, W, ctx.shape[1]))
    np.add.at(want, (coords[ok, 3], coords[ok, 2], coords[ok, 0], coords[ok, 1]), rows[ok])
    return want, ok


@pytest.mark.parametrize("cams,D,fh,fw,c,dtype", [(2, 7, 4, 8, 80, torch.float32), (3, 5, 3, 4, 16, torch.float32),
                                                   (1, 9, 32, 12, 64, torch.bfloat16), (6, 61, 5, 8, 8, torch.float32),
                                                   (2, 118, 6, 4, 80, torch.float32), (1, 3, 1, 4, 256, torch.float32)])
def test_random_cells_every_row_its_own_run(dev, cams, D, fh, fw, c, dtype):
    """Random cell per point: runs are single rows (plus hot cells and dropped points) — the column plan's worst case, forced."""
    rng = np.random.default_rng(cams * 1000 + D * 10 + fh)
    B, Dz, H, W = 1, 2, 9, 11
    n = cams * D * fh * fw
    coords = np.stack([rng.integers(-1, H + 1, n), rng.integers(-1, W + 1, n), rng.integers(0, Dz, n), np.zeros(n, np.int64)], 1)
    hot = min(n // 3, 600)                                                         # a hot cell: long runs in the first columns
    coords[:hot, 0], coords[:hot, 1], coords[:hot, 2] = 4, 5, 1                     # (flagship cells hold up to ~900 points of weight <= 1/118)
    depth = rng.random(n).astype(np.float32)
    ctx = torch.from_numpy(rng.standard_normal((cams * fh * fw, c)).astype(np.float32)).to(dtype)
    plan = BevPoolPlan.from_coords(torch.from_numpy(coords).to(dev), B, Dz, H, W)
    got = plan.launch_fused(torch.from_numpy(depth).to(dev), ctx.to(dev), D, fh, fw, mode="columns!").cpu().numpy()
    want, ok = float64_reference(depth, ctx.float().numpy(), coords, cams, D, fh, fw, B, Dz, H, W)
    assert np.max(np.abs(got - want)) <= 1e-4
    cells = plan.launch_fused(torch.from_numpy(depth).to(dev), ctx.to(dev), D, fh, fw, mode="cells").cpu().numpy()
    assert np.max(np.abs(got - cells)) <= 1e-4
    # auto mode keeps the cell-centric kernel for such a plan (as many runs as points): same bits as "cells"
    auto = plan.launch_fused(torch.from_numpy(depth).to(dev), ctx.to(dev), D, fh, fw).cpu().numpy()
    cols = plan.fused_columns(D, fh, fw, c, force=True)
    if cols.nruns > 0.25 * max(cols.n_kept, 1):
        assert np.array_equal(auto, cells)


def test_column_plan_against_numpy(dev):
    """keep / end masks, run numbering, slot_of_run (stable order by frame-major cell) and the CSR, on two frames."""
    rng = np.random.default_rng(5)
    B, cams_per, D, fh, fw = 2, 2, 5, 7, 4
    Dz, H, W = 1, 6, 5
    cams = B * cams_per
    n = cams * D * fh * fw
    # column-structured cells: a few cells per column, with dropped stretches
    coords = np.zeros((n, 4), np.int64)
    idx = np.arange(n).reshape(cams, D, fh, fw)
    for cam in range(cams):
        for d in range(D):
            for w in range(fw):
                cuts = np.sort(rng.integers(0, fh + 1, 2))
                cell = rng.integers(-1, H + 1, 3), rng.integers(0, W, 3)
                for h in range(fh):
                    seg = int(h >= cuts[0]) + int(h >= cuts[1])
                    coords[idx[cam, d, h, w]] = (cell[0][seg], cell[1][seg], 0, cam // cams_per)
    plan = BevPoolPlan.from_coords(torch.from_numpy(coords).to(dev), B, Dz, H, W)
    cols = plan.fused_columns(D, fh, fw, 80, force=True)
    ok = (coords[:, 0] >= 0) & (coords[:, 0] < H)
    rank = coords[:, 0] * (W * Dz * B) + coords[:, 1] * (Dz * B) + coords[:, 2] * B + coords[:, 3]
    per_frame = Dz * H * W
    fkey = (rank % B) * per_frame + rank // B
    keep = np.zeros(cams * D * fw, np.uint32)
    end = np.zeros(cams * D * fw, np.uint32)
    run_keys = []
    for cam in range(cams):
        for d in range(D):
            for w in range(fw):
                col = (cam * D + d) * fw + w
                for h in range(fh):
                    p = idx[cam, d, h, w]
                    if not ok[p]:
                        continue
                    keep[col] |= 1 << h
                    nxt = idx[cam, d, h + 1, w] if h + 1 < fh else None
                    if nxt is None or not ok[nxt] or rank[nxt] != rank[p]:
                        end[col] |= 1 << h
                        run_keys.append(fkey[p])
    run_keys = np.array(run_keys)
    assert np.array_equal(cols.keep.cpu().numpy().view(np.uint32), keep)
    assert np.array_equal(cols.end.cpu().numpy().view(np.uint32), end)
    nr = np.array([bin(int(e)).count("1") for e in end])
    assert np.array_equal(cols.run_first.cpu().numpy(), np.concatenate([[0], np.cumsum(nr)[:-1]]))
    assert cols.nruns == run_keys.shape[0] == int(nr.sum())
    order = np.argsort(run_keys, kind="stable")
    slot = np.empty_like(order)
    slot[order] = np.arange(order.shape[0])
    start = np.searchsorted(run_keys[order], np.arange(B * per_frame + 1), side="left")
    assert np.array_equal(cols.prow_start.cpu().numpy(), start)
    # round 5: a run that is alone in its cell carries (1 << 31 | row of out [b, z, x, y]) instead of its slot
    cell_of_run = run_keys                                            # frame-major: b * per_frame + (x * W + y) * Dz + z
    alone = (start[cell_of_run + 1] - start[cell_of_run]) == 1
    b_, local = cell_of_run // per_frame, cell_of_run % per_frame
    gz, gy, gx = local % Dz, (local // Dz) % W, local // (Dz * W)
    out_row = ((b_ * Dz + gz) * H + gx) * W + gy
    want = np.where(alone, (1 << 31) | out_row, slot).astype(np.uint32)
    assert alone.any() and (~alone).any()
    assert np.array_equal(cols.slot_of_run.cpu().numpy()[: cols.nruns].view(np.uint32), want)


rigged_geometry = synth.rigged_geometry   # moved into the package: bench.py times the fused pooling on such a rig too


@pytest.mark.parametrize("B,n_cam,D,fh,fw,c,dtype,aug", [(1, 6, 118, 32, 88, 80, torch.float32, True),
                                                         (2, 3, 40, 32, 44, 80, torch.float32, True),
                                                         (1, 2, 118, 16, 88, 64, torch.bfloat16, True),
                                                         (3, 1, 59, 32, 88, 80, torch.float32, False)])
def test_camera_geometry_with_pitched_cameras_and_rotated_augmentation(dev, B, n_cam, D, fh, fw, c, dtype, aug):
    cfg = synth.CL_CONFIG
    geom = rigged_geometry(B, n_cam, D, fh, fw, seed=B * 10 + n_cam, rot_deg=5.4 if aug else 0.0, flip=aug)
    dx, bx, nx = synth.gen_dx_bx(cfg["xbound"], cfg["ybound"], cfg["zbound"])
    origin = (bx - dx / np.float32(2.0)).astype(np.float32)
    H, W, Dz = (int(v) for v in nx)
    g = torch.from_numpy(geom.reshape(-1, 3)).to(dev)
    plan = BevPoolPlan.from_geometry(g, B, origin, dx, nx)
    rng = np.random.default_rng(3)
    cams = B * n_cam
    depth = torch.softmax(torch.from_numpy(rng.standard_normal((cams, D, fh, fw)).astype(np.float32)), 1).numpy()
    ctx = torch.from_numpy((rng.standard_normal((cams * fh * fw, c)) * 0.5).astype(np.float32)).to(dtype)
    cols = plan.fused_columns(D, fh, fw, c)
    assert cols is not None, "a camera rig must take the column formulation in auto mode"
    n_kept = plan.n_kept()
    assert cols.nruns < 0.25 * n_kept                     # a few runs per column, ~fh points per run
    got = plan.launch_fused(torch.from_numpy(depth).reshape(-1).to(dev), ctx.to(dev), D, fh, fw)
    # float64 reference with the reference's own cell arithmetic (fp32 subtract / divide / truncation, base.py:149-169)
    cell = ((geom.reshape(-1, 3) - origin) / dx).astype(np.int64)
    bidx = np.repeat(np.arange(B), geom.reshape(-1, 3).shape[0] // B)
    coords = np.concatenate([cell, bidx[:, None]], 1)
    want, ok = float64_reference(depth, ctx.float().numpy(), coords, cams, D, fh, fw, B, Dz, H, W)
    assert int(ok.sum()) == n_kept
    err = float(np.max(np.abs(got.cpu().numpy() - want)))
    assert err <= 1e-4, err
    cells = plan.launch_fused(torch.from_numpy(depth).reshape(-1).to(dev), ctx.to(dev), D, fh, fw, mode="cells")
    assert float((got - cells).abs().max()) <= 1e-4
    # deterministic: a second launch gives the same bits
    again = plan.launch_fused(torch.from_numpy(depth).reshape(-1).to(dev), ctx.to(dev), D, fh, fw)
    assert torch.equal(got, again)


def test_empty_plans_and_dropped_columns(dev):
    B, Dz, H, W = 1, 1, 4, 4
    cams, D, fh, fw, c = 1, 4, 3, 4, 8
    n = cams * D * fh * fw
    coords = np.full((n, 4), -5, np.int64)
    coords[:, 2:] = 0
    depth = torch.rand(n, device=dev)
    ctx = torch.randn(cams * fh * fw, c, device=dev)
    plan = BevPoolPlan.from_coords(torch.from_numpy(coords).to(dev), B, Dz, H, W)       # nothing survives the range mask
    out = plan.launch_fused(depth, ctx, D, fh, fw, mode="columns!")
    assert plan.fused_columns(D, fh, fw, c, force=True).nruns == 0 and not out.any()
    coords[:, :2] = 1
    coords[np.arange(n) % fw != 2, 0] = -1                                                 # only image column w = 2 survives
    plan = BevPoolPlan.from_coords(torch.from_numpy(coords).to(dev), B, Dz, H, W)
    out = plan.launch_fused(depth, ctx, D, fh, fw, mode="columns!")
    want, _ = float64_reference(depth.cpu().numpy(), ctx.cpu().numpy(), coords, cams, D, fh, fw, B, Dz, H, W)
    assert np.max(np.abs(out.cpu().numpy() - want)) <= 1e-5
    assert plan.fused_columns(D, fh, fw, c, force=True).nruns == D


def test_unsupported_shapes_fall_back_or_raise(dev):
    lib = _capi.load()
    assert lib.bevamd_bev_pool_fused_columns_supported(80, 118, 32, 88) == 1
    assert lib.bevamd_bev_pool_fused_columns_supported(80, 118, 33, 88) == 0      # more rows than mask bits
    assert lib.bevamd_bev_pool_fused_columns_supported(80, 118, 32, 86) == 0      # fw % 4
    assert lib.bevamd_bev_pool_fused_columns_supported(6, 118, 32, 88) == 0       # c % 4
    n = 2 * 3 * 2 * 5
    plan = BevPoolPlan.from_coords(torch.zeros((n, 4), dtype=torch.int64, device=dev), 1, 1, 2, 2)
    depth, ctx = torch.rand(n, device=dev), torch.randn(2 * 2 * 5, 8, device=dev)
    with pytest.raises(RuntimeError, match="column formulation"):
        plan.launch_fused(depth, ctx, 3, 2, 5, mode="columns!")
    out = plan.launch_fused(depth, ctx, 3, 2, 5)                                   # auto: the cell-centric kernel
    assert torch.allclose(out[0, 0, 0, 0], (depth.view(2, 3, 10, 1) * ctx.view(2, 1, 10, 8)).sum((0, 1, 2)), atol=1e-4)


def test_flagship_tile_keeps_two_workgroups_per_compute_unit(dev):
    """Pass 1's tile (context rows + depth tile + run metadata of 4 image columns x 60 depth bins) is sized so that TWO workgroups
    share a compute unit's 160 KB of LDS; a third metadata array once pushed it to one (round 5) — the runtime's occupancy
    calculator is the witness."""
    lib = _capi.load()
    assert lib.bevamd_bev_pool_fused_columns_occupancy(80, 118, 32, 88) >= 2
    assert lib.bevamd_bev_pool_fused_columns_occupancy(80, 118, 33, 88) < 0       # unsupported shape: an error code


def test_column_path_replays_from_a_graph(dev):
    B, n_cam, D, fh, fw, c = 1, 2, 24, 32, 44, 80
    cfg = synth.CL_CONFIG
    geom = rigged_geometry(B, n_cam, D, fh, fw, seed=4)
    dx, bx, nx = synth.gen_dx_bx(cfg["xbound"], cfg["ybound"], cfg["zbound"])
    origin = (bx - dx / np.float32(2.0)).astype(np.float32)
    plan = BevPoolPlan.from_geometry(torch.from_numpy(geom.reshape(-1, 3)).to(dev), B, origin, dx, nx)
    depth = torch.rand(B * n_cam * D * fh * fw, device=dev)
    ctx = torch.randn(B * n_cam * fh * fw, c, device=dev)
    plan.prepare_fused(D, fh, fw, c)
    H, W, Dz = (int(v) for v in nx)
    out = torch.empty((B, Dz, H, W, c), device=dev)
    ref = plan.launch_fused(depth, ctx, D, fh, fw).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        plan.launch_fused(depth, ctx, D, fh, fw, out=out)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        plan.launch_fused(depth, ctx, D, fh, fw, out=out, mode="columns!")
    out.zero_()
    graph.replay()
    assert torch.equal(out, ref)
    depth.mul_(0.5)
    graph.replay()
    assert float((out - ref * 0.5).abs().max()) <= 1e-5


@pytest.mark.parametrize("B,n_cam,D,fh,fw,c,aug", [(1, 3, 118, 32, 88, 80, True), (2, 2, 40, 16, 44, 64, True), (1, 1, 59, 32, 88, 80, False)])
def test_column_backward_vs_point_kernels_and_float64(dev, B, n_cam, D, fh, fw, c, aug):
    """d depth / d context through the column masks (bevamd_bev_pool_fused_backward_columns) against the point-wise kernels of the
    same plan and against float64 formulas; a pitched / rolled rig with rotated augmentation gives several runs per column."""
    from bevfusion_amd import bev_pool as bp

    cfg = synth.CL_CONFIG
    geom = rigged_geometry(B, n_cam, D, fh, fw, seed=21 + n_cam, rot_deg=5.4 if aug else 0.0, flip=aug)
    dx, bx, nx = synth.gen_dx_bx(cfg["xbound"], cfg["ybound"], cfg["zbound"])
    origin = (bx - dx / np.float32(2.0)).astype(np.float32)
    H, W, Dz = (int(v) for v in nx)
    plan = BevPoolPlan.from_geometry(torch.from_numpy(geom.reshape(-1, 3)).to(dev), B, origin, dx, nx)
    cams = B * n_cam
    g = torch.Generator(device=dev).manual_seed(5)
    depth = torch.softmax(torch.randn((cams, D, fh, fw), generator=g, device=dev), 1)
    ctx = torch.randn((cams * fh * fw, c), generator=g, device=dev) * 0.5
    gout = torch.randn((B, Dz, H, W, c), generator=g, device=dev)
    assert plan.fused_columns(D, fh, fw, c) is not None
    dd_c, dc_c = plan.launch_fused_backward(gout, depth, ctx, D, fh, fw)
    assert tuple(dd_c.shape) == tuple(depth.shape) and dd_c.stride(-1) != 1      # the transposed buffer, viewed in place
    old, bp._FUSED_MODE = bp._FUSED_MODE, "cells"
    try:
        dd_p, dc_p = plan.launch_fused_backward(gout, depth, ctx, D, fh, fw)
    finally:
        bp._FUSED_MODE = old
    assert float((dd_c - dd_p).abs().max()) <= 1e-4 * (1 + float(dd_p.abs().max()))
    assert float((dc_c - dc_p).abs().max()) <= 1e-4 * (1 + float(dc_p.abs().max()))
    # float64: cell of every point by the reference's arithmetic
    cell = ((geom.reshape(-1, 3) - origin) / dx).astype(np.int64)
    ok = ((cell >= 0) & (cell < np.array([H, W, Dz]))).all(1)
    bidx = np.repeat(np.arange(B), cell.shape[0] // B)
    gn = gout.cpu().numpy().astype(np.float64)
    grow = np.zeros((cell.shape[0], c))
    grow[ok] = gn[bidx[ok], cell[ok, 2], cell[ok, 0], cell[ok, 1]]
    p = np.arange(cell.shape[0])
    pix = (p // (D * fh * fw)) * fh * fw + (p % (D * fh * fw)) % (fh * fw)
    cn = ctx.cpu().numpy().astype(np.float64)
    want_dd = (grow * cn[pix]).sum(1).reshape(cams, D, fh, fw)
    want_dc = np.zeros_like(cn)
    np.add.at(want_dc, pix, depth.cpu().numpy().reshape(-1, 1).astype(np.float64) * grow)
    assert np.max(np.abs(dd_c.cpu().numpy() - want_dd)) <= 1e-4 * (1 + np.abs(want_dd).max())
    assert np.max(np.abs(dc_c.cpu().numpy() - want_dc)) <= 1e-4 * (1 + np.abs(want_dc).max())
    # autograd end to end on a [B, N, D, fH, fW] depth tensor (what the view transform passes)
    d5 = depth.view(B, n_cam, D, fh, fw).clone().requires_grad_(True)
    c2 = ctx.clone().requires_grad_(True)
    plan.fused(d5, c2, D, fh, fw).backward(gout)
    assert torch.equal(d5.grad.view(cams, D, fh, fw), dd_c.contiguous()) and torch.equal(c2.grad, dc_c)


def test_column_backward_with_more_runs_than_fit_at_once(dev):
    """Every row its own run (random cells): 16 x 8 = 128 ... 59 x 16 = 944 runs per column, several resident passes."""
    rng = np.random.default_rng(2)
    for cams, D, fh, fw, c in ((2, 16, 8, 4, 16), (1, 59, 16, 8, 80)):
        B, Dz, H, W = 1, 1, 9, 11
        n = cams * D * fh * fw
        coords = np.stack([rng.integers(-1, H + 1, n), rng.integers(-1, W + 1, n), np.zeros(n, np.int64), np.zeros(n, np.int64)], 1)
        plan = BevPoolPlan.from_coords(torch.from_numpy(coords).to(dev), B, Dz, H, W)
        depth = torch.from_numpy(rng.random((cams, D, fh, fw)).astype(np.float32)).to(dev)
        ctx = torch.from_numpy(rng.standard_normal((cams * fh * fw, c)).astype(np.float32)).to(dev)
        gout = torch.from_numpy(rng.standard_normal((B, Dz, H, W, c)).astype(np.float32)).to(dev)
        cols = plan.fused_columns(D, fh, fw, c, force=True)
        lib = _capi.load()
        dd_t = torch.empty((cams, fw, D, fh), device=dev)
        dc = torch.empty_like(ctx)
        rc = lib.bevamd_bev_pool_fused_backward_columns(_capi.ptr(gout), _capi.ptr(depth), _capi.ptr(ctx), _capi.ptr(cols.keep),
                                                        _capi.ptr(cols.end), _capi.ptr(plan.cell_of_point()), _capi.ptr(dd_t),
                                                        _capi.ptr(dc), n, c, D, fh, fw, B, Dz, H, W, _capi.stream_ptr(dev))
        _capi.check(rc, "bev_pool_fused_backward_columns")
        from bevfusion_amd import bev_pool as bp
        old, bp._FUSED_MODE = bp._FUSED_MODE, "cells"
        try:
            dd_p, dc_p = plan.launch_fused_backward(gout, depth, ctx, D, fh, fw)
        finally:
            bp._FUSED_MODE = old
        assert float((dd_t.permute(0, 2, 3, 1) - dd_p).abs().max()) <= 1e-4
        assert float((dc - dc_p).abs().max()) <= 1e-4 * (1 + float(dc_p.abs().max()))


def test_column_backward_with_dropped_columns_and_poisoned_lds(dev):
    """an image column without a kept point stages no gradient rows; its d_ctx items used to read LDS row 0 with weight 0
    (0 * stale NaN = NaN).  Every CU's LDS is filled with NaN bits first; whole columns and whole cameras are dropped."""
    lib = _capi.load()
    assert lib.bevamd_bev_pool_fused_backward_columns_supported(80, 118, 32, 88) == 1
    assert lib.bevamd_bev_pool_fused_backward_columns_supported(256, 118, 32, 88) == 0       # fh * c / 4 > 1024 items
    assert lib.bevamd_bev_pool_fused_backward_columns_supported(80, 3000, 32, 88) == 0       # >= 65535 runs per column
    rng = np.random.default_rng(11)
    B, Dz, H, W = 1, 1, 6, 7
    cams, D, fh, fw, c = 2, 9, 8, 8, 32
    n = cams * D * fh * fw
    coords = np.stack([rng.integers(0, H, n), rng.integers(0, W, n), np.zeros(n, np.int64), np.zeros(n, np.int64)], 1)
    wcol = np.arange(n) % fw
    coords[(wcol == 1) | (wcol == 6), 0] = -3                 # image columns w = 1, 6 dropped in both cameras
    coords[n // 2:, 1] = W + 2                                # ... and the whole second camera
    plan = BevPoolPlan.from_coords(torch.from_numpy(coords).to(dev), B, Dz, H, W)
    depth = torch.from_numpy(rng.random((cams, D, fh, fw)).astype(np.float32)).to(dev)
    ctx = torch.from_numpy(rng.standard_normal((cams * fh * fw, c)).astype(np.float32)).to(dev)
    gout = torch.from_numpy(rng.standard_normal((B, Dz, H, W, c)).astype(np.float32)).to(dev)
    cols = plan.fused_columns(D, fh, fw, c, force=True)
    for pattern in (0x7FC00000, 0x7F800000, 0xFFFFFFFF):      # NaN, +Inf, NaN with every bit set
        dd_t = torch.full((cams, fw, D, fh), 7.0, device=dev)
        dc = torch.full_like(ctx, 7.0)
        _capi.check(lib.bevamd_debug_lds_poison(pattern, _capi.stream_ptr(dev)), "lds_poison")
        rc = lib.bevamd_bev_pool_fused_backward_columns(_capi.ptr(gout), _capi.ptr(depth), _capi.ptr(ctx), _capi.ptr(cols.keep),
                                                        _capi.ptr(cols.end), _capi.ptr(plan.cell_of_point()), _capi.ptr(dd_t),
                                                        _capi.ptr(dc), n, c, D, fh, fw, B, Dz, H, W, _capi.stream_ptr(dev))
        _capi.check(rc, "bev_pool_fused_backward_columns")
        assert bool(torch.isfinite(dc).all()) and bool(torch.isfinite(dd_t).all())
        ok = ((coords[:, 0] >= 0) & (coords[:, 1] < W))
        gn = gout.cpu().numpy().astype(np.float64)
        grow = np.zeros((n, c))
        grow[ok] = gn[0, 0, coords[ok, 0], coords[ok, 1]]
        p = np.arange(n)
        pix = (p // (D * fh * fw)) * fh * fw + (p % (D * fh * fw)) % (fh * fw)
        cn = ctx.cpu().numpy().astype(np.float64)
        want_dd = (grow * cn[pix]).sum(1).reshape(cams, D, fh, fw)
        want_dc = np.zeros_like(cn)
        np.add.at(want_dc, pix, depth.cpu().numpy().reshape(-1, 1).astype(np.float64) * grow)
        assert np.max(np.abs(dd_t.permute(0, 2, 3, 1).cpu().numpy() - want_dd)) <= 1e-4
        assert np.max(np.abs(dc.cpu().numpy() - want_dc)) <= 1e-4
        dropped = want_dc.reshape(cams, fh, fw, c)[:, :, [1, 6]]
        assert not dropped.any() and not dc.view(cams, fh, fw, c)[:, :, [1, 6]].any() and not dc.view(cams, fh, fw, c)[1].any()


@pytest.mark.parametrize("B,seed", [(2, 0), (4, 7)])
def test_per_step_training_plan_equals_the_oracle(dev, B, seed):
    """What `bench.py --mode train-step` rebuilds inside every timed step (round 6): that step's augmentation
    matrices (synth.training_augmentation = ImageAug3D + GlobalRotScaleTrans as the reference samples them,
    transforms_3d.py:85-165, 196-230) -> get_geometry on the device (base.py:92-135) -> rank / sort / CSR plan -> column plan ->
    pooled BEV — against a float64 segment sum of the explicit outer product with the reference's cell arithmetic
    (base.py:149-169) on the SAME geometry, and the materialised-volume op on that plan against the float64 oracle."""
    import oracle
    from bevfusion_amd.vtransforms import DepthLSSTransform

    cfg = synth.CL_CONFIG
    n_cam, (fh, fw), c = cfg["num_cameras"], cfg["feature_size"], 80
    vt = DepthLSSTransform(256, c, cfg["image_size"], cfg["feature_size"], cfg["xbound"], cfg["ybound"], cfg["zbound"], cfg["dbound"],
                           downsample=2).to(dev).eval()
    rig = synth.camera_rig(n_cam)
    a = synth.training_augmentation(np.random.default_rng(seed), B, n_cam, cfg)
    t = lambda v: torch.from_numpy(v).to(dev)
    with torch.no_grad():
        gm = vt.get_geometry(t(np.tile(rig["camera2lidar_rots"], (B, 1, 1, 1))), t(np.tile(rig["camera2lidar_trans"], (B, 1, 1))),
                             t(np.tile(rig["intrins"], (B, 1, 1, 1))), t(a["post_rots"]), t(a["post_trans"]),
                             extra_rots=t(a["extra_rots"]), extra_trans=t(a["extra_trans"]))
    D = gm.shape[2]
    dx, bx, nx = synth.gen_dx_bx(cfg["xbound"], cfg["ybound"], cfg["zbound"])
    origin = (bx - dx / np.float32(2.0)).astype(np.float32)
    H, W, Dz = (int(v) for v in nx)
    plan = BevPoolPlan.from_geometry(gm.view(-1, 3), B, origin, dx, nx)
    plan.prepare_fused(D, fh, fw, c)
    geom = gm.cpu().numpy().reshape(-1, 3)
    cell = ((geom - origin) / dx).astype(np.int64)                       # fp32 subtract / divide, truncation like .long()
    coords = np.concatenate([cell, np.repeat(np.arange(B), geom.shape[0] // B)[:, None]], 1)
    rng = np.random.default_rng(seed + 1)
    cams = B * n_cam
    depth = torch.softmax(torch.from_numpy(rng.standard_normal((cams, D, fh, fw)).astype(np.float32)), 1).numpy()
    ctx = (rng.standard_normal((cams * fh * fw, c)) * 0.5).astype(np.float32)
    want, ok = float64_reference(depth, ctx, coords, cams, D, fh, fw, B, Dz, H, W)
    assert int(ok.sum()) == plan.n_kept() and 0.5 * geom.shape[0] < plan.n_kept() <= geom.shape[0]
    got = plan.launch_fused(torch.from_numpy(depth).reshape(-1).to(dev), torch.from_numpy(ctx).to(dev), D, fh, fw)
    err = float(np.max(np.abs(got.cpu().numpy() - want)))
    assert err <= 1e-4, err
    # the augmented rig crosses cells inside a column: more runs per column than the unaugmented test-time rig, still the column kernel
    cols = plan.fused_columns(D, fh, fw, c, build=False)
    assert cols is not None and cols.nruns / (cams * D * fw) > 1.0
    # the API-level op on the materialised volume of one frame of that plan
    feats = (rng.standard_normal((geom.shape[0], 8)) * 0.25).astype(np.float32)
    planm = BevPoolPlan.from_geometry(gm.view(-1, 3), B, origin, dx, nx)
    outm = planm.forward(torch.from_numpy(feats).to(dev)).cpu().numpy()
    refm = oracle.bev_pool(feats[ok], coords[ok], B, Dz, H, W).transpose(0, 2, 3, 4, 1)
    assert float(np.max(np.abs(outm - refm))) <= 1e-4
