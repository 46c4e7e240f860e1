"""GPU parity: HIP bev_pool (through the C ABI) vs the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): ranks / sort / interval arrays bit-exact; BEV feature sums within
1e-4 of the float64 oracle (ABS_TOL below; sums of up to ~900 N(0,1) fp32 values per cell)."""
import os

import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import _capi, synth
from bevfusion_amd.bev_pool import BevPoolPlan, QuickCumsumCuda, bev_pool, bev_pool_ext

from conftest import record_parity

pytestmark = pytest.mark.gpu

ABS_TOL = 1e-4
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _case(seed, n, B, D, H, W, c, hot=None):
    rng = np.random.default_rng(seed)
    coords = np.stack([rng.integers(0, H, n), rng.integers(0, W, n), rng.integers(0, D, n), rng.integers(0, B, n)], 1)
    if hot is not None:  # a heavy-tailed interval: `hot` rows in one cell
        coords[:hot] = coords[0]
    feats = (rng.standard_normal((n, c)) * 0.25).astype(np.float32)
    return feats, coords.astype(np.int64)


def _sorted_inputs(feats, coords, B, D, H, W, dev):
    pro = oracle.bev_pool_prologue(coords, B, D, H, W)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)
    return pro, t(feats[pro["order"]], torch.float32), t(pro["geom_sorted"], torch.int32), \
        t(pro["interval_lengths"], torch.int32), t(pro["interval_starts"], torch.int32)


CASES = [
    # n, B, D, H, W, C, hot
    (1, 1, 1, 1, 1, 4, None),
    (777, 2, 3, 5, 7, 80, None),
    (20000, 1, 1, 40, 40, 80, 3000),      # one 3000-row interval among short ones
    (5000, 3, 1, 8, 8, 64, None),
    (5000, 1, 2, 8, 8, 128, None),
    (3000, 1, 1, 6, 6, 256, None),
    (3000, 2, 1, 6, 6, 3, None),          # scalar path (C % 4 != 0)
    (3000, 1, 1, 6, 6, 1, None),
    (2000, 1, 1, 5, 5, 260, None),        # scalar path (C/4 > 64)
    (4000, 1, 1, 9, 9, 12, None),         # 3 lanes per row, 21 rows per wave instruction
    (100000, 4, 1, 64, 64, 80, 5000),
]


@pytest.mark.parametrize("n,B,D,H,W,c,hot", CASES)
def test_forward_drop_in_vs_oracle(dev, n, B, D, H, W, c, hot):
    feats, coords = _case(n + c, n, B, D, H, W, c, hot)
    pro, x, geom, lengths, starts = _sorted_inputs(feats, coords, B, D, H, W, dev)
    out = bev_pool_ext.bev_pool_forward(x, geom, lengths, starts, B, D, H, W)
    assert out.shape == (B, D, H, W, c) and out.dtype == torch.float32
    ref = oracle.bev_pool_forward_sorted(feats[pro["order"]], pro["geom_sorted"], pro["interval_starts"],
                                         pro["interval_lengths"], B, D, H, W)
    err = np.max(np.abs(out.cpu().numpy().astype(np.float64) - ref))
    record_parity("bev_pool forward vs float64 oracle (absolute; north_star bar 1e-4)", err, ABS_TOL)
    assert err <= ABS_TOL, err
    # cells no interval touches are exactly zero
    g = pro["geom_sorted"][pro["interval_starts"]]
    touched = np.zeros((B, D, H, W), bool)
    touched[g[:, 3], g[:, 2], g[:, 0], g[:, 1]] = True
    assert np.all(out.cpu().numpy()[~touched] == 0)


@pytest.mark.parametrize("n,B,D,H,W,c,hot", CASES[:8])
def test_backward_drop_in_is_bit_exact(dev, n, B, D, H, W, c, hot):
    feats, coords = _case(n + c + 1, n, B, D, H, W, c, hot)
    pro, x, geom, lengths, starts = _sorted_inputs(feats, coords, B, D, H, W, dev)
    og = np.random.default_rng(5).standard_normal((B, D, H, W, c)).astype(np.float32)
    xg = bev_pool_ext.bev_pool_backward(torch.from_numpy(og).to(dev), geom, lengths, starts, B, D, H, W)
    ref = oracle.bev_pool_backward_sorted(og, pro["geom_sorted"], pro["interval_starts"], pro["interval_lengths"], n,
                                          B, D, H, W)
    assert np.array_equal(xg.cpu().numpy(), ref)


def test_backward_zero_fills_rows_outside_intervals(dev):
    """bev_pool_cpu.cpp:78 allocates x_grad with torch::zeros; rows no interval covers stay zero."""
    B, D, H, W, c, n = 1, 1, 4, 4, 8, 100
    feats, coords = _case(9, n, B, D, H, W, c)
    pro, x, geom, lengths, starts = _sorted_inputs(feats, coords, B, D, H, W, dev)
    og = torch.ones(B, D, H, W, c, device=dev)
    xg = bev_pool_ext.bev_pool_backward(og, geom, lengths[:1], starts[:1], B, D, H, W)
    L0 = int(pro["interval_lengths"][0])
    assert torch.all(xg[:L0] == 1) and torch.all(xg[L0:] == 0)


@pytest.mark.parametrize("n,B,D,H,W", [(1, 1, 1, 1, 1), (4097, 2, 3, 5, 7), (200000, 1, 1, 360, 360),
                                       (150000, 8, 1, 90, 90), (30000, 2, 2, 16, 16)])
@pytest.mark.parametrize("i64", [True, False])
def test_prepare_ranks_sort_intervals_bit_exact(dev, n, B, D, H, W, i64):
    _, coords = _case(n, n, B, D, H, W, 1)
    pro = oracle.bev_pool_prologue(coords, B, D, H, W)
    ct = torch.from_numpy(coords if i64 else coords.astype(np.int32)).to(dev)
    plan = BevPoolPlan.from_coords(ct, B, D, H, W, want_intervals=True, want_geom=True)
    k = plan.n_intervals()
    assert k == len(pro["interval_starts"]) and plan.n_kept() == n
    cs = plan.cell_start.cpu().numpy().astype(np.int64)
    counts = np.bincount(pro["ranks"], minlength=B * D * H * W)
    assert np.array_equal(cs[1:B * D * H * W + 1] - cs[:B * D * H * W], counts) and cs[-1] == n
    assert np.array_equal(plan.ranks_sorted.cpu().numpy().astype(np.int64), pro["ranks_sorted"])
    assert np.array_equal(plan.order.cpu().numpy().astype(np.int64), pro["order"])  # stable tie order
    assert np.array_equal(plan.interval_starts[:k].cpu().numpy(), pro["interval_starts"])
    assert np.array_equal(plan.interval_lengths[:k].cpu().numpy(), pro["interval_lengths"])
    assert np.array_equal(plan.geom_sorted.cpu().numpy(), pro["geom_sorted"])


def test_prepare_drops_out_of_range_rows(dev):
    B, D, H, W, n = 2, 1, 6, 5, 5000
    rng = np.random.default_rng(0)
    coords = np.stack([rng.integers(-2, H + 2, n), rng.integers(-2, W + 2, n), rng.integers(-1, D + 1, n),
                       rng.integers(0, B, n)], 1).astype(np.int64)
    kept = (coords[:, 0] >= 0) & (coords[:, 0] < H) & (coords[:, 1] >= 0) & (coords[:, 1] < W) & \
        (coords[:, 2] >= 0) & (coords[:, 2] < D)
    pro = oracle.bev_pool_prologue(coords[kept], B, D, H, W)
    plan = BevPoolPlan.from_coords(torch.from_numpy(coords).to(dev), B, D, H, W)
    nk = plan.n_kept()
    assert nk == int(kept.sum())
    assert np.array_equal(plan.ranks_sorted[:nk].cpu().numpy().astype(np.int64), pro["ranks_sorted"])
    assert np.array_equal(plan.order[:nk].cpu().numpy(), np.nonzero(kept)[0][pro["order"]])
    feats = rng.standard_normal((n, 16)).astype(np.float32)
    out = plan.forward(torch.from_numpy(feats).to(dev))
    ref = oracle.bev_pool(feats[kept], coords[kept], B, D, H, W).transpose(0, 2, 3, 4, 1)
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= ABS_TOL


@pytest.mark.parametrize("n,B,D,H,W,c,hot", [CASES[1], CASES[2], CASES[6], CASES[10]])
def test_full_op_matches_oracle(dev, n, B, D, H, W, c, hot):
    """bev_pool(feats, coords, B, D, H, W) -> [B, C, D, H, W], unsorted inputs, like the reference."""
    feats, coords = _case(n, n, B, D, H, W, c, hot)
    out = bev_pool(torch.from_numpy(feats).to(dev), torch.from_numpy(coords).to(dev), B, D, H, W)
    assert out.shape == (B, c, D, H, W) and out.is_contiguous()
    ref = oracle.bev_pool(feats, coords, B, D, H, W)
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= ABS_TOL
    view = bev_pool(torch.from_numpy(feats).to(dev), torch.from_numpy(coords).to(dev), B, D, H, W,
                    channels_last_view=True)
    assert torch.equal(view, out)


# wave per cell (1, 2), cooperative (3-7), batched tails (14, 15 = the defaults), XCD-striped (16-19; + 100 * lines per step of
# the rotating stripe map), 2 x 4 cell tiles (23, 25)
FORWARD_VARIANTS = (1, 2, 3, 4, 5, 6, 7, 14, 15, 16, 17, 18, 19, 23, 25, 118, 219, 316, 817)
PROFILING_ONLY = {16, 17, 18, 19, 23, 25}     # rejected walks (EXPERIMENTS C.7): in -DBEVAMD_PROFILING builds only since round 6


@pytest.mark.parametrize("n,B,D,H,W,c,hot", [
    (20000, 1, 1, 40, 40, 80, 3000),      # rows of 10 groups of 4 cells: padded to 16 (SW 2) / 16 (SW 4) workgroups per line
    (9000, 3, 1, 7, 13, 80, 600),         # 13 cells per grid row: the last group of a row is ragged, 3 frames
    (9000, 2, 3, 5, 33, 64, None),        # W * D = 99 cells per row over z
    (300, 1, 1, 3, 3, 12, None),          # fewer cells than one striped line
    (6000, 8, 1, 6, 50, 256, 900),
])
@pytest.mark.parametrize("bf16", [False, True])
def test_every_forward_variant_matches_oracle_and_variant_1(dev, n, B, D, H, W, c, hot, bf16):
    """The tuned walks (bevamd_bev_pool_forward_cells_tuned) only renumber workgroups / batch the tail loads: every cell's
    rows are added in the same order, so all of them return the SAME BITS, and those are within the bar of float64."""
    feats, coords = _case(n * 3 + c, n, B, D, H, W, c, hot)
    lib = _capi.load()
    x = torch.from_numpy(feats).to(dev)
    if bf16:
        x = x.bfloat16()
    plan = BevPoolPlan.from_coords(torch.from_numpy(coords).to(dev), B, D, H, W)
    ref = oracle.bev_pool(x.float().cpu().numpy(), coords, B, D, H, W).transpose(0, 2, 3, 4, 1)
    first = None
    for v in FORWARD_VARIANTS:
        out = torch.full((B, D, H, W, c), float("nan"), dtype=torch.float32, device=dev)
        rc = lib.bevamd_bev_pool_forward_cells_tuned(_capi.ptr(x), int(bf16), _capi.ptr(plan.order), _capi.ptr(plan.cell_start),
                                                    _capi.ptr(out), plan.n, c, B, D, H, W, v, _capi.stream_ptr(dev))
        if rc == 4 and v % 100 in PROFILING_ONLY:      # BEVAMD_ERR_UNSUPPORTED: not in the shipped library
            continue
        _capi.check(rc, f"variant {v}")
        got = out.cpu().numpy()
        assert np.isfinite(got).all(), f"variant {v} left cells unwritten"
        assert np.max(np.abs(got - ref)) <= ABS_TOL, v
        if v in (1, 2) or v >= 14:     # one wave per cell: one fixed order (the cooperative flavours split a cell over waves)
            if first is None:
                first = got
            assert np.array_equal(got, first), f"variant {v} differs from variant 1"
    rc = lib.bevamd_bev_pool_forward_cells_tuned(_capi.ptr(x), int(bf16), _capi.ptr(plan.order), _capi.ptr(plan.cell_start),
                                                _capi.ptr(out), plan.n, c, B, D, H, W, 99, _capi.stream_ptr(dev))
    assert rc != 0 or c % (8 if bf16 else 4) != 0      # (the any-width scalar kernel has no variants)


def test_bf16_features(dev):
    n, B, D, H, W, c = 30000, 2, 1, 20, 20, 80
    feats, coords = _case(3, n, B, D, H, W, c, 2000)
    fb = torch.from_numpy(feats).to(dev).bfloat16()
    out = bev_pool(fb, torch.from_numpy(coords).to(dev), B, D, H, W)
    ref = oracle.bev_pool(fb.float().cpu().numpy(), coords, B, D, H, W)  # oracle on the bf16-rounded values
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= ABS_TOL
    pro, x, geom, lengths, starts = _sorted_inputs(fb.float().cpu().numpy(), coords, B, D, H, W, dev)
    out2 = bev_pool_ext.bev_pool_forward(x.bfloat16(), geom, lengths, starts, B, D, H, W)
    assert np.max(np.abs(out2.permute(0, 4, 1, 2, 3).cpu().numpy() - ref)) <= ABS_TOL


def test_autograd_through_plan_and_quickcumsum(dev):
    n, B, D, H, W, c = 4000, 2, 1, 8, 8, 16
    feats, coords = _case(11, n, B, D, H, W, c)
    x = torch.from_numpy(feats).to(dev).requires_grad_(True)
    out = bev_pool(x, torch.from_numpy(coords).to(dev), B, D, H, W)
    wgt = torch.randn_like(out)
    (out * wgt).sum().backward()
    exp = wgt.permute(0, 2, 3, 4, 1)[coords[:, 3], coords[:, 2], coords[:, 0], coords[:, 1]]
    assert torch.equal(x.grad, exp)
    # reference-shaped autograd function on pre-sorted inputs
    pro = oracle.bev_pool_prologue(coords, B, D, H, W)
    xs = torch.from_numpy(feats[pro["order"]]).to(dev).requires_grad_(True)
    o2 = QuickCumsumCuda.apply(xs, torch.from_numpy(coords[pro["order"]]).to(dev),
                               torch.from_numpy(pro["ranks_sorted"]).to(dev), B, D, H, W)
    assert torch.allclose(o2.permute(0, 4, 1, 2, 3), out, atol=1e-5)
    (o2.permute(0, 4, 1, 2, 3) * wgt).sum().backward()
    assert torch.equal(xs.grad, exp[torch.from_numpy(pro["order"]).to(dev)])


def test_empty_inputs(dev):
    out = bev_pool(torch.zeros(0, 8, device=dev), torch.zeros(0, 4, dtype=torch.long, device=dev), 1, 1, 3, 3)
    assert out.shape == (1, 8, 1, 3, 3) and torch.all(out == 0)


def test_plan_from_geometry_matches_oracle_cell_index(dev):
    """BASELINE config 1 shapes (1 camera, 64x64 BEV): geometry -> truncation -> mask -> ranks."""
    inp = synth.bev_pool_inputs(synth.LSS_SMALL_CONFIG, batch=2, channels=8, seed=1)
    coords, kept = oracle.bev_cell_index(inp["geom"], 2, inp["origin"], inp["dx"], inp["nx"])
    H, W, D = (int(v) for v in inp["nx"])
    pro = oracle.bev_pool_prologue(coords[kept], 2, D, H, W)
    plan = BevPoolPlan.from_geometry(torch.from_numpy(inp["geom"]).to(dev), 2, inp["origin"], inp["dx"], inp["nx"],
                                     want_intervals=True)
    nk = plan.n_kept()
    assert nk == int(kept.sum())
    assert np.array_equal(plan.ranks_sorted[:nk].cpu().numpy().astype(np.int64), pro["ranks_sorted"])
    k = plan.n_intervals()
    assert np.array_equal(plan.interval_lengths[:k].cpu().numpy(), pro["interval_lengths"])
    out = plan.forward(torch.from_numpy(inp["feats"]).to(dev))
    ref = oracle.bev_pool(inp["feats"][kept], coords[kept], 2, D, H, W).transpose(0, 2, 3, 4, 1)
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= ABS_TOL


def test_flagship_size_properties(dev):
    """Full C+L size (1 993 728 frustum points, C=80, 360x360): size-independent properties —
    linearity, channel-sum checksum, and the interval statistics of SURVEY.md §8d."""
    inp = synth.bev_pool_inputs(seed=0)
    H, W, D = (int(v) for v in inp["nx"])
    geom = torch.from_numpy(inp["geom"]).to(dev)
    plan = BevPoolPlan.from_geometry(geom, 1, inp["origin"], inp["dx"], inp["nx"], want_intervals=True)
    assert plan.n_kept() == 1815552 and plan.n_intervals() == 45469
    L = plan.interval_lengths[:45469]
    assert int(L.max()) == 864 and int(L.sum()) == 1815552
    x = torch.from_numpy(inp["feats"]).to(dev)
    out = plan.forward(x)
    # checksum of checksums: total mass is preserved per channel (float64 on the kept rows)
    kept_rows = plan.order[:1815552].long()
    exp = x[kept_rows].double().sum(0)
    got = out.double().sum((0, 1, 2, 3))
    assert torch.max(torch.abs(exp - got)) < 1e-2
    # linearity: pool(2x + y) == 2 pool(x) + pool(y) to rounding
    y = torch.roll(x, 1, 0)
    lhs = plan.forward(2 * x + y)
    rhs = 2 * out + plan.forward(y)
    assert torch.max(torch.abs(lhs - rhs)) < 1e-3
    # non-empty cell count
    assert int((out.abs().sum(-1) > 0).sum()) == 45469


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, "bev_pool_ref_small.npz")), reason="golden not committed")
def test_against_reference_kernel_golden(dev):
    z = np.load(os.path.join(GOLDEN, "bev_pool_ref_small.npz"))
    t = lambda a: torch.from_numpy(a).to(dev)
    B, D, H, W = (int(z[k]) for k in "BDHW")
    out = bev_pool_ext.bev_pool_forward(t(z["x"]), t(z["geom"]), t(z["interval_lengths"]), t(z["interval_starts"]),
                                        B, D, H, W)
    record_parity("bev_pool forward vs the reference kernel's own output (absolute)", np.max(np.abs(out.cpu().numpy() - z["out"])), ABS_TOL)
    assert np.max(np.abs(out.cpu().numpy() - z["out"])) <= ABS_TOL
    xg = bev_pool_ext.bev_pool_backward(t(z["out_grad"]), t(z["geom"]), t(z["interval_lengths"]),
                                        t(z["interval_starts"]), B, D, H, W)
    assert np.array_equal(xg.cpu().numpy(), z["x_grad"])


@pytest.mark.parametrize("cams,D,fh,fw,c,dtype", [(2, 7, 4, 5, 80, torch.float32), (3, 5, 3, 8, 16, torch.float32),
                                                   (1, 9, 6, 4, 64, torch.bfloat16), (6, 4, 2, 3, 8, torch.float32),
                                                   (2, 3, 5, 5, 256, torch.float32)])
def test_fused_depth_context_vs_float64(dev, cams, D, fh, fw, c, dtype):
    """Fused depth (x) context -> BEV against a float64 segment sum of the explicit outer product; random cells with
    out-of-range points (dropped), empty cells and hot cells.  Bar: 1e-4 abs (north_star)."""
    rng = np.random.default_rng(cams * 100 + D)
    B, Dz, H, W = 1, 2, 9, 11
    n = cams * D * fh * fw
    coords = np.stack([rng.integers(-1, H + 1, n), rng.integers(-1, W + 1, n), rng.integers(0, Dz, n),
                       np.zeros(n, np.int64)], 1)
    coords[: n // 3, 0], coords[: n // 3, 1] = 4, 5                  # a hot cell
    depth = rng.random(n).astype(np.float32)
    ctx = torch.from_numpy(rng.standard_normal((cams * fh * fw, c)).astype(np.float32)).to(dtype)
    plan = BevPoolPlan.from_coords(torch.from_numpy(coords).to(dev), B, Dz, H, W)
    got = plan.launch_fused(torch.from_numpy(depth).to(dev), ctx.to(dev), D, fh, fw).cpu().numpy().astype(np.float64)
    p = np.arange(n)
    cam, rem = p // (D * fh * fw), p % (D * fh * fw)
    pix = cam * fh * fw + rem % (fh * fw)
    rows = depth[:, None].astype(np.float64) * ctx.float().numpy()[pix].astype(np.float64)
    want = np.zeros((B, Dz, H, W, c))
    ok = (coords[:, 0] >= 0) & (coords[:, 0] < H) & (coords[:, 1] >= 0) & (coords[:, 1] < W)
    np.add.at(want, (coords[ok, 3], coords[ok, 2], coords[ok, 0], coords[ok, 1]), rows[ok])
    record_parity("fused depth x context pooling vs float64 (absolute; north_star bar 1e-4)", np.max(np.abs(got - want)), 1e-4)
    assert np.max(np.abs(got - want)) <= 1e-4
    # same plan, unfused op on the materialised rows: the two paths agree to rounding
    unf = plan.launch_forward(torch.from_numpy(rows.astype(np.float32)).to(dev)).cpu().numpy()
    assert np.max(np.abs(got - unf)) <= 1e-4


def test_fused_rejects_bad_inputs(dev):
    plan = BevPoolPlan.from_coords(torch.zeros((24, 4), dtype=torch.int64, device=dev), 1, 1, 2, 2)
    with pytest.raises(RuntimeError, match="fp32"):
        plan.launch_fused(torch.zeros(24, device=dev).half(), torch.zeros((6, 8), device=dev), 4, 2, 3)
    with pytest.raises(RuntimeError, match="rows x depth_bins"):
        plan.launch_fused(torch.zeros(24, device=dev), torch.zeros((5, 8), device=dev), 4, 2, 3)
    with pytest.raises(RuntimeError, match="multiple of"):
        plan.launch_fused(torch.zeros(24, device=dev), torch.zeros((6, 6), device=dev), 4, 2, 3)


def test_fused_depth_context_backward_vs_float64_and_unfused_autograd(dev):
    """d depth / d context of the fused op against float64 formulas and against autograd through the materialised
    outer product + the unfused native op (same plan)."""
    rng = np.random.default_rng(9)
    cams, D, fh, fw, c = 3, 6, 4, 5, 80
    B, Dz, H, W = 1, 1, 8, 9
    n = cams * D * fh * fw
    coords = np.stack([rng.integers(-1, H + 1, n), rng.integers(-1, W + 1, n), np.zeros(n, np.int64), np.zeros(n, np.int64)], 1)
    plan = BevPoolPlan.from_coords(torch.from_numpy(coords).to(dev), B, Dz, H, W)
    depth = torch.from_numpy(rng.random((cams, D, fh, fw)).astype(np.float32)).to(dev).requires_grad_(True)
    ctx = torch.from_numpy(rng.standard_normal((cams * fh * fw, c)).astype(np.float32)).to(dev).requires_grad_(True)
    g = torch.from_numpy(rng.standard_normal((B, Dz, H, W, c)).astype(np.float32)).to(dev)
    out = plan.fused(depth, ctx, D, fh, fw)
    out.backward(g)
    # float64 reference
    p = np.arange(n)
    pix = (p // (D * fh * fw)) * fh * fw + (p % (D * fh * fw)) % (fh * fw)
    ok = (coords[:, 0] >= 0) & (coords[:, 0] < H) & (coords[:, 1] >= 0) & (coords[:, 1] < W)
    gn, cn, dn = g.cpu().numpy().astype(np.float64), ctx.detach().cpu().numpy().astype(np.float64), depth.detach().cpu().numpy().reshape(-1).astype(np.float64)
    grow = np.zeros((n, c))
    grow[ok] = gn[0, 0, coords[ok, 0], coords[ok, 1]]
    want_dd = (grow * cn[pix]).sum(1)
    want_dc = np.zeros_like(cn)
    np.add.at(want_dc, pix, dn[:, None] * grow)
    assert np.max(np.abs(depth.grad.cpu().numpy().reshape(-1) - want_dd)) <= 1e-4
    assert np.max(np.abs(ctx.grad.cpu().numpy() - want_dc)) <= 1e-4
    # autograd through the reference formulation: outer product -> unfused op
    d2 = depth.detach().clone().requires_grad_(True)
    c2 = ctx.detach().clone().requires_grad_(True)
    rows = (d2.reshape(cams, D, fh * fw, 1) * c2.reshape(cams, 1, fh * fw, c)).reshape(n, c)
    out2 = plan.forward(rows)
    out2.backward(g)
    assert float((out - out2).detach().abs().max()) <= 1e-4
    assert float((depth.grad - d2.grad).abs().max()) <= 1e-4 and float((ctx.grad - c2.grad).abs().max()) <= 1e-4
