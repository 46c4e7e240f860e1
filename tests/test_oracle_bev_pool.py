"""CPU: the bev_pool oracle (oracle/bev_pool_oracle.c) against independent numpy formulations
and against the reference's only device-agnostic algorithm, QuickCumsum (bev_pool.py:8-34)."""
import os

import numpy as np
import pytest

import oracle
from bevfusion_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _random_case(rng, n, B, D, H, W, c):
    coords = np.stack([rng.integers(0, H, n), rng.integers(0, W, n), rng.integers(0, D, n), rng.integers(0, B, n)], 1)
    feats = rng.standard_normal((n, c)).astype(np.float32)
    return feats, coords.astype(np.int64)


def _numpy_bev_pool(feats, coords, B, D, H, W):
    out = np.zeros((B, D, H, W, feats.shape[1]), dtype=np.float64)
    np.add.at(out, (coords[:, 3], coords[:, 2], coords[:, 0], coords[:, 1]), feats.astype(np.float64))
    return out.transpose(0, 4, 1, 2, 3)


@pytest.mark.parametrize("n,B,D,H,W,c", [(1, 1, 1, 1, 1, 1), (500, 2, 3, 7, 5, 8), (20000, 1, 1, 40, 40, 80),
                                          (3000, 4, 2, 9, 11, 3)])
def test_oracle_matches_index_add(n, B, D, H, W, c):
    rng = np.random.default_rng(n)
    feats, coords = _random_case(rng, n, B, D, H, W, c)
    got = oracle.bev_pool(feats, coords, B, D, H, W)
    ref = _numpy_bev_pool(feats, coords, B, D, H, W)
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-9)


def test_rank_formula_and_intervals():
    rng = np.random.default_rng(1)
    B, D, H, W = 3, 2, 5, 4
    _, coords = _random_case(rng, 1000, B, D, H, W, 1)
    pro = oracle.bev_pool_prologue(coords, B, D, H, W)
    exp = coords[:, 0] * (W * D * B) + coords[:, 1] * (D * B) + coords[:, 2] * B + coords[:, 3]
    assert np.array_equal(pro["ranks"], exp)
    rs = pro["ranks_sorted"]
    assert np.all(rs[1:] >= rs[:-1])
    uniq, first, counts = np.unique(rs, return_index=True, return_counts=True)
    assert np.array_equal(pro["interval_starts"], first.astype(np.int32))
    assert np.array_equal(pro["interval_lengths"], counts.astype(np.int32))
    # stable: inside an interval the original row indices ascend
    o = pro["order"]
    same = rs[1:] == rs[:-1]
    assert np.all(o[1:][same] > o[:-1][same])


def test_quickcumsum_agrees_with_oracle():
    """QuickCumsum (bev_pool.py:8-34: cumsum, boundary mask, difference) == interval sums."""
    import torch

    rng = np.random.default_rng(2)
    B, D, H, W, c = 1, 1, 16, 16, 8
    feats, coords = _random_case(rng, 5000, B, D, H, W, c)
    pro = oracle.bev_pool_prologue(coords, B, D, H, W)
    x = torch.from_numpy(feats[pro["order"]]).double()
    ranks = torch.from_numpy(pro["ranks_sorted"])
    xc = x.cumsum(0)
    kept = torch.ones(x.shape[0], dtype=torch.bool)
    kept[:-1] = ranks[1:] != ranks[:-1]
    xs = xc[kept]
    xs = torch.cat((xs[:1], xs[1:] - xs[:-1]))
    g = torch.from_numpy(pro["geom_sorted"])[kept].long()
    dense = torch.zeros(B, D, H, W, c, dtype=torch.float64)
    dense[g[:, 3], g[:, 2], g[:, 0], g[:, 1]] = xs
    got = oracle.bev_pool(feats, coords, B, D, H, W)
    np.testing.assert_allclose(got, dense.permute(0, 4, 1, 2, 3).numpy(), atol=1e-9)


def test_f32_sequential_variant_close_to_f64():
    rng = np.random.default_rng(3)
    feats, coords = _random_case(rng, 30000, 1, 1, 10, 10, 16)
    a = oracle.bev_pool(feats, coords, 1, 1, 10, 10, dtype=np.float64)
    b = oracle.bev_pool(feats, coords, 1, 1, 10, 10, dtype=np.float32)
    assert np.max(np.abs(a - b)) < 1e-3  # ~300 adds per cell of N(0,1) values


def test_backward_is_broadcast():
    rng = np.random.default_rng(4)
    B, D, H, W, c = 2, 1, 6, 6, 4
    _, coords = _random_case(rng, 700, B, D, H, W, c)
    pro = oracle.bev_pool_prologue(coords, B, D, H, W)
    og = rng.standard_normal((B, D, H, W, c)).astype(np.float32)
    xg = oracle.bev_pool_backward_sorted(og, pro["geom_sorted"], pro["interval_starts"], pro["interval_lengths"],
                                         coords.shape[0], B, D, H, W)
    g = pro["geom_sorted"]
    assert np.array_equal(xg, og[g[:, 3], g[:, 2], g[:, 0], g[:, 1]])


def test_cell_index_truncates_toward_zero():
    """vtransforms/base.py:149: `.long()` truncates, so values in (-1, 0) land in cell 0 and are KEPT."""
    origin = np.array([0.0, 0.0, 0.0], np.float32)
    dx = np.array([1.0, 1.0, 1.0], np.float32)
    nx = np.array([4, 4, 1], np.int64)
    geom = np.array([[-0.5, 0.5, 0.2], [-1.0, 0.5, 0.2], [3.999, 3.2, 0.9], [4.0, 0.0, 0.0], [1.5, -0.999, -0.3]],
                    np.float32)
    coords, kept = oracle.bev_cell_index(geom, 1, origin, dx, nx)
    assert coords[0, 0] == 0 and kept[0]
    assert coords[1, 0] == -1 and not kept[1]
    assert kept[2] and not kept[3]
    assert coords[4, 1] == 0 and coords[4, 2] == 0 and kept[4]


def test_flagship_rig_statistics():
    """The synthetic rig reproduces SURVEY.md §8d: 1 815 552 of 1 993 728 frustum points kept,
    45 469 non-empty cells, interval length median 32 / max 864."""
    inp = synth.bev_pool_inputs(with_feats=False)
    assert inp["geom"].shape[0] == 1993728
    coords, kept = oracle.bev_cell_index(inp["geom"], 1, inp["origin"], inp["dx"], inp["nx"])
    assert int(kept.sum()) == 1815552
    pro = oracle.bev_pool_prologue(coords[kept], 1, 1, 360, 360)
    L = pro["interval_lengths"]
    assert len(L) == 45469 and int(np.median(L)) == 32 and int(L.max()) == 864


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, "bev_pool_ref_small.npz")),
                    reason="golden fixture produced by the reference kernel on the GPU box not committed yet")
def test_oracle_against_reference_kernel_golden():
    """Pin: outputs of the REFERENCE's own bev_pool kernel (hipified from /root/reference into
    oracle/_ref, run on an MI355X by tests/golden/make_bev_pool_golden.py)."""
    z = np.load(os.path.join(GOLDEN, "bev_pool_ref_small.npz"))
    out = oracle.bev_pool_forward_sorted(z["x"], z["geom"], z["interval_starts"], z["interval_lengths"],
                                         int(z["B"]), int(z["D"]), int(z["H"]), int(z["W"]), dtype=np.float32)
    # same row order, same sequential fp32 adds -> bit-exact
    assert np.array_equal(out, z["out"])
    xg = oracle.bev_pool_backward_sorted(z["out_grad"], z["geom"], z["interval_starts"], z["interval_lengths"],
                                         z["x"].shape[0], int(z["B"]), int(z["D"]), int(z["H"]), int(z["W"]))
    assert np.array_equal(xg, z["x_grad"])
