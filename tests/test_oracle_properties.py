"""CPU: size-independent properties of the oracle (hypothesis) — the invariants the GPU parity tests rely on when they
compare whole tensors: hard voxelization, bev_pool (linearity, point-order invariance), dynamic
scatter (conservation), SubM / transposed rulebooks (symmetry)."""
import numpy as np
from hypothesis import given, settings, strategies as st

import oracle

SETTINGS = dict(max_examples=25, deadline=None)


@settings(**SETTINGS)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 400), st.integers(1, 6), st.integers(1, 40))
def test_hard_voxelization_invariants(seed, n, max_points, max_voxels):
    rng = np.random.default_rng(seed)
    vs, cr = [0.5, 0.25, 1.0], [0.0, -2.0, -1.0, 4.0, 2.0, 3.0]      # 8 x 16 x 4 grid
    pts = rng.uniform(-0.5, 4.5, (n, 4)).astype(np.float32)
    pts[:, 1] = rng.uniform(-2.5, 2.5, n)
    pts[:, 2] = rng.uniform(-1.5, 3.5, n)
    voxels, coors, npv = oracle.hard_voxelize(pts, vs, cr, max_points, max_voxels)
    m = coors.shape[0]
    assert m <= max_voxels and np.all(npv >= 1) and np.all(npv <= max_points)
    assert len({tuple(c) for c in coors}) == m                      # one row per voxel
    per_point = oracle.dynamic_voxelize(pts, vs, cr)
    inside = (per_point >= 0).all(1)
    # voxel order = order of first appearance among the in-range points; kept set = the first max_voxels of them
    first_seen = []
    seen = set()
    for c in per_point[inside]:
        t = tuple(c)
        if t not in seen:
            seen.add(t)
            first_seen.append(t)
    assert [tuple(c) for c in coors] == first_seen[:max_voxels]
    # every stored point lies in its voxel, in input order, unused slots are zero
    for v in range(m):
        mine = pts[inside][(per_point[inside] == coors[v]).all(1)][:max_points]
        assert npv[v] == mine.shape[0] and np.array_equal(voxels[v, :npv[v]], mine)
        assert not voxels[v, npv[v]:].any()


@settings(**SETTINGS)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 300))
def test_bev_pool_is_linear_and_order_invariant(seed, n):
    rng = np.random.default_rng(seed)
    B, D, H, W, C = 2, 1, 6, 5, 3
    coords = np.stack([rng.integers(0, H, n), rng.integers(0, W, n), rng.integers(0, D, n), rng.integers(0, B, n)], 1)
    f1, f2 = rng.standard_normal((n, C)).astype(np.float32), rng.standard_normal((n, C)).astype(np.float32)
    a = oracle.bev_pool(f1, coords, B, D, H, W)
    b = oracle.bev_pool(f2, coords, B, D, H, W)
    ab = oracle.bev_pool((f1.astype(np.float64) * 2 + f2).astype(np.float32), coords, B, D, H, W)
    assert np.allclose(ab, 2 * a + b, atol=1e-5)
    perm = rng.permutation(n)
    assert np.allclose(oracle.bev_pool(f1[perm], coords[perm], B, D, H, W), a, atol=1e-9)   # float64 sums
    assert np.isclose(a.sum(), f1.astype(np.float64).sum(), atol=1e-6)                     # nothing lost, nothing counted twice


@settings(**SETTINGS)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 300), st.integers(1, 4))
def test_dynamic_scatter_conserves_points(seed, n, ndim):
    rng = np.random.default_rng(seed)
    coors = rng.integers(-1, 5, (n, ndim)).astype(np.int32)
    feats = rng.standard_normal((n, 3)).astype(np.float32)
    s, oc, cmap, cnt = oracle.dynamic_scatter(feats, coors, "sum")
    mean = oracle.dynamic_scatter(feats, coors, "mean")[0]
    mx = oracle.dynamic_scatter(feats, coors, "max")[0]
    valid = (coors >= 0).all(1)
    assert cnt.sum() == valid.sum() and np.array_equal(cmap >= 0, valid)
    assert np.allclose(s.sum(0), feats[valid].astype(np.float64).sum(0), atol=1e-9)
    assert np.allclose(mean * cnt[:, None], s, atol=1e-9)
    if oc.shape[0]:
        assert np.all(mx >= mean - 1e-12) and [tuple(r) for r in oc] == sorted({tuple(r) for r in coors[valid]})
        g = rng.standard_normal(mx.shape).astype(np.float32)
        gm = oracle.dynamic_scatter_backward(g, feats, mx.astype(np.float32), cmap, cnt, "max")
        assert np.count_nonzero(gm) <= g.size and np.isclose(gm.sum(), g.sum(), rtol=1e-5, atol=1e-5)   # one taker per (voxel, ch)


@settings(**SETTINGS)
@given(st.integers(0, 2 ** 31 - 1), st.integers(1, 120))
def test_rulebook_symmetries(seed, n):
    rng = np.random.default_rng(seed)
    shape = (7, 6, 5)
    lin = rng.choice(int(np.prod(shape)), size=min(n, int(np.prod(shape))), replace=False)
    ind = np.concatenate([np.zeros((len(lin), 1), np.int64), np.stack(np.unravel_index(lin, shape), 1)], 1).astype(np.int32)
    # SubM 3x3x3: offset k pairs (i -> o) mirror offset 26-k pairs (o -> i); the centre is the identity
    _, pairs, num, _ = oracle.get_indice_pairs(ind, 1, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), (1, 1, 1), 1)
    sets = oracle.pairs_as_sets(pairs, num)
    assert num[13] == ind.shape[0] and np.array_equal(sets[13][:, 0], sets[13][:, 1])
    for k in range(13):
        mirror = sets[26 - k][:, ::-1]
        assert np.array_equal(sets[k][np.lexsort((sets[k][:, 0], sets[k][:, 1]))], mirror[np.lexsort((mirror[:, 0], mirror[:, 1]))])
    # transposed k2 s2: every input fans out to its own 2x2x2 block — 8 pairs each, all outputs distinct
    oi, pairs_t, num_t, oshape = oracle.get_indice_pairs(ind, 1, shape, (2, 2, 2), (2, 2, 2), (0, 0, 0), (1, 1, 1), 0,
                                                         transpose=True)
    assert list(oshape) == [14, 12, 10] and oi.shape[0] == 8 * ind.shape[0] and np.all(num_t == ind.shape[0])
