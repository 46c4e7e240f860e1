"""GPU parity for the rest of the `sparse_conv_ext` surface: transposed / dilated / 2D rulebooks, the inverse
convolution, max pooling (forward + backward) and the modules built on them — against golden vectors produced by the
reference's own CPU functors (tests/golden/make_spconv_ext_golden.py) and against the CPU oracle on random cases.

Bars: output indices (CUDA row order), pair counts and pair SETS bit-exact; max pooling forward and backward bit-exact
(fp32 and fp16); convolution sums within REL_TOL of the reference's fp32 result."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import spconv
from bevfusion_amd.spconv import functional as Fsp
from bevfusion_amd.spconv import ops as sops

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
PATHS = sorted(glob.glob(os.path.join(GOLDEN, "spconv_ext_*.npz")))
REL_TOL = 2e-5
ext = sops.sparse_conv_ext


def _t(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


def _random_indices(rng, B, shape, n):
    idx = []
    for b in range(B):
        lin = rng.choice(int(np.prod(shape)), size=min(n, int(np.prod(shape))), replace=False)
        idx.append(np.concatenate([np.full((len(lin), 1), b), np.stack(np.unravel_index(lin, shape), 1)], 1))
    ind = np.concatenate(idx).astype(np.int32)
    rng.shuffle(ind, axis=0)
    return ind


def _check_rulebook(indices, B, shape, ks, st, pd, dl, op, subm, transpose, dev):
    oi, opairs, onum, oshape = oracle.get_indice_pairs(indices, B, shape, ks, st, pd, dl, subm, order="cuda",
                                                       transpose=transpose, out_padding=op)
    gi, gpairs, gnum = spconv.get_indice_pairs(_t(indices, dev), B, list(shape), list(ks), list(st), list(pd), list(dl),
                                               list(op), bool(subm), bool(transpose))
    assert np.array_equal(gi.cpu().numpy(), oi)
    assert np.array_equal(gnum.cpu().numpy(), onum)
    assert tuple(gpairs.shape) == (int(np.prod(ks)), 2, indices.shape[0])
    a = oracle.pairs_as_sets(opairs, onum)
    b = oracle.pairs_as_sets(gpairs.cpu().numpy(), gnum.cpu().numpy())
    for k in range(len(a)):
        assert np.array_equal(a[k], b[k]), f"offset {k}"
    return oi, opairs, onum, oshape


def test_fixtures_present():
    assert len(PATHS) >= 10


@pytest.mark.parametrize("path", PATHS)
def test_rulebook_on_golden_cases(dev, path):
    z = np.load(path)
    oi, _, _, _ = _check_rulebook(z["indices"], int(z["batch_size"]), tuple(z["spatial_shape"]), tuple(z["ksize"]),
                                  tuple(z["stride"]), tuple(z["padding"]), tuple(z["dilation"]), tuple(z["out_padding"]),
                                  int(z["subm"]), bool(z["transpose"]), dev)
    # same active set as the reference's CPU functor (which numbers rows by first appearance instead)
    assert sorted(map(tuple, oi)) == sorted(map(tuple, z["out_indices"]))


@pytest.mark.parametrize("path", PATHS)
def test_pybind_entry_points_on_reference_pairs(dev, path):
    """The reference's own pair arrays (CPU row numbering) through the drop-in functions: rows come out in the
    reference's numbering, so results compare element for element."""
    z = np.load(path)
    pairs, num = _t(z["indice_pairs"], dev), _t(z["indice_num"], dev)
    M, N = z["out_indices"].shape[0], z["indices"].shape[0]
    subm = int(z["subm"])
    out = ext.indice_conv_fp32(_t(z["features"], dev), _t(z["filters"], dev), pairs, num, M, 0, subm)
    assert np.max(np.abs(out.cpu().numpy() - z["out"])) <= REL_TOL * (1 + np.abs(z["out"]).max())
    inv = ext.indice_conv_fp32(_t(z["features_out"], dev), _t(z["filters"], dev), pairs, num, N, 1, 0)
    assert np.max(np.abs(inv.cpu().numpy() - z["inverse_out"])) <= REL_TOL * (1 + np.abs(z["inverse_out"]).max())
    pooled = ext.indice_maxpool_fp32(_t(z["pool_features"], dev), pairs, num, M)
    assert np.array_equal(pooled.cpu().numpy(), z["pooled"])
    gin = ext.indice_maxpool_backward_fp32(_t(z["pool_features"], dev), pooled, _t(z["pool_out_grad"], dev), pairs, num)
    assert np.array_equal(gin.cpu().numpy(), z["pool_in_grad"])
    # half entry points: the pooling inputs are multiples of 0.5, exact in fp16 -> the maximum is bit-identical
    pooled_h = ext.indice_maxpool_half(_t(z["pool_features"], dev), pairs, num, M)
    assert pooled_h.dtype == torch.float16 and np.array_equal(pooled_h.float().cpu().numpy(), z["pooled"])


@pytest.mark.parametrize("B,shape,n,ks,st,pd,dl,op,subm,transpose", [
    (2, (20, 18, 7), 900, (3, 3, 3), (2, 2, 2), (1, 1, 1), (1, 1, 1), (1, 1, 1), 0, 1),
    (2, (20, 18, 7), 900, (2, 2, 2), (2, 2, 2), (0, 0, 0), (1, 1, 1), (0, 0, 0), 0, 1),
    (1, (9, 30, 4), 500, (3, 2, 1), (1, 3, 2), (0, 1, 0), (1, 1, 1), (0, 2, 1), 0, 1),
    (3, (16, 16, 8), 1200, (3, 3, 3), (1, 1, 1), (2, 2, 2), (2, 2, 2), (0, 0, 0), 0, 0),
    (3, (16, 16, 8), 1200, (3, 3, 3), (1, 1, 1), (1, 1, 1), (2, 2, 2), (0, 0, 0), 1, 0),
    (2, (16, 16, 8), 1200, (3, 3, 3), (1, 1, 1), (3, 3, 3), (3, 2, 1), (0, 0, 0), 0, 0),
    (2, (17, 13, 6), 700, (2, 2, 2), (1, 1, 1), (1, 1, 1), (1, 1, 1), (0, 0, 0), 1, 0),
    (2, (64, 48), 1500, (3, 3), (1, 1), (1, 1), (1, 1), (0, 0), 1, 0),
    (2, (64, 48), 1500, (3, 3), (2, 2), (1, 1), (1, 1), (0, 0), 0, 0),
    (2, (31, 25), 600, (3, 3), (2, 2), (1, 1), (1, 1), (1, 1), 0, 1),
    (4, (200, 176), 12000, (3, 3), (2, 2), (1, 1), (1, 1), (0, 0), 0, 0),
])
def test_rulebook_random(dev, B, shape, n, ks, st, pd, dl, op, subm, transpose):
    rng = np.random.default_rng(B * n + subm + 2 * transpose)
    _check_rulebook(_random_indices(rng, B, shape, n), B, shape, ks, st, pd, dl, op, subm, transpose, dev)


def test_empty_and_single_inputs(dev):
    for nd, shape in ((3, (5, 5, 5)), (2, (6, 6))):
        empty = torch.zeros((0, nd + 1), dtype=torch.int32, device=dev)
        oi, pairs, num = spconv.get_indice_pairs(empty, 1, list(shape), 3, 2, 1, 1, 0, False, True)
        assert oi.shape[0] == 0 and int(num.sum()) == 0
        one = torch.zeros((1, nd + 1), dtype=torch.int32, device=dev)
        oi, pairs, num = spconv.get_indice_pairs(one, 1, list(shape), 2, 2, 0, 1, 0, False, True)
        assert oi.shape == (2 ** nd, nd + 1) and int(num.sum()) == 2 ** nd   # one input fans out to a 2^nd block


def test_4d_raises(dev):
    ind = torch.zeros((3, 5), dtype=torch.int32, device=dev)
    with pytest.raises(NotImplementedError):
        spconv.get_indice_pairs(ind, 1, [4, 4, 4, 4], 3, 1, 1)
    with pytest.raises(NotImplementedError):
        ext.get_indice_pairs_4d(ind, 1, [4] * 4, [4] * 4, [3] * 4, [1] * 4, [1] * 4, [1] * 4, [0] * 4, 0, 0)
    conv = spconv.SubMConv4d(4, 4, 3).to(dev)       # constructible, like the reference's class
    x = spconv.SparseConvTensor(torch.zeros(3, 4, device=dev), ind, [4, 4, 4, 4], 1)
    with pytest.raises(NotImplementedError):
        conv(x)


# ------------------------------------------------------------------------------------------------------------
# max pooling
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("C", [1, 5, 64, 130])
def test_maxpool_forward_backward_vs_oracle(dev, dtype, C):
    rng = np.random.default_rng(C)
    B, shape = 2, (24, 20, 9)
    indices = _random_indices(rng, B, shape, 2500)
    oi, opairs, onum, _ = oracle.get_indice_pairs(indices, B, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), [1, 1, 1], 0)
    f = np.round(rng.standard_normal((indices.shape[0], C)) * 4).astype(np.float32) / 4     # ties, exact in 16 bits
    g = np.round(rng.standard_normal((oi.shape[0], C)) * 2).astype(np.float32) / 2          # sums stay exact in bf16 too
    ref = oracle.indice_maxpool(f, opairs, onum, oi.shape[0])
    ref_g = oracle.indice_maxpool_backward(f, ref, g, opairs, onum)
    rb = spconv.build_rulebook(_t(indices, dev), B, list(shape), 3, 2, 1)
    x = _t(f, dev, dtype).requires_grad_(True)
    y = Fsp.rulebook_maxpool(x, rb)
    assert y.dtype == dtype and np.array_equal(y.detach().float().cpu().numpy(), ref)
    y.backward(_t(g, dev, dtype))
    assert np.array_equal(x.grad.float().cpu().numpy(), ref_g)
    # reference-shaped entry (pair lists) gives the same bits
    gp, gn = rb.indice_pairs()
    y2 = Fsp.indice_maxpool(_t(f, dev, dtype), gp, gn, rb.num_out)
    assert torch.equal(y2, y.detach())


def test_maxpool_floors_at_zero_and_ignores_nan(dev):
    ind = torch.tensor([[0, 0, 0, 0], [0, 1, 1, 1], [0, 5, 5, 5]], dtype=torch.int32, device=dev)
    f = torch.tensor([[-3.0, float("nan")], [-1.0, 2.0], [-7.0, -0.5]], device=dev)
    pool = spconv.SparseMaxPool3d(2, 2)
    out = pool(spconv.SparseConvTensor(f, ind, [6, 6, 6], 1))
    assert out.spatial_shape == [3, 3, 3] and out.indices.cpu().tolist() == [[0, 0, 0, 0], [0, 2, 2, 2]]
    assert out.features.cpu().tolist() == [[0.0, 2.0], [0.0, 0.0]]


@pytest.mark.parametrize("nd", [2, 3])
def test_maxpool_module_equals_dense_maxpool(dev, nd):
    """On non-negative features sparse max pooling is dense max pooling of the scattered tensor (absent cells are 0)."""
    rng = np.random.default_rng(nd)
    B, C = 2, 6
    shape = (14, 11, 6)[:nd]
    indices = _random_indices(rng, B, shape, 300)
    f = rng.random((indices.shape[0], C)).astype(np.float32)
    x = spconv.SparseConvTensor(_t(f, dev), _t(indices, dev), list(shape), B)
    pool = (spconv.SparseMaxPool2d if nd == 2 else spconv.SparseMaxPool3d)(3, 2, 1)
    y = pool(x)
    dense = x.dense()
    ref = (torch.nn.functional.max_pool2d if nd == 2 else torch.nn.functional.max_pool3d)(dense, 3, 2, 1)
    assert list(y.spatial_shape) == list(ref.shape[2:])
    # cells no input touches carry no row (dense() leaves them 0, like the dense maximum over an all-zero window)
    assert torch.equal(y.dense(), ref)


# ------------------------------------------------------------------------------------------------------------
# modules: transposed, inverse, 2D, dilated SubM
# ------------------------------------------------------------------------------------------------------------
def _dense_weight(w):
    """W[k..., ci, co] -> conv weight [co, ci, k...]."""
    nd = w.dim() - 2
    return w.permute(nd + 1, nd, *range(nd)).contiguous()


@pytest.mark.parametrize("nd", [2, 3])
def test_transposed_conv_module_equals_dense_conv_transpose(dev, nd):
    rng = np.random.default_rng(10 + nd)
    B, cin, cout = 2, 5, 7
    shape = (7, 6, 5)[:nd]
    indices = _random_indices(rng, B, shape, 60)
    f = rng.standard_normal((indices.shape[0], cin)).astype(np.float32)
    cls = spconv.SparseConvTranspose2d if nd == 2 else spconv.SparseConvTranspose3d
    conv = cls(cin, cout, 3, stride=2, padding=1, bias=True).to(dev)
    x = spconv.SparseConvTensor(_t(f, dev), _t(indices, dev), list(shape), B)
    y = conv(x)
    w = conv.weight.detach()
    # out = in*stride - padding + k is conv_transpose's own indexing (no kernel flip)
    wt = w.permute(nd, nd + 1, *range(nd)).contiguous()                       # [ci, co, k...]
    fn = torch.nn.functional.conv_transpose2d if nd == 2 else torch.nn.functional.conv_transpose3d
    ref = fn(x.dense(), wt, stride=2, padding=1)
    assert list(y.spatial_shape) == list(ref.shape[2:])
    yd = y.dense()
    active = (yd != 0).any(1, keepdim=True)
    ref_b = ref + conv.bias.detach().view(1, -1, *([1] * nd))
    assert torch.allclose(yd, ref_b * active, atol=1e-4)
    # outputs nobody reaches carry no row; every reached cell does
    touched = fn((x.dense() != 0).any(1, keepdim=True).float(), torch.ones(1, 1, *([3] * nd), device=dev), stride=2,
                 padding=1) > 0
    assert int(touched.sum()) == y.indices.shape[0]


def test_inverse_conv_restores_the_input_set_and_matches_oracle(dev):
    rng = np.random.default_rng(21)
    B, shape, c0, c1 = 2, (12, 10, 8), 6, 9
    indices = _random_indices(rng, B, shape, 400)
    f = rng.standard_normal((indices.shape[0], c0)).astype(np.float32)
    down = spconv.SparseConv3d(c0, c1, 3, stride=2, padding=1, bias=False, indice_key="down1").to(dev)
    up = spconv.SparseInverseConv3d(c1, c0, 3, indice_key="down1", bias=False).to(dev)
    x = spconv.SparseConvTensor(_t(f, dev).requires_grad_(True), _t(indices, dev), list(shape), B)
    mid = down(x)
    y = up(mid)
    assert list(y.spatial_shape) == list(shape) and torch.equal(y.indices, x.indices)
    oi, opairs, onum, _ = oracle.get_indice_pairs(indices, B, shape, (3, 3, 3), (2, 2, 2), (1, 1, 1), [1, 1, 1], 0)
    ref_mid = oracle.indice_conv(f, down.weight.detach().cpu().numpy(), opairs, onum, oi.shape[0])
    ref = oracle.indice_conv(ref_mid.astype(np.float32), up.weight.detach().cpu().numpy(), opairs, onum, indices.shape[0],
                             inverse=True)
    assert np.max(np.abs(y.features.detach().cpu().numpy() - ref)) <= 1e-4 * (1 + np.abs(ref).max())
    # gradients flow through both convolutions; compare with the oracle's backward of the inverse conv
    g = rng.standard_normal(ref.shape).astype(np.float32)
    y.features.backward(_t(g, dev))
    gi_mid, gw_up = oracle.indice_conv_backward(ref_mid.astype(np.float32), up.weight.detach().cpu().numpy(), g, opairs,
                                                onum, inverse=True)
    assert np.max(np.abs(up.weight.grad.cpu().numpy() - gw_up)) <= 2e-4 * (1 + np.abs(gw_up).max())
    gi, gw_down = oracle.indice_conv_backward(f, down.weight.detach().cpu().numpy(), gi_mid.astype(np.float32), opairs, onum)
    assert np.max(np.abs(x.features.grad.cpu().numpy() - gi)) <= 2e-4 * (1 + np.abs(gi).max())
    assert np.max(np.abs(down.weight.grad.cpu().numpy() - gw_down)) <= 2e-4 * (1 + np.abs(gw_down).max())


def test_conv2d_modules_equal_dense_conv2d(dev):
    rng = np.random.default_rng(31)
    B, shape, cin, cout = 3, (20, 17), 8, 12
    indices = _random_indices(rng, B, shape, 150)
    f = rng.standard_normal((indices.shape[0], cin)).astype(np.float32)
    x = spconv.SparseConvTensor(_t(f, dev), _t(indices, dev), list(shape), B)
    for cls, kw in ((spconv.SubMConv2d, dict(padding=1)), (spconv.SparseConv2d, dict(stride=2, padding=1))):
        conv = cls(cin, cout, 3, bias=False, **kw).to(dev)
        y = conv(x)
        ref = torch.nn.functional.conv2d(x.dense(), _dense_weight(conv.weight.detach()), stride=kw.get("stride", 1),
                                         padding=1)
        got = y.dense()
        rows = y.indices.long()
        assert torch.allclose(got[rows[:, 0], :, rows[:, 1], rows[:, 2]], ref[rows[:, 0], :, rows[:, 1], rows[:, 2]],
                              atol=1e-4)


def test_dilated_subm_reproduces_the_reference_shortcut(dev):
    """Golden case subm3_dil2: the reference runs the fullest offset as an identity GEMM although no offset of a dilated
    SubM rulebook is the identity; module, native op and pybind entry all reproduce its numbers."""
    z = np.load(os.path.join(GOLDEN, "spconv_ext_subm3_dil2.npz"))
    B, shape = int(z["batch_size"]), [int(v) for v in z["spatial_shape"]]
    conv = spconv.SubMConv3d(z["features"].shape[1], z["filters"].shape[-1], 3, dilation=2, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(_t(z["filters"], dev))
    y = conv(spconv.SparseConvTensor(_t(z["features"], dev), _t(z["indices"], dev), shape, B))
    assert np.max(np.abs(y.features.detach().cpu().numpy() - z["out"])) <= REL_TOL * (1 + np.abs(z["out"]).max())
    plain = oracle.indice_conv(z["features"], z["filters"], z["indice_pairs"], z["indice_num"], z["indices"].shape[0])
    assert np.max(np.abs(plain - z["out"])) > 1e-2       # the shortcut is not a no-op here
