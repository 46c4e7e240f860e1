"""GPU parity: HIP spconv (rulebook + fused MFMA convolution, through the C ABI) vs the CPU oracle and the
golden vectors produced by the reference's own CPU functors.

Bars: output indices (CUDA row order) and rulebook pair SETS bit-exact; convolution sums within REL_TOL of
the float64 oracle for fp32 (exact-fp32 MFMA chain), and within half-precision rounding for fp16/bf16."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import spconv, synth
from bevfusion_amd.spconv import ops as sops

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
PATHS = sorted(glob.glob(os.path.join(GOLDEN, "spconv_ref_*.npz")))
REL_TOL = 2e-5   # fp32: |err| <= REL_TOL * (1 + max|ref|)


def _random_indices(rng, B, shape, n):
    idx = []
    for b in range(B):
        lin = rng.choice(int(np.prod(shape)), size=min(n, int(np.prod(shape))), replace=False)
        idx.append(np.concatenate([np.full((len(lin), 1), b), np.stack(np.unravel_index(lin, shape), 1)], 1))
    ind = np.concatenate(idx).astype(np.int32)
    rng.shuffle(ind, axis=0)
    return ind


def _check_rulebook(indices, B, shape, ks, st, pd, subm, dev):
    oi, opairs, onum, oshape = oracle.get_indice_pairs(indices, B, shape, ks, st, pd, [1, 1, 1], subm, order="cuda")
    gi, gpairs, gnum = spconv.get_indice_pairs(torch.from_numpy(indices).to(dev), B, list(shape), list(ks), list(st),
                                               list(pd), 1, 0, bool(subm), False)
    assert np.array_equal(gi.cpu().numpy(), oi)                      # rows: ascending linear index / input order
    assert np.array_equal(gnum.cpu().numpy(), onum)
    assert tuple(gpairs.shape) == (int(np.prod(ks)), 2, indices.shape[0])
    a = oracle.pairs_as_sets(opairs, onum)
    b = oracle.pairs_as_sets(gpairs.cpu().numpy(), gnum.cpu().numpy())
    for k in range(len(a)):
        assert np.array_equal(a[k], b[k]), f"offset {k}"
    gp = gpairs.cpu().numpy()
    for k in range(gp.shape[0]):                                      # -1 padding behind the valid pairs
        assert np.all(gp[k, :, int(onum[k]):] == -1)
    return oi, opairs, onum


@pytest.mark.parametrize("path", PATHS)
def test_rulebook_on_golden_cases(dev, path):
    z = np.load(path)
    _check_rulebook(z["indices"], int(z["batch_size"]), tuple(z["spatial_shape"]), tuple(z["ksize"]), tuple(z["stride"]),
                    tuple(z["padding"]), int(z["subm"]), dev)


@pytest.mark.parametrize("B,shape,n,ks,st,pd,subm", [
    (1, (1, 1, 1), 1, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1),
    (2, (40, 36, 11), 3000, (3, 3, 3), (1, 1, 1), (1, 1, 1), 1),
    (2, (40, 36, 11), 3000, (3, 3, 3), (2, 2, 2), (1, 1, 1), 0),
    (3, (23, 17, 9), 1500, (3, 3, 3), (2, 2, 2), (1, 1, 0), 0),
    (2, (20, 20, 5), 900, (1, 1, 3), (1, 1, 2), (0, 0, 0), 0),
    (1, (16, 16, 16), 2000, (3, 3, 3), (1, 1, 1), (0, 0, 0), 0),     # stride 1, 27 candidates per input
    (2, (9, 9, 9), 700, (2, 2, 2), (2, 2, 2), (0, 0, 0), 0),
    (4, (64, 64, 8), 20000, (3, 3, 3), (2, 2, 2), (1, 1, 1), 0),
])
def test_rulebook_random(dev, B, shape, n, ks, st, pd, subm):
    rng = np.random.default_rng(B * n + subm)
    _check_rulebook(_random_indices(rng, B, shape, n), B, shape, ks, st, pd, subm, dev)


def _conv_case(rng, n_in, cin, cout, ks, dtype):
    w = (rng.standard_normal(tuple(ks) + (cin, cout)) / np.sqrt(cin * np.prod(ks) / 4)).astype(np.float32)
    f = rng.standard_normal((n_in, cin)).astype(np.float32)
    if dtype != torch.float32:  # oracle consumes the rounded values
        f = torch.from_numpy(f).to(dtype).float().numpy()
        w = torch.from_numpy(w).to(dtype).float().numpy()
    return f, w


@pytest.mark.parametrize("cin,cout", [(5, 16), (16, 16), (16, 32), (32, 64), (64, 64), (64, 128), (128, 128), (7, 9),
                                      (4, 48), (24, 100)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_conv_forward_vs_oracle(dev, cin, cout, dtype):
    rng = np.random.default_rng(cin * 131 + cout)
    B, shape = 2, (24, 20, 9)
    indices = _random_indices(rng, B, shape, 1200)
    for ks, st, pd, subm in [((3, 3, 3), (1, 1, 1), (1, 1, 1), 1), ((3, 3, 3), (2, 2, 2), (1, 1, 1), 0)]:
        oi, opairs, onum, _ = oracle.get_indice_pairs(indices, B, shape, ks, st, pd, [1, 1, 1], subm, order="cuda")
        f, w = _conv_case(rng, indices.shape[0], cin, cout, ks, dtype)
        ref = oracle.indice_conv(f, w, opairs, onum, oi.shape[0])
        rb = spconv.build_rulebook(torch.from_numpy(indices).to(dev), B, list(shape), list(ks), list(st), list(pd), 1, subm)
        out = spconv.sparse_conv(torch.from_numpy(f).to(dev).to(dtype), torch.from_numpy(w).to(dev).to(dtype), rb.nbr,
                                 rb.num_out)
        assert out.dtype == dtype and tuple(out.shape) == (oi.shape[0], cout)
        err = np.max(np.abs(out.float().cpu().numpy().astype(np.float64) - ref))
        scale = 1.0 + np.max(np.abs(ref))
        tol = REL_TOL if dtype == torch.float32 else (2e-3 if dtype == torch.float16 else 1.6e-2)
        assert err <= tol * scale, (err, scale)


def test_conv_is_bit_reproducible_and_transpose_detecting(dev):
    """Same inputs twice -> identical bits (fixed summation order); asymmetric weights/inputs (A=I style check)."""
    rng = np.random.default_rng(3)
    B, shape, cin, cout = 1, (12, 12, 6), 16, 32
    indices = _random_indices(rng, B, shape, 400)
    rb = spconv.build_rulebook(torch.from_numpy(indices).to(dev), B, list(shape), 3, 1, 1, 1, True)
    f = torch.zeros(indices.shape[0], cin, device=dev)
    f[torch.arange(indices.shape[0]), torch.arange(indices.shape[0]) % cin] = 1.0   # one-hot rows
    w = torch.arange(27 * cin * cout, device=dev, dtype=torch.float32).view(3, 3, 3, cin, cout) / 1000.0  # asymmetric
    o1 = spconv.sparse_conv(f, w, rb.nbr, rb.num_out)
    o2 = spconv.sparse_conv(f, w, rb.nbr, rb.num_out)
    assert torch.equal(o1, o2)
    _, opairs, onum, _ = oracle.get_indice_pairs(indices, B, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), [1, 1, 1], 1)
    ref = oracle.indice_conv(f.cpu().numpy(), w.cpu().numpy(), opairs, onum, indices.shape[0])
    assert np.max(np.abs(o1.cpu().numpy() - ref)) < 1e-4


def test_many_rows_two_tiles_per_wave(dev):
    """>= 65536 output rows selects the 32-rows-per-wave instantiation."""
    rng = np.random.default_rng(9)
    B, shape, cin, cout = 2, (96, 96, 12), 16, 16
    indices = _random_indices(rng, B, shape, 40000)
    oi, opairs, onum, _ = oracle.get_indice_pairs(indices, B, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), [1, 1, 1], 1)
    f, w = _conv_case(rng, indices.shape[0], cin, cout, (3, 3, 3), torch.float32)
    ref = oracle.indice_conv(f, w, opairs, onum, oi.shape[0])
    rb = spconv.build_rulebook(torch.from_numpy(indices).to(dev), B, list(shape), 3, 1, 1, 1, True)
    assert rb.num_out >= 65536
    out = spconv.sparse_conv(torch.from_numpy(f).to(dev), torch.from_numpy(w).to(dev), rb.nbr, rb.num_out)
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= REL_TOL * (1 + np.abs(ref).max())


def test_fused_epilogue(dev):
    rng = np.random.default_rng(4)
    B, shape, cin, cout = 1, (14, 14, 6), 32, 48
    indices = _random_indices(rng, B, shape, 500)
    rb = spconv.build_rulebook(torch.from_numpy(indices).to(dev), B, list(shape), 3, 1, 1, 1, True)
    f, w = _conv_case(rng, indices.shape[0], cin, cout, (3, 3, 3), torch.float32)
    f, w = torch.from_numpy(f).to(dev), torch.from_numpy(w).to(dev)
    bias, sc, sh = (torch.randn(cout, device=dev) for _ in range(3))
    res = torch.randn(rb.num_out, cout, device=dev)
    plain = spconv.sparse_conv(f, w, rb.nbr, rb.num_out)
    fused = spconv.sparse_conv(f, w, rb.nbr, rb.num_out, bias=bias, bn_scale=sc, bn_shift=sh, residual=res, relu=True)
    exp = torch.relu((plain + bias) * sc + sh + res)
    assert torch.allclose(fused, exp, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("path", PATHS)
def test_drop_in_ext_vs_reference_cpu_golden(dev, path):
    """sparse_conv_ext.{get_indice_pairs_3d, indice_conv_fp32, indice_conv_backward_fp32} on the REFERENCE's
    own rulebook arrays (CPU row order) reproduce the reference's CPU outputs."""
    z = np.load(path)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ext = spconv.sparse_conv_ext
    subm = int(z["subm"])
    out = ext.indice_conv_fp32(t(z["features"]), t(z["filters"]), t(z["indice_pairs"]), t(z["indice_num"]),
                               z["out_indices"].shape[0], 0, subm)
    assert np.max(np.abs(out.cpu().numpy() - z["out"])) <= 3e-5 * (1 + np.abs(z["out"]).max())
    gi, gw = ext.indice_conv_backward_fp32(t(z["features"]), t(z["filters"]), t(z["out_grad"]), t(z["indice_pairs"]),
                                           t(z["indice_num"]), 0, subm)
    assert np.max(np.abs(gi.cpu().numpy() - z["in_grad"])) <= 5e-5 * (1 + np.abs(z["in_grad"]).max())
    assert np.max(np.abs(gw.cpu().numpy() - z["filter_grad"])) <= 1e-4 * (1 + np.abs(z["filter_grad"]).max())
    oi, pairs, num = ext.get_indice_pairs_3d(t(z["indices"]), int(z["batch_size"]), list(z["out_shape"]),
                                             list(z["spatial_shape"]), list(z["ksize"]), list(z["stride"]),
                                             list(z["padding"]), [1, 1, 1], [0, 0, 0], subm, 0)
    assert np.array_equal(num.cpu().numpy(), z["indice_num"])
    assert {tuple(r) for r in oi.cpu().numpy()} == {tuple(r) for r in z["out_indices"]}  # same set; CUDA order differs (D8)
    half = ext.indice_conv_half(t(z["features"]).half(), t(z["filters"]).half(), t(z["indice_pairs"]),
                                t(z["indice_num"]), z["out_indices"].shape[0], 0, subm)
    assert half.dtype == torch.float16
    assert np.max(np.abs(half.float().cpu().numpy() - z["out"])) <= 2e-2 * (1 + np.abs(z["out"]).max())


@pytest.mark.parametrize("subm", [1, 0])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_autograd_backward_vs_oracle(dev, subm, dtype):
    rng = np.random.default_rng(11 + subm)
    B, shape, cin, cout = 2, (16, 14, 7), 16, 32
    indices = _random_indices(rng, B, shape, 600)
    ks, st, pd = ((3, 3, 3), (1, 1, 1), (1, 1, 1)) if subm else ((3, 3, 3), (2, 2, 2), (1, 1, 1))
    oi, opairs, onum, _ = oracle.get_indice_pairs(indices, B, shape, ks, st, pd, [1, 1, 1], subm, order="cuda")
    f, w = _conv_case(rng, indices.shape[0], cin, cout, ks, dtype)
    og = rng.standard_normal((oi.shape[0], cout)).astype(np.float32)
    if dtype != torch.float32:
        og = torch.from_numpy(og).to(dtype).float().numpy()
    gi_ref, gw_ref = oracle.indice_conv_backward(f, w, og, opairs, onum)
    conv = (spconv.SubMConv3d if subm else spconv.SparseConv3d)(cin, cout, 3, stride=st[0], padding=1, bias=False).to(dev)
    with torch.no_grad():
        conv.weight.copy_(torch.from_numpy(w))
    conv = conv.to(dtype)
    x = torch.from_numpy(f).to(dev).to(dtype).requires_grad_(True)
    sp = spconv.SparseConvTensor(x, torch.from_numpy(indices).to(dev), list(shape), B)
    out = conv(sp)
    assert np.array_equal(out.indices.cpu().numpy(), oi)
    out.features.backward(torch.from_numpy(og).to(dev).to(dtype))
    tol_i, tol_w = (5e-5, 2e-4) if dtype == torch.float32 else (5e-3, 2e-2)
    assert np.max(np.abs(x.grad.float().cpu().numpy() - gi_ref)) <= tol_i * (1 + np.abs(gi_ref).max())
    assert np.max(np.abs(conv.weight.grad.float().cpu().numpy() - gw_ref)) <= tol_w * (1 + np.abs(gw_ref).max())


@pytest.mark.parametrize("cin,cout", [(5, 16), (16, 32), (32, 32), (64, 128), (128, 128), (7, 100), (128, 9)])
def test_filter_gradient_mfma_any_width_deterministic(dev, cin, cout):
    """bevamd_spconv_conv_wgrad (MFMA, slab partials + fixed-order reduce): every channel count up to 128 (the old
    kernel refused cin*cout > 16384 only at backward time), fp32 within 2e-4 of the float64 oracle, bit-identical run to run
    (no atomics), several row slabs."""
    rng = np.random.default_rng(cin * 131 + cout)
    B, shape = 2, (24, 20, 9)
    indices = _random_indices(rng, B, shape, 2300)       # 4600 rows -> 3 row slabs
    oi, opairs, onum, _ = oracle.get_indice_pairs(indices, B, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), [1, 1, 1], 1, order="cuda")
    f, w = _conv_case(rng, indices.shape[0], cin, cout, (3, 3, 3), torch.float32)
    og = rng.standard_normal((oi.shape[0], cout)).astype(np.float32)
    _, gw_ref = oracle.indice_conv_backward(f, w, og, opairs, onum)
    rb = spconv.build_rulebook(torch.from_numpy(indices).to(dev), B, list(shape), [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, True)
    from bevfusion_amd.spconv import ops as sops

    x, wt, g = torch.from_numpy(f).to(dev), torch.from_numpy(w).to(dev), torch.from_numpy(og).to(dev)
    nbr, nbr_t = rb.conv_tables()
    outs = [sops.sparse_conv_backward(x, wt, g, nbr, nbr_t, x.shape[0])[1] for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    got = outs[0].cpu().numpy()
    assert got.shape == gw_ref.shape
    assert np.max(np.abs(got - gw_ref)) <= 2e-4 * (1 + np.abs(gw_ref).max())
    # 16-bit features: same kernel, products of the widened values (exact), one rounding of the result
    gh = sops.sparse_conv_backward(x.half(), wt.half(), g.half(), nbr, nbr_t, x.shape[0])[1]
    _, gw_h = oracle.indice_conv_backward(x.half().float().cpu().numpy(), wt.half().float().cpu().numpy(),
                                          g.half().float().cpu().numpy(), opairs, onum)
    assert gh.dtype == torch.float16
    assert np.max(np.abs(gh.float().cpu().numpy() - gw_h)) <= 2e-3 * (1 + np.abs(gw_h).max())


def test_empty_tensor(dev):
    conv = spconv.SubMConv3d(4, 8, 3, padding=1, bias=False).to(dev)
    sp = spconv.SparseConvTensor(torch.zeros(0, 4, device=dev), torch.zeros(0, 4, dtype=torch.int32, device=dev), [8, 8, 8], 1)
    out = conv(sp)
    assert tuple(out.features.shape) == (0, 8)
    conv2 = spconv.SparseConv3d(4, 8, 3, stride=2, padding=1, bias=True).to(dev)
    out = conv2(sp)
    assert tuple(out.features.shape) == (0, 8) and out.spatial_shape == [4, 4, 4]


def test_rulebook_cache_and_indice_key_semantics(dev):
    """indice_key shares rulebooks by name (conv.py:152-183); key=None convs are cached by geometry (D7)."""
    rng = np.random.default_rng(2)
    indices = _random_indices(rng, 1, (16, 16, 8), 500)
    a = spconv.SubMConv3d(4, 4, 3, padding=1, bias=False, indice_key="subm1").to(dev)
    b = spconv.SubMConv3d(4, 4, 3, padding=1, bias=False, indice_key="subm1").to(dev)
    c = spconv.SubMConv3d(4, 4, 3, padding=1, bias=False).to(dev)
    sp = spconv.SparseConvTensor(torch.randn(indices.shape[0], 4, device=dev), torch.from_numpy(indices).to(dev), [16, 16, 8], 1)
    o1 = a(sp)
    o2 = b(o1)
    o3 = c(o2)
    d = o3.indice_dict
    assert d["subm1"].rulebook is [v for k, v in d.items() if isinstance(k, tuple)][0].rulebook
    assert len([k for k in d if isinstance(k, tuple)]) == 1           # one SubM rulebook for all three convs
    outids, inds, pairs, num, shp = d["subm1"]                        # unpacks like the reference's 5-tuple
    assert pairs.shape == (27, 2, indices.shape[0]) and int(num[13]) == indices.shape[0]
    dense = o3.dense()
    assert tuple(dense.shape) == (1, 4, 16, 16, 8)
    got = dense[0, :, indices[:, 1], indices[:, 2], indices[:, 3]].t()
    assert torch.equal(got, o3.features)


@pytest.mark.parametrize("dtype,fn", [(torch.float32, "fused_indice_conv_fp32"), (torch.float16, "fused_indice_conv_half")])
def test_drop_in_fused_indice_conv_adds_bias(dev, dtype, fn):
    """sparse_conv_ext.fused_indice_conv_* (all.cc:32-37) == indice_conv + bias."""
    rng = np.random.default_rng(17)
    B, shape, cin, cout = 2, (16, 14, 7), 16, 32
    indices = _random_indices(rng, B, shape, 600)
    ti = torch.from_numpy(indices).to(dev)
    oi, pairs, num = spconv.get_indice_pairs(ti, B, list(shape), [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, 0, True, False)
    f = torch.from_numpy(rng.standard_normal((indices.shape[0], cin)).astype(np.float32)).to(dev).to(dtype)
    w = torch.from_numpy((rng.standard_normal((3, 3, 3, cin, cout)) * 0.1).astype(np.float32)).to(dev).to(dtype)
    b = torch.from_numpy(rng.standard_normal(cout).astype(np.float32)).to(dev).to(dtype)
    plain = sops.indice_conv(f, w, pairs, num, oi.shape[0], False, True)
    fused = getattr(sops.sparse_conv_ext, fn)(f, w, b, pairs, num, oi.shape[0], 0, 1)
    assert fused.dtype == dtype
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((fused.float() - (plain.float() + b.float())).abs().max()) <= tol
