"""GPU parity: the staged-rows filter gradient (bevamd_spconv_conv_wgrad_slab, csrc/spconv_wgrad_slab.h) against the float64
oracle restating indiceConvBackward (spconv_ops.h:363-456) and against the gather kernel on the same rulebook.

Bars: 16-bit features, fp32 accumulation, one rounding of the result: <= 2e-3 * (1 + max|ref|) of the float64 oracle on the rounded
inputs (the bar of test_gpu_spconv.py::test_filter_gradient_mfma_any_width_deterministic), bit-identical run to run; rows that are
NOT in linear order must not take this path at all (the range test of Rulebook.slab_meta_wgrad is measured, not promised)."""
import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import spconv
from bevfusion_amd.spconv import ops as sops
from conftest import record_parity

pytestmark = pytest.mark.gpu


def _sorted_indices(rng, B, shape, n):
    """n distinct cells per sample, rows in ascending (b, x, y, z) = ascending linear index"""
    idx = []
    for b in range(B):
        lin = np.sort(rng.choice(int(np.prod(shape)), size=min(n, int(np.prod(shape))), replace=False))
        idx.append(np.concatenate([np.full((len(lin), 1), b), np.stack(np.unravel_index(lin, shape), 1)], 1))
    return np.concatenate(idx).astype(np.int32)


def _case(rng, indices, B, shape, c, dtype, dev):
    oi, opairs, onum, _ = oracle.get_indice_pairs(indices, B, shape, (3, 3, 3), (1, 1, 1), (1, 1, 1), [1, 1, 1], 1, order="cuda")
    n = indices.shape[0]
    x = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32)).to(dev).to(dtype)
    g = torch.from_numpy(rng.standard_normal((n, c)).astype(np.float32)).to(dev).to(dtype)
    w = np.zeros((3, 3, 3, c, c), np.float32)
    _, ref = oracle.indice_conv_backward(x.float().cpu().numpy(), w, g.float().cpu().numpy(), opairs, onum)
    rb = spconv.build_rulebook(torch.from_numpy(indices).to(dev), B, list(shape), [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, True)
    return x, g, rb, ref.reshape(27, c, c)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("c", [16, 32, 64, 128])
def test_slab_filter_gradient_vs_oracle_and_gather_kernel(dev, c, dtype):
    rng = np.random.default_rng(c + (7 if dtype == torch.bfloat16 else 0))
    B, shape = 2, (40, 24, 9)
    indices = _sorted_indices(rng, B, shape, 2500)           # 5000 rows: 40 blocks, a partial last block, several slabs
    x, g, rb, ref = _case(rng, indices, B, shape, c, dtype, dev)
    meta = rb.slab_meta_wgrad(c)
    assert meta is not None, "rows in linear order must qualify for the staged-rows path"
    outs = [sops.sparse_conv_wgrad_slab(x, g, meta, c, c) for _ in range(3)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])          # no atomics, fixed-order partials
    got = outs[0].float().cpu().numpy()
    scale = 1 + np.abs(ref).max()
    bar = 2e-3 if dtype == torch.float16 else 1.6e-2        # one rounding of the result type (bf16: 8 bits of mantissa)
    err = float(np.max(np.abs(got - ref))) / scale
    record_parity(f"wgrad_slab_{c}_{str(dtype).split('.')[-1]}", err, bar)
    assert err <= bar, err
    # the gather kernel on the same rulebook: the same sums in another order, rounded once each
    wt = torch.zeros((3, 3, 3, c, c), dtype=dtype, device=dev)
    lib_gather = sops.sparse_conv_backward(x, wt, g, *rb.conv_tables(), x.shape[0])[1].float().cpu().numpy().reshape(27, c, c)
    assert float(np.max(np.abs(got - lib_gather))) / scale <= 2 * bar


def test_ranges_longer_than_the_stage_are_walked_in_pieces(dev):
    """A tall, dense (y, z) plane makes the input range of a block through a kernel plane longer than the 192 / 256 staged rows:
    the kernel walks it in pieces (slots outside the piece read the zero row)."""
    rng = np.random.default_rng(5)
    B, shape = 1, (5, 16, 96)
    indices = _sorted_indices(rng, B, shape, 6900)            # 90 % occupancy, 96 cells per (x, y) line: ranges of ~300 rows
    for c, dtype in ((32, torch.float16), (16, torch.float16), (64, torch.bfloat16)):
        x, g, rb, ref = _case(rng, indices, B, shape, c, dtype, dev)
        meta = rb.slab_meta_wgrad(c)
        assert meta is not None
        nblk = (indices.shape[0] + 127) // 128
        cnt = (meta.hdr[:nblk * 24].view(torch.int32).view(nblk, 3, 2)[:, :, 1] & 0x3FFFFFFF).cpu().numpy()
        assert cnt.max() > 256, "the case must exercise the piece loop"
        got = sops.sparse_conv_wgrad_slab(x, g, meta, c, c).float().cpu().numpy()
        bar = 2e-3 if dtype == torch.float16 else 1.6e-2
        assert float(np.max(np.abs(got - ref))) / (1 + np.abs(ref).max()) <= bar


def test_unordered_rows_keep_the_gather_kernel(dev):
    rng = np.random.default_rng(11)
    B, shape = 2, (40, 24, 9)
    indices = _sorted_indices(rng, B, shape, 2500)
    rng.shuffle(indices, axis=0)
    x, g, rb, ref = _case(rng, indices, B, shape, 32, torch.float16, dev)
    assert rb.slab_meta_wgrad(32) is None
    wt = torch.zeros((3, 3, 3, 32, 32), dtype=torch.float16, device=dev)
    got = sops.sparse_conv_backward(x, wt, g, *rb.conv_tables(), x.shape[0], rulebook=rb)[1].float().cpu().numpy().reshape(27, 32, 32)
    assert float(np.max(np.abs(got - ref))) / (1 + np.abs(ref).max()) <= 2e-3


def test_autograd_takes_the_staged_rows_path_on_ordered_rows(dev):
    """SubMConv3d under autocast on rows in linear order: the filter gradient autograd returns equals the direct call."""
    rng = np.random.default_rng(3)
    B, shape = 1, (40, 24, 9)
    indices = _sorted_indices(rng, B, shape, 3000)
    conv = spconv.SubMConv3d(32, 32, 3, padding=1, bias=False, indice_key="s").to(dev)
    feats = torch.randn(indices.shape[0], 32, device=dev)
    sp = spconv.SparseConvTensor(feats, torch.from_numpy(indices).to(dev), list(shape), B)
    with torch.autocast("cuda", dtype=torch.float16):
        out = conv(sp)
    gy = torch.randn_like(out.features)
    out.features.backward(gy)
    rb = out.indice_dict["s"].rulebook
    meta = rb.slab_meta_wgrad(32)
    assert meta is not None
    direct = sops.sparse_conv_wgrad_slab(feats.half(), gy.half(), meta, 32, 32).view(3, 3, 3, 32, 32)
    assert torch.equal(conv.weight.grad, direct.to(conv.weight.grad.dtype))
