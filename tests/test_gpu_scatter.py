"""GPU parity: HIP dynamic scatter (through the C ABI) vs golden vectors from the reference's own GPU kernels and vs the
numpy oracle on random inputs.

Bars: voxel coordinates, point->voxel map and counts bit-exact; max (forward and backward) bit-exact; sum / mean bit-exact
where the data make every order of summation exact (the golden fixture), within 1e-5 relative of the float64 oracle
otherwise — the reference's own atomicAdd order is unspecified."""
import os

import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import voxel

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "scatter_ref.npz"))
CASES = sorted({k.split(".")[0] for k in Z.files})
MODES = ["sum", "mean", "max"]


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", CASES)
def test_golden_forward_backward(dev, name, mode):
    feats, coors = Z[f"{name}.feats"], Z[f"{name}.coors"]
    red, oc, cmap, cnt = voxel.dynamic_point_to_voxel_forward(_t(feats, dev), _t(coors, dev), mode)
    assert oc.dtype == torch.int32 and cmap.dtype == torch.int32 and cnt.dtype == torch.int32
    assert np.array_equal(oc.cpu().numpy(), Z[f"{name}.{mode}.out_coors"])
    assert np.array_equal(cmap.cpu().numpy(), Z[f"{name}.{mode}.coors_map"])
    assert np.array_equal(cnt.cpu().numpy(), Z[f"{name}.{mode}.count"])
    assert np.array_equal(red.cpu().numpy(), Z[f"{name}.{mode}.reduced"])
    gf = torch.full((feats.shape[0], feats.shape[1]), 7.0, device=dev)        # the call overwrites every element
    voxel.dynamic_point_to_voxel_backward(gf, _t(Z[f"{name}.{mode}.grad_reduced"], dev), _t(feats, dev), red, cmap, cnt, mode)
    assert np.array_equal(gf.cpu().numpy(), Z[f"{name}.{mode}.grad_feats"])


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("n,ndim,hi,c", [(1, 3, (5, 5, 5), 4), (70000, 3, (1440, 1440, 41), 5), (300000, 3, (360, 360, 10), 5),
                                         (20000, 2, (100000, 70000), 3), (5000, 4, (2, 30000, 30000, 40), 2),
                                         (4000, 1, (50,), 9), (3000, 3, (2 ** 30, 2 ** 30, 2 ** 30), 2)])
def test_random_vs_oracle(dev, mode, n, ndim, hi, c):
    rng = np.random.default_rng(n + ndim)
    coors = np.stack([rng.integers(0, h, n) for h in hi], 1).astype(np.int32)
    coors[n // 3: n // 3 + n // 10] = coors[:n // 10]                          # repeated rows far apart
    neg = rng.random(n) < 0.03
    coors[neg, rng.integers(0, ndim, int(neg.sum()))] = -rng.integers(1, 5)
    feats = rng.standard_normal((n, c)).astype(np.float32)
    ref, roc, rmap, rcnt = oracle.dynamic_scatter(feats, coors, mode)
    red, oc, cmap, cnt = voxel.dynamic_point_to_voxel_forward(_t(feats, dev), _t(coors, dev), mode)
    assert np.array_equal(oc.cpu().numpy(), roc) and np.array_equal(cmap.cpu().numpy(), rmap)
    assert np.array_equal(cnt.cpu().numpy(), rcnt)
    got = red.cpu().numpy()
    if mode == "max":
        assert np.array_equal(got, ref.astype(np.float32))
    else:
        assert np.max(np.abs(got - ref)) <= 1e-5 * (1 + np.abs(ref).max())
    # twice the same bits (fixed summation order)
    red2 = voxel.dynamic_point_to_voxel_forward(_t(feats, dev), _t(coors, dev), mode)[0]
    assert torch.equal(red, red2)
    g = rng.standard_normal(got.shape).astype(np.float32)
    gf = torch.empty(n, c, device=dev)
    voxel.dynamic_point_to_voxel_backward(gf, _t(g, dev), _t(feats, dev), red, cmap, cnt, mode)
    rg = oracle.dynamic_scatter_backward(g, feats, got, rmap, rcnt, mode)
    assert np.array_equal(gf.cpu().numpy(), rg)


def test_empty_and_int64_inputs(dev):
    feats = torch.zeros((0, 4), device=dev)
    coors = torch.zeros((0, 3), dtype=torch.int32, device=dev)
    red, oc, cmap, cnt = voxel.dynamic_point_to_voxel_forward(feats, coors, "max")
    assert red.shape == (0, 4) and oc.shape == (0, 3) and cmap.shape == (0,) and cnt.shape == (0,)
    c64 = torch.tensor([[2, 1, 0], [0, 5, 5], [2, 1, 0], [-1, 0, 0]], dtype=torch.int64, device=dev)
    f = torch.tensor([[1.0], [2.0], [5.0], [9.0]], device=dev)
    red, oc, cmap, cnt = voxel.dynamic_point_to_voxel_forward(f, c64, "sum")
    assert oc.dtype == torch.int64 and oc.cpu().tolist() == [[0, 5, 5], [2, 1, 0]]
    assert red.cpu().tolist() == [[2.0], [6.0]] and cmap.cpu().tolist() == [1, 0, 1, -1] and cnt.cpu().tolist() == [1, 2]
    with pytest.raises(RuntimeError):
        voxel.dynamic_point_to_voxel_forward(f, c64, "median")
    with pytest.raises(RuntimeError):
        voxel.dynamic_point_to_voxel_forward(f.cpu(), c64.cpu(), "max")


@pytest.mark.parametrize("average", [True, False])
def test_dynamic_scatter_module_and_autograd(dev, average):
    """DynamicScatter on batched coordinates [N, 4] = one scatter per sample, concatenated (scatter_points.py:86-99)."""
    rng = np.random.default_rng(5)
    n, c = 6000, 4
    coors = np.stack([np.sort(rng.integers(0, 3, n)), rng.integers(0, 12, n), rng.integers(0, 9, n), rng.integers(0, 4, n)],
                     1).astype(np.int32)
    feats = rng.standard_normal((n, c)).astype(np.float32)
    layer = voxel.DynamicScatter([0.2, 0.2, 4], [0, -40, -3, 70.4, 40, 1], average)
    x = _t(feats, dev).requires_grad_(True)
    vf, vc = layer(x, _t(coors, dev))
    mode = "mean" if average else "max"
    refs, refc, grads = [], [], np.zeros_like(feats)
    g = rng.standard_normal((vf.shape[0], c)).astype(np.float32)
    row = 0
    for b in range(3):
        sel = np.nonzero(coors[:, 0] == b)[0]
        red, oc, cmap, cnt = oracle.dynamic_scatter(feats[sel], coors[sel][:, 1:], mode)
        refs.append(red)
        refc.append(np.concatenate([np.full((oc.shape[0], 1), b, oc.dtype), oc], 1))
        grads[sel] = oracle.dynamic_scatter_backward(g[row:row + red.shape[0]], feats[sel], red.astype(np.float32), cmap, cnt,
                                                     mode)
        row += red.shape[0]
    assert np.array_equal(vc.cpu().numpy(), np.concatenate(refc))
    assert np.max(np.abs(vf.detach().cpu().numpy() - np.concatenate(refs))) <= 1e-5
    vf.backward(_t(g, dev))
    assert np.max(np.abs(x.grad.cpu().numpy() - grads)) <= 1e-6
    assert "DynamicScatter(voxel_size=" in repr(layer)
