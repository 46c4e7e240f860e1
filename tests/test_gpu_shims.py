"""The pybind11 shims of integration/ LINKED against libbevfusion_amd.so and EXECUTED on the GPU (they were
syntax-checked only), under the reference's own module names and positional signatures:

  bev_pool_ext.bev_pool_forward / bev_pool_backward                (ops/bev_pool/src/bev_pool_cpu.cpp:89-94)
  voxel_layer.hard_voxelize / dynamic_voxelize                     (ops/voxel/src/voxelization.cpp:6-11)
  sparse_conv_ext.get_indice_pairs_3d / indice_conv_{fp32,half} / fused_indice_conv_fp32 / indice_conv_backward_fp32
                                                                   (ops/spconv/src/all.cc:21-51)
on the committed golden vectors of the reference's own kernels / CPU functors, and then the REFERENCE'S OWN PYTHON WRAPPERS
(ops/bev_pool/bev_pool.py, ops/voxel/voxelize.py, ops/spconv/{ops,functional,structure}.py — staged, git-ignored, by
integration/build_shims.py; they import nothing but torch and their extension module) driven over those drop-in modules:
`bev_pool(feats, coords, B, D, H, W)`, `Voxelization(...)`, `get_indice_pairs` + `SparseConvFunction` / `SubMConvFunction`
(forward and backward) must reproduce the goldens / the oracle.

The modules are built in the CPU container (`__graft_entry__.build()` -> `integration.build_shims.build_all()`) and travel with
the snapshot; on a box where they are missing the tests build them (g++, ~1 min) or skip if torch's headers are not there."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from integration import build_shims  # noqa: E402


@pytest.fixture(scope="module")
def shims():
    mods = {}
    for name in build_shims.SHIMS:
        if not os.path.exists(build_shims.so_path(name)):
            try:
                build_shims.build_one(name)
            except Exception as e:  # no compiler / headers on this box
                pytest.skip(f"shim {name} is not built and cannot be built here: {e}")
        mods[name] = build_shims.load_shim(name)
    return mods


@pytest.fixture(scope="module")
def refpy(shims):
    """The reference's Python wrappers as packages ref_bev_pool / ref_voxel / ref_spconv whose extension module is the shim."""
    root = build_shims.stage_reference_python()
    if root is None or not os.path.exists(os.path.join(root, "ref_bev_pool", "bev_pool.py")):
        pytest.skip("the reference's Python wrappers were not staged (run integration.build_shims where /root/reference exists)")
    if root not in sys.path:
        sys.path.insert(0, root)
    for pkg, ext in (("ref_bev_pool", "bev_pool_ext"), ("ref_voxel", "voxel_layer"), ("ref_spconv", "sparse_conv_ext")):
        importlib.import_module(pkg)
        sys.modules[f"{pkg}.{ext}"] = shims[ext]            # what `from . import bev_pool_ext` / `from .voxel_layer import ...` find
        setattr(sys.modules[pkg], ext, shims[ext])
    return dict(bev_pool=importlib.import_module("ref_bev_pool.bev_pool"), voxelize=importlib.import_module("ref_voxel.voxelize"),
                ops=importlib.import_module("ref_spconv.ops"), functional=importlib.import_module("ref_spconv.functional"),
                structure=importlib.import_module("ref_spconv.structure"))


def T(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rulebook_vs_golden(z, out_indices, pairs, num):
    """The golden rulebook comes from the reference's CPU functors, whose strided output rows are in hash-iteration order; the
    drop-in numbers them like the reference's CUDA path (ascending linear index).  The oracle (pinned to the goldens by
    tests/test_oracle_spconv.py, both orders) gives the CUDA-order rulebook; `perm` maps golden rows to drop-in rows."""
    L = lambda k: tuple(int(v) for v in z[k])
    oi, opairs, onum, _ = oracle.get_indice_pairs(z["indices"], int(z["batch_size"]), L("spatial_shape"), L("ksize"), L("stride"),
                                                  L("padding"), [1, 1, 1], int(z["subm"]), order="cuda")
    m = oi.shape[0]
    assert m == z["out_indices"].shape[0]
    assert np.array_equal(out_indices[:m].cpu().numpy(), oi) and np.array_equal(num.cpu().numpy(), onum)
    assert np.array_equal(onum, z["indice_num"])
    a, b = oracle.pairs_as_sets(opairs, onum), oracle.pairs_as_sets(pairs.cpu().numpy(), num.cpu().numpy())
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    shape = L("out_shape")
    key = lambda c: ((c[:, 0].astype(np.int64) * shape[0] + c[:, 1]) * shape[1] + c[:, 2]) * shape[2] + c[:, 3]
    if int(z["subm"]):
        gold_sorted = np.arange(m)                                       # submanifold: output rows ARE the input rows, in input order
    else:
        gold_sorted = np.argsort(key(z["out_indices"]), kind="stable")  # golden row of the j-th smallest cell = drop-in row j
    assert np.array_equal(z["out_indices"][gold_sorted], oi)
    return m, gold_sorted


# ---- the extension modules themselves --------------------------------------------------------------------------------------
def test_bev_pool_ext_on_the_reference_kernel_golden(dev, shims):
    ext = shims["bev_pool_ext"]
    z = np.load(os.path.join(GOLDEN, "bev_pool_ref_small.npz"))
    B, D, H, W = (int(z[k]) for k in "BDHW")
    out = ext.bev_pool_forward(T(z["x"], dev), T(z["geom"], dev), T(z["interval_lengths"], dev), T(z["interval_starts"], dev), B, D, H, W)
    assert tuple(out.shape) == z["out"].shape and np.max(np.abs(out.cpu().numpy() - z["out"])) <= 1e-4
    xg = ext.bev_pool_backward(T(z["out_grad"], dev), T(z["geom"], dev), T(z["interval_lengths"], dev), T(z["interval_starts"], dev),
                               B, D, H, W)
    assert np.array_equal(xg.cpu().numpy(), z["x_grad"])
    with pytest.raises(RuntimeError):                                 # TORCH_CHECK -> RuntimeError, like the reference's CHECK_CUDA
        ext.bev_pool_forward(torch.from_numpy(z["x"]), T(z["geom"], dev), T(z["interval_lengths"], dev), T(z["interval_starts"], dev),
                             B, D, H, W)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_voxel_layer_on_the_reference_cpu_golden(dev, shims, case):
    ext = shims["voxel_layer"]
    z = np.load(os.path.join(GOLDEN, f"voxel_ref_{case}.npz"))
    pts = T(z["points"], dev)
    mp, mv = int(z["max_points"]), int(z["max_voxels"])
    voxels = pts.new_zeros((mv, mp, pts.shape[1]))
    coors = pts.new_zeros((mv, 3), dtype=torch.int)
    npv = pts.new_zeros((mv,), dtype=torch.int)
    n = ext.hard_voxelize(pts, voxels, coors, npv, z["voxel_size"].tolist(), z["coors_range"].tolist(), mp, mv, 3, True)
    assert n == z["coors"].shape[0]
    assert np.array_equal(coors[:n].cpu().numpy(), z["coors"]) and np.array_equal(npv[:n].cpu().numpy(), z["num_points_per_voxel"])
    assert np.array_equal(voxels[:n].cpu().numpy(), z["voxels"])
    dyn = pts.new_zeros((pts.shape[0], 3), dtype=torch.int)
    ext.dynamic_voxelize(pts, dyn, z["voxel_size"].tolist(), z["coors_range"].tolist(), 3)
    assert np.array_equal(dyn.cpu().numpy(), z["dynamic_coors"])


SPCONV_CASES = ["spconv_ref_subm3", "spconv_ref_conv_s2", "spconv_ref_conv_p110", "spconv_ref_conv_out", "spconv_ref_subm_dense"]


@pytest.mark.parametrize("case", SPCONV_CASES)
def test_sparse_conv_ext_on_the_reference_cpu_functor_goldens(dev, shims, case):
    ext = shims["sparse_conv_ext"]
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    L = lambda k: [int(v) for v in z[k]]
    subm = int(z["subm"])
    res = ext.get_indice_pairs_3d(T(z["indices"], dev), int(z["batch_size"]), L("out_shape"), L("spatial_shape"), L("ksize"),
                                  L("stride"), L("padding"), [1, 1, 1], [0, 0, 0], subm, 0)
    out_indices, pairs, num = res[0], res[1], res[2]
    m, gold_sorted = _rulebook_vs_golden(z, out_indices, pairs, num)
    feats, filt = T(z["features"], dev), T(z["filters"], dev)
    out = ext.indice_conv_fp32(feats, filt, pairs, num, m, 0, subm)
    ref = z["out"][gold_sorted]
    assert np.max(np.abs(out.cpu().numpy() - ref)) <= 2e-5 * (1 + np.abs(ref).max())
    outh = ext.indice_conv_half(feats.half(), filt.half(), pairs, num, m, 0, subm)
    assert outh.dtype == torch.float16 and np.max(np.abs(outh.float().cpu().numpy() - ref)) <= 4e-3 * (1 + np.abs(ref).max())
    bias = torch.linspace(-0.5, 0.5, filt.shape[-1], device=dev)
    outb = ext.fused_indice_conv_fp32(feats, filt, bias, pairs, num, m, 0, subm)
    assert np.max(np.abs(outb.cpu().numpy() - (ref + bias.cpu().numpy()))) <= 2e-5 * (1 + np.abs(ref).max())
    # backward against the goldens' OWN in_grad / filter_grad: the reference's indice_conv_backward_fp32 on the fixture's out_grad
    # (spconv_ops.h:363-456; the golden's output rows permuted into the drop-in's row order — input rows and filter are order-free)
    go = z["out_grad"][gold_sorted]
    gi, gw = ext.indice_conv_backward_fp32(feats, filt, T(go, dev), pairs, num, 0, subm)
    assert np.max(np.abs(gi.cpu().numpy() - z["in_grad"])) <= 2e-5 * (1 + np.abs(z["in_grad"]).max())
    assert np.max(np.abs(gw.cpu().numpy() - z["filter_grad"])) <= 1e-4 * (1 + np.abs(z["filter_grad"]).max())
    # ... and on a second, random gradient against the float64 restatement
    rng = np.random.default_rng(3)
    go = rng.standard_normal(ref.shape).astype(np.float32)
    gi, gw = ext.indice_conv_backward_fp32(feats, filt, T(go, dev), pairs, num, 0, subm)
    rgi, rgw = oracle.indice_conv_backward(z["features"], z["filters"], go, pairs.cpu().numpy(), num.cpu().numpy())
    assert np.max(np.abs(gi.cpu().numpy() - rgi)) <= 2e-5 * (1 + np.abs(rgi).max())
    assert np.max(np.abs(gw.cpu().numpy() - rgw)) <= 1e-4 * (1 + np.abs(rgw).max())


# ---- the reference's own Python over the drop-in modules -----------------------------------------------------------------------
def test_reference_bev_pool_py_over_the_shim(dev, refpy):
    """ops/bev_pool/bev_pool.py:83-97 `bev_pool(feats, coords, B, D, H, W)` + QuickCumsumCuda.backward, unmodified."""
    rng = np.random.default_rng(5)
    B, D, H, W, C, n = 2, 2, 12, 10, 80, 5000
    coords = np.stack([rng.integers(0, H, n), rng.integers(0, W, n), rng.integers(0, D, n), rng.integers(0, B, n)], 1).astype(np.int64)
    feats = (rng.standard_normal((n, C)) * 0.25).astype(np.float32)
    x = T(feats, dev).requires_grad_(True)
    out = refpy["bev_pool"].bev_pool(x, T(coords, dev), B, D, H, W)          # [B, C, D, H, W]
    ref = oracle.bev_pool(feats, coords, B, D, H, W)
    assert tuple(out.shape) == ref.shape and float(np.max(np.abs(out.detach().cpu().numpy() - ref))) <= 1e-4
    g = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(T(g, dev))
    want = g[coords[:, 3], :, coords[:, 2], coords[:, 0], coords[:, 1]]      # every point receives its cell's gradient
    assert np.array_equal(x.grad.cpu().numpy(), want)


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_reference_voxelize_py_over_the_shim(dev, refpy, case):
    """ops/voxel/voxelize.py:10-71 `Voxelization(...)` (module + autograd Function), unmodified, on the reference-CPU goldens."""
    z = np.load(os.path.join(GOLDEN, f"voxel_ref_{case}.npz"))
    mv = int(z["max_voxels"])
    layer = refpy["voxelize"].Voxelization(z["voxel_size"].tolist(), z["coors_range"].tolist(), int(z["max_points"]), max_voxels=(mv, mv))
    voxels, coors, npv = layer(T(z["points"], dev))
    assert np.array_equal(coors.cpu().numpy(), z["coors"]) and np.array_equal(npv.cpu().numpy(), z["num_points_per_voxel"])
    assert np.array_equal(voxels.cpu().numpy(), z["voxels"])
    dyn = refpy["voxelize"].Voxelization(z["voxel_size"].tolist(), z["coors_range"].tolist(), -1)(T(z["points"], dev))
    assert np.array_equal(dyn.cpu().numpy(), z["dynamic_coors"])


@pytest.mark.parametrize("case", ["spconv_ref_subm3", "spconv_ref_conv_s2", "spconv_ref_conv_out"])
def test_reference_spconv_py_over_the_shim(dev, refpy, case):
    """ops/spconv/ops.py:85-211 `get_indice_pairs` / `indice_conv`, functional.py:22-123 SparseConvFunction / SubMConvFunction
    (forward + backward through autograd) and structure.py `SparseConvTensor.dense()`, unmodified."""
    ops, fn, st = refpy["ops"], refpy["functional"], refpy["structure"]
    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    L = lambda k: [int(v) for v in z[k]]
    subm = bool(int(z["subm"]))
    outids, pairs, num = ops.get_indice_pairs(T(z["indices"], dev), int(z["batch_size"]), L("spatial_shape"), L("ksize"), L("stride"),
                                              L("padding"), [1, 1, 1], 0, subm, False)[:3]
    m, gold_sorted = _rulebook_vs_golden(z, outids, pairs, num)
    feats = T(z["features"], dev).requires_grad_(True)
    filt = T(z["filters"], dev).requires_grad_(True)
    func = fn.SubMConvFunction if subm else fn.SparseConvFunction
    out = func.apply(feats, filt, pairs, num, m)
    ref = z["out"][gold_sorted]
    assert np.max(np.abs(out.detach().cpu().numpy() - ref)) <= 2e-5 * (1 + np.abs(ref).max())
    rng = np.random.default_rng(9)
    go = rng.standard_normal(ref.shape).astype(np.float32)
    out.backward(T(go, dev))
    rgi, rgw = oracle.indice_conv_backward(z["features"], z["filters"], go, pairs.cpu().numpy(), num.cpu().numpy())
    assert np.max(np.abs(feats.grad.cpu().numpy() - rgi)) <= 2e-5 * (1 + np.abs(rgi).max())
    assert np.max(np.abs(filt.grad.cpu().numpy() - rgw)) <= 1e-4 * (1 + np.abs(rgw).max())
    # SparseConvTensor.dense() of the reference on the GPU result
    t = st.SparseConvTensor(out.detach(), outids[:m], L("out_shape"), int(z["batch_size"]))
    dense = t.dense().cpu().numpy()                                       # [B, C, X, Y, Z]
    oi = z["out_indices"][gold_sorted]
    assert np.max(np.abs(dense[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]] - ref)) <= 2e-5 * (1 + np.abs(ref).max())
    assert np.count_nonzero(np.abs(dense).sum(1)) <= m
