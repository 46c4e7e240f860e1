"""CPU: the spconv oracle (oracle/spconv_oracle.c) against golden vectors from the reference's own CPU
functors (tests/golden/make_spconv_golden.py) and against dense conv3d on the scattered tensor — the
cross-check upstream spconv's test_utils documents."""
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
PATHS = sorted(glob.glob(os.path.join(GOLDEN, "spconv_ref_*.npz")))


@pytest.mark.parametrize("path", PATHS)
def test_rulebook_matches_reference_cpu_exactly(path):
    z = np.load(path)
    oi, pairs, num, oshape = oracle.get_indice_pairs(z["indices"], int(z["batch_size"]), z["spatial_shape"], z["ksize"],
                                                     z["stride"], z["padding"], [1, 1, 1], int(z["subm"]), order="cpu")
    assert list(oshape) == list(z["out_shape"])
    assert np.array_equal(oi, z["out_indices"])        # CPU row order (first appearance)
    assert np.array_equal(num, z["indice_num"])
    assert np.array_equal(pairs, z["indice_pairs"])    # same enumeration order as geometry.h


@pytest.mark.parametrize("path", PATHS)
def test_conv_forward_backward_match_reference_cpu(path):
    z = np.load(path)
    out = oracle.indice_conv(z["features"], z["filters"], z["indice_pairs"], z["indice_num"], z["out_indices"].shape[0])
    assert np.max(np.abs(out - z["out"])) < 2e-5
    gi, gw = oracle.indice_conv_backward(z["features"], z["filters"], z["out_grad"], z["indice_pairs"], z["indice_num"])
    assert np.max(np.abs(gi - z["in_grad"])) < 5e-5
    assert np.max(np.abs(gw - z["filter_grad"])) < 2e-4


@pytest.mark.parametrize("path", PATHS)
def test_cuda_order_is_a_consistent_renumbering(path):
    z = np.load(path)
    args = (z["indices"], int(z["batch_size"]), z["spatial_shape"], z["ksize"], z["stride"], z["padding"], [1, 1, 1],
            int(z["subm"]))
    oi_c, p_c, n_c, osz = oracle.get_indice_pairs(*args, order="cpu")
    oi_g, p_g, n_g, _ = oracle.get_indice_pairs(*args, order="cuda")
    assert np.array_equal(n_c, n_g)
    if not int(z["subm"]):
        lin = ((oi_g[:, 0].astype(np.int64) * osz[0] + oi_g[:, 1]) * osz[1] + oi_g[:, 2]) * osz[2] + oi_g[:, 3]
        assert np.all(np.diff(lin) > 0)               # ascending linear index, no duplicates
    # the convolution result is the same tensor, rows permuted
    oc = oracle.indice_conv(z["features"], z["filters"], p_c, n_c, oi_c.shape[0])
    og = oracle.indice_conv(z["features"], z["filters"], p_g, n_g, oi_g.shape[0])
    key = lambda a: [tuple(r) for r in a]
    d = {k: v for k, v in zip(key(oi_c), oc)}
    assert all(np.allclose(d[k], v, atol=1e-12) for k, v in zip(key(oi_g), og))


@pytest.mark.parametrize("subm,ks,st,pd", [(1, (3, 3, 3), (1, 1, 1), (1, 1, 1)), (0, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                           (0, (3, 3, 3), (2, 2, 2), (1, 1, 0)), (0, (1, 1, 3), (1, 1, 2), (0, 0, 0)),
                                           (0, (3, 3, 3), (1, 1, 1), (0, 0, 0))])
def test_sparse_conv_equals_dense_conv3d(subm, ks, st, pd):
    rng = np.random.default_rng(5)
    B, shape, cin, cout = 2, (11, 9, 8), 6, 7
    idx = []
    for b in range(B):
        lin = rng.choice(int(np.prod(shape)), size=120, replace=False)
        idx.append(np.concatenate([np.full((120, 1), b), np.stack(np.unravel_index(lin, shape), 1)], 1))
    indices = np.concatenate(idx).astype(np.int32)
    feats = rng.standard_normal((indices.shape[0], cin)).astype(np.float32)
    w = rng.standard_normal(ks + (cin, cout)).astype(np.float32)
    oi, pairs, num, oshape = oracle.get_indice_pairs(indices, B, shape, ks, st, pd, [1, 1, 1], subm, order="cuda")
    out = oracle.indice_conv(feats, w, pairs, num, oi.shape[0])
    dense = torch.zeros(B, cin, *shape, dtype=torch.float64)
    dense[indices[:, 0], :, indices[:, 1], indices[:, 2], indices[:, 3]] = torch.from_numpy(feats).double()
    wt = torch.from_numpy(w).double().permute(4, 3, 0, 1, 2)   # W_ref[kx,ky,kz,ci,co] == conv3d.weight[co,ci,kx,ky,kz]
    ref = F.conv3d(dense, wt, stride=st, padding=pd)
    got = ref[oi[:, 0], :, oi[:, 1], oi[:, 2], oi[:, 3]].numpy()
    assert np.max(np.abs(got - out)) < 1e-10
    if not subm:
        # regular sparse conv activates exactly the outputs some input touches
        occ = torch.zeros(B, 1, *shape, dtype=torch.float64)
        occ[indices[:, 0], 0, indices[:, 1], indices[:, 2], indices[:, 3]] = 1
        act = F.conv3d(occ, torch.ones(1, 1, *ks, dtype=torch.float64), stride=st, padding=pd) > 0
        assert int(act.sum()) == oi.shape[0]
    else:
        assert np.array_equal(oi, indices) and num[num.size // 2] == indices.shape[0]  # centre offset = identity
