"""CPU: the dynamic-scatter oracle against golden vectors produced by the reference's own GPU kernels
(tests/golden/make_scatter_golden.py; feature values are multiples of 1/8, so the reference's atomic sums are exact and
sum / mean are pinned bit for bit as well as max)."""
import os

import numpy as np
import pytest

import oracle

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "scatter_ref.npz"))
CASES = sorted({k.split(".")[0] for k in Z.files})


def test_fixture_present():
    assert CASES == ["allbad", "batch4", "dense3", "small3", "wide3"]


@pytest.mark.parametrize("mode", ["sum", "mean", "max"])
@pytest.mark.parametrize("name", CASES)
def test_forward_matches_reference_gpu(name, mode):
    feats, coors = Z[f"{name}.feats"], Z[f"{name}.coors"]
    red, oc, cmap, cnt = oracle.dynamic_scatter(feats, coors, mode, dtype=np.float32)
    assert np.array_equal(oc, Z[f"{name}.{mode}.out_coors"])          # ascending lexicographic order
    assert np.array_equal(cmap, Z[f"{name}.{mode}.coors_map"])
    assert np.array_equal(cnt, Z[f"{name}.{mode}.count"])
    assert np.array_equal(red, Z[f"{name}.{mode}.reduced"])
    if oc.shape[0] > 1:
        keys = [tuple(r) for r in oc]
        assert keys == sorted(keys) and len(set(keys)) == len(keys)


@pytest.mark.parametrize("mode", ["sum", "mean", "max"])
@pytest.mark.parametrize("name", CASES)
def test_backward_matches_reference_gpu(name, mode):
    feats = Z[f"{name}.feats"]
    g = oracle.dynamic_scatter_backward(Z[f"{name}.{mode}.grad_reduced"], feats, Z[f"{name}.{mode}.reduced"],
                                        Z[f"{name}.{mode}.coors_map"], Z[f"{name}.{mode}.count"], mode)
    assert np.array_equal(g, Z[f"{name}.{mode}.grad_feats"])
    if mode == "max" and feats.shape[0] and Z[f"{name}.{mode}.reduced"].shape[0]:
        # exactly one point per (voxel, channel) takes the gradient
        cmap = Z[f"{name}.{mode}.coors_map"]
        assert np.count_nonzero(g[cmap < 0]) == 0
