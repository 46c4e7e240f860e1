"""CPU: the oracle's transposed / dilated / 2D rulebooks, inverse convolution and max pooling against golden vectors
produced by the reference's own CPU functors (tests/golden/make_spconv_ext_golden.py)."""
import glob
import os

import numpy as np
import pytest

import oracle

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
PATHS = sorted(glob.glob(os.path.join(GOLDEN, "spconv_ext_*.npz")))


def _rulebook(z, order):
    return oracle.get_indice_pairs(z["indices"], int(z["batch_size"]), z["spatial_shape"], z["ksize"], z["stride"],
                                   z["padding"], z["dilation"], int(z["subm"]), order=order,
                                   transpose=bool(z["transpose"]), out_padding=z["out_padding"])


def test_fixtures_present():
    assert len(PATHS) >= 10


@pytest.mark.parametrize("path", PATHS)
def test_rulebook_matches_reference_cpu_exactly(path):
    z = np.load(path)
    oi, pairs, num, oshape = _rulebook(z, "cpu")
    assert [int(v) for v in oshape] == [int(v) for v in z["out_shape"]]
    assert np.array_equal(oi, z["out_indices"])
    assert np.array_equal(num, z["indice_num"])
    assert np.array_equal(pairs, z["indice_pairs"])


@pytest.mark.parametrize("path", PATHS)
def test_conv_and_inverse_conv_match_reference_cpu(path):
    z = np.load(path)
    M, N = z["out_indices"].shape[0], z["indices"].shape[0]
    out = oracle.indice_conv(z["features"], z["filters"], z["indice_pairs"], z["indice_num"], M, subm=bool(z["subm"]))
    assert np.max(np.abs(out - z["out"])) < 2e-5
    inv = oracle.indice_conv(z["features_out"], z["filters"], z["indice_pairs"], z["indice_num"], N, inverse=True)
    assert np.max(np.abs(inv - z["inverse_out"])) < 2e-5


@pytest.mark.parametrize("path", PATHS)
def test_maxpool_matches_reference_cpu_bit_exactly(path):
    z = np.load(path)
    M = z["out_indices"].shape[0]
    pooled = oracle.indice_maxpool(z["pool_features"], z["indice_pairs"], z["indice_num"], M)
    assert np.array_equal(pooled, z["pooled"])
    assert pooled.min() >= 0.0                                   # the zero-initialised output floors the maximum
    gi = oracle.indice_maxpool_backward(z["pool_features"], pooled, z["pool_out_grad"], z["indice_pairs"],
                                        z["indice_num"])
    assert np.array_equal(gi, z["pool_in_grad"])                 # same offset order -> same fp32 sums


@pytest.mark.parametrize("path", PATHS)
def test_transposed_rulebook_is_the_mirror_of_a_regular_one(path):
    """A transposed conv's pairs, read backwards, are the pairs of the regular conv from its outputs to its inputs."""
    z = np.load(path)
    if not bool(z["transpose"]):
        pytest.skip("regular rulebook")
    oi, pairs, num, oshape = _rulebook(z, "cuda")
    nd = len(oshape)
    # regular conv over the transposed conv's OUTPUT set with the same geometry reaches a superset of the inputs
    oi2, pairs2, num2, shape2 = oracle.get_indice_pairs(oi, int(z["batch_size"]), oshape, z["ksize"], z["stride"],
                                                        z["padding"], z["dilation"], 0, order="cuda")
    assert all(int(shape2[i]) >= int(z["spatial_shape"][i]) for i in range(nd))
    key2 = {tuple(r): i for i, r in enumerate(oi2)}
    for k in range(pairs.shape[0]):
        for t in range(int(num[k])):
            i, o = int(pairs[k, 0, t]), int(pairs[k, 1, t])
            r = key2[tuple(z["indices"][i])]
            m = int(num2[k])
            assert np.any((pairs2[k, 0, :m] == o) & (pairs2[k, 1, :m] == r))
