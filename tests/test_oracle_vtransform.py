"""Pins the view-transform oracle (oracle/vtransform_oracle.c) and the module's host-side mirror to
tests/golden/vtransform_ref.npz — outputs of the REFERENCE's own Python function bodies
(/root/reference/mmdet3d/models/vtransforms/base.py:92-135, 149-169, 283-329) exec'd on CPU torch by
tests/golden/make_vtransform_golden.py.  Everything here is bit-exact: geometry floats, cell indices, range mask, hit-pixel
sets, per-pixel depths and collision winners; the flagship-size case is compared through SHA-256 digests."""
import hashlib
import os

import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import synth
from bevfusion_amd.vtransforms import DepthLSSTransform

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vtransform_ref.npz")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def case(gold, prefix):
    return {k[len(prefix) + 1:]: gold[k] for k in gold.files if k.startswith(prefix + "_")}


def points_of(c):
    pts = [synth.lidar_points(seed=int(s), sweeps=int(c["points_sweeps"])) for s in c["points_seed"]]
    for p, h in zip(pts, c["points_sha256"]):
        assert sha(p) == str(h), "synthetic LiDAR generator drifted from the fixture"
    return pts


def oracle_geometry(c):
    B = c["la"].shape[0]
    return oracle.lss_geometry(c["frustum"], None, c["ia"][..., :3, 3], None, c["c2l"][..., :3, 3], None,
                               extra_rots=c["la"][:, :3, :3], extra_trans=c["la"][:, :3, 3],
                               inv_post_rots=c["inv_post_rots"], combine=c["combine"]), B


def dense_depth(c):
    d = np.zeros(int(np.prod(c["depth_shape"])), np.float32)
    d[c["depth_lin"]] = c["depth_val"]
    return d.reshape(c["depth_shape"])


def test_oracle_geometry_and_cells_bit_exact_small(gold):
    c = case(gold, "small")
    geom, B = oracle_geometry(c)
    assert np.array_equal(geom.view(np.uint32), c["geom"].view(np.uint32))                 # float bits
    coords, kept = oracle.bev_cell_index(geom.reshape(-1, 3), B, c["origin"], c["dx"], c["nx"])
    assert np.array_equal(coords[:, :3], c["cells"][:, :3].astype(np.int64))
    assert np.array_equal(coords[:, 3], c["cells"][:, 3].astype(np.int64))
    assert np.array_equal(kept, c["kept"])


def test_oracle_geometry_and_cells_bit_exact_flagship(gold):
    c = case(gold, "flag")
    geom, B = oracle_geometry(c)
    assert geom.shape == (1, 6, 118, 32, 88, 3)
    assert sha(geom) == str(c["geom_sha256"])
    coords, kept = oracle.bev_cell_index(geom.reshape(-1, 3), B, c["origin"], c["dx"], c["nx"])
    assert sha(coords.astype(np.int32)) == str(c["cells_sha256"])
    assert sha(kept.astype(np.uint8)) == str(c["kept_sha256"]) and int(kept.sum()) == int(c["n_kept"])


@pytest.mark.parametrize("prefix", ["small", "flag"])
def test_oracle_depth_raster_bit_exact(gold, prefix):
    c = case(gold, prefix)
    pts = points_of(c)
    ref = dense_depth(c)
    for b, p in enumerate(pts):
        d, winner = oracle.depth_raster(p, c["l2i"][b], c["ia"][b], c["la"][b], ref.shape[-2:],
                                        inv_lidar_aug_rot=c["inv_lidar_aug_rot"][b])
        assert np.array_equal(d.view(np.uint32), ref[b].view(np.uint32))     # pixel set, depths and collision winners
        assert int((winner >= 0).sum()) == int((ref[b] != 0).sum()) > 1000


def test_module_host_formulation_is_the_reference(gold):
    """The module's torch formulation (CPU tensors / autograd path) is the reference's, call for call: same bits."""
    c = case(gold, "small")
    cfg = dict(synth.CL_CONFIG, feature_size=(8, 22), dbound=(1.0, 60.0, 2.0))
    vt = DepthLSSTransform(256, 80, cfg["image_size"], cfg["feature_size"], cfg["xbound"], cfg["ybound"], cfg["zbound"],
                           cfg["dbound"], downsample=2)
    assert np.array_equal(vt.frustum.numpy(), c["frustum"]) and np.array_equal(vt.dx.numpy(), c["dx"])
    assert np.array_equal(vt.bx.numpy(), c["bx"]) and np.array_equal(vt.nx.numpy(), c["nx"])
    t = torch.from_numpy
    geom = vt.get_geometry(t(c["c2l"][..., :3, :3]), t(c["c2l"][..., :3, 3]), t(c["K"][..., :3, :3]), t(c["ia"][..., :3, :3]),
                           t(c["ia"][..., :3, 3]), extra_rots=t(c["la"][:, :3, :3]), extra_trans=t(c["la"][:, :3, 3]))
    assert np.array_equal(geom.numpy().view(np.uint32), c["geom"].view(np.uint32))
    pts = points_of(c)
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)           # sequential index_put = last point wins, like the fixture
    try:
        d = vt.depth_raster(torch.zeros(len(pts), 6, 1, 1, 1), [t(p) for p in pts], t(c["l2i"]), t(c["ia"]), t(c["la"]))
    finally:
        torch.set_num_threads(nthreads)
    assert np.array_equal(d.numpy().view(np.uint32), dense_depth(c).view(np.uint32))
