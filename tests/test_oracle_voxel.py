"""CPU: the voxelization oracle (oracle/voxel_oracle.c) against golden vectors produced by the
reference's own hard_voxelize_cpu (tests/golden/make_voxel_golden.py) and against an independent
pure-python statement of the semantics."""
import glob
import os

import numpy as np
import pytest

import oracle
from bevfusion_amd import synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "voxel_ref_*.npz"))))
def test_oracle_matches_reference_cpu_golden(path):
    z = np.load(path)
    v, c, n = oracle.hard_voxelize(z["points"], z["voxel_size"], z["coors_range"], int(z["max_points"]),
                                   int(z["max_voxels"]))
    assert np.array_equal(c, z["coors"])                # voxel order + coords: bit-exact
    assert np.array_equal(n, z["num_points_per_voxel"])
    assert np.array_equal(v, z["voxels"])               # copies are exact
    assert np.array_equal(oracle.dynamic_voxelize(z["points"], z["voxel_size"], z["coors_range"]), z["dynamic_coors"])


def _python_hard_voxelize(points, vs, cr, max_points, max_voxels):
    """Independent restatement: dict keyed by coordinate tuple, first-appearance numbering."""
    vs = np.asarray(vs, np.float32)
    cr = np.asarray(cr, np.float32)
    grid = np.round((cr[3:] - cr[:3]) / vs).astype(np.int64)
    table, voxels, coors, counts = {}, [], [], []
    for p in points:
        c = np.floor((p[:3] - cr[:3]) / vs)
        if np.any(c < 0) or np.any(c >= grid):
            continue
        key = tuple(int(t) for t in c)
        if key not in table:
            if len(coors) >= max_voxels:
                continue
            table[key] = len(coors)
            coors.append(key)
            voxels.append(np.zeros((max_points, points.shape[1]), np.float32))
            counts.append(0)
        i = table[key]
        if counts[i] < max_points:
            voxels[i][counts[i]] = p
            counts[i] += 1
    return np.array(voxels), np.array(coors, np.int32), np.array(counts, np.int32)


@pytest.mark.parametrize("max_points,max_voxels", [(10, 100000), (2, 300), (1, 50)])
def test_oracle_matches_python_on_non_cubic_grid(max_points, max_voxels):
    rng = np.random.default_rng(7)
    pts = rng.uniform(-1.2, 1.2, size=(4000, 5)).astype(np.float32)
    vs, cr = [0.1, 0.05, 0.4], [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]  # 20 x 40 x 5 grid
    v, c, n = oracle.hard_voxelize(pts, vs, cr, max_points, max_voxels)
    pv, pc, pn = _python_hard_voxelize(pts, vs, cr, max_points, max_voxels)
    assert np.array_equal(c, pc) and np.array_equal(n, pn) and np.array_equal(v, pv)


def test_flagship_grid_properties():
    """Real 1440x1440x40 grid (where the reference CPU code segfaults): invariants of the semantics."""
    cfg = synth.CL_CONFIG
    pts = synth.lidar_points(seed=0)
    v, c, n = oracle.hard_voxelize(pts, cfg["voxel_size"], cfg["point_cloud_range"], 10, 160000)
    assert v.shape == (160000, 10, 5) and n.max() <= 10 and n.min() >= 1
    assert len(np.unique(c, axis=0)) == len(c)                       # distinct voxels
    assert np.all(c >= 0) and np.all(c < np.array([1440, 1440, 40]))
    dyn = oracle.dynamic_voxelize(pts, cfg["voxel_size"], cfg["point_cloud_range"])
    # every stored point lies in its voxel
    first = v[:, 0, :]
    assert np.array_equal(oracle.dynamic_voxelize(first, cfg["voxel_size"], cfg["point_cloud_range"]), c)
    # voxel order = order of first appearance among the first 160 000 distinct voxels
    _, first_idx = np.unique(dyn[dyn[:, 0] >= 0], axis=0, return_index=True)
    order = np.sort(first_idx)[:160000]
    assert np.array_equal(dyn[dyn[:, 0] >= 0][order], c)
    f = oracle.voxel_mean(v, n)
    assert np.allclose(f, v.sum(1) / n[:, None], atol=1e-4)
