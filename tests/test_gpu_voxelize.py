"""GPU parity: HIP hard voxelization (through the C ABI) vs the CPU oracle and the reference-CPU golden
vectors.  Bar: voxel order, coordinates, counts and the copied point features are all BIT-EXACT."""
import glob
import os

import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import synth
from bevfusion_amd.voxel import Voxelization, voxel_layer, voxelization, voxelize_batch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _run(pts, vs, cr, mp, mv, dev):
    v, c, n = voxelization(torch.from_numpy(pts).to(dev), list(vs), list(cr), mp, mv, True)
    return v.cpu().numpy(), c.cpu().numpy(), n.cpu().numpy()


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "voxel_ref_*.npz"))))
def test_against_reference_cpu_golden(dev, path):
    z = np.load(path)
    v, c, n = _run(z["points"], z["voxel_size"], z["coors_range"], int(z["max_points"]), int(z["max_voxels"]), dev)
    assert np.array_equal(c, z["coors"]) and np.array_equal(n, z["num_points_per_voxel"])
    assert np.array_equal(v, z["voxels"])
    dyn = torch.zeros(z["points"].shape[0], 3, dtype=torch.int32, device=dev)
    voxel_layer.dynamic_voxelize(torch.from_numpy(z["points"]).to(dev), dyn, list(z["voxel_size"]),
                                 list(z["coors_range"]), 3)
    assert np.array_equal(dyn.cpu().numpy(), z["dynamic_coors"])


@pytest.mark.parametrize("n,mp,mv,f", [(1, 5, 10, 4), (5000, 10, 100000, 5), (5000, 2, 300, 5), (70000, 1, 50, 3),
                                       (4097, 3, 4000, 7), (200000, 10, 20000, 5)])
def test_random_clouds_vs_oracle(dev, n, mp, mv, f):
    rng = np.random.default_rng(n + mp)
    pts = rng.uniform(-1.3, 1.3, size=(n, f)).astype(np.float32)
    pts[::5, :3] = pts[0, :3]  # a crowded voxel: exercises the max_points cap and slot order
    vs, cr = [0.1, 0.05, 0.4], [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
    v, c, m = _run(pts, vs, cr, mp, mv, dev)
    ov, oc, om = oracle.hard_voxelize(pts, vs, cr, mp, mv)
    assert np.array_equal(c, oc) and np.array_equal(m, om) and np.array_equal(v, ov)


def test_edge_cases(dev):
    vs, cr = [0.5, 0.5, 0.5], [0.0, 0.0, 0.0, 4.0, 4.0, 4.0]
    # empty cloud
    v, c, n = _run(np.zeros((0, 4), np.float32), vs, cr, 5, 10, dev)
    assert v.shape == (0, 5, 4) and c.shape == (0, 3) and n.shape == (0,)
    # every point outside the range, NaN and inf included
    pts = np.array([[-1, 0, 0, 1], [0, 9, 0, 1], [np.nan, 1, 1, 1], [np.inf, 1, 1, 1], [1, 1, 4.0, 1]], np.float32)
    v, c, n = _run(pts, vs, cr, 5, 10, dev)
    assert v.shape[0] == 0
    # points exactly on the lower bound are in, on the upper bound are out
    pts = np.array([[0, 0, 0, 7], [4.0, 0, 0, 8], [3.9999, 3.9999, 3.9999, 9]], np.float32)
    v, c, n = _run(pts, vs, cr, 5, 10, dev)
    assert np.array_equal(c, np.array([[0, 0, 0], [7, 7, 7]], np.int32)) and list(n) == [1, 1]
    ov, oc, on = oracle.hard_voxelize(pts, vs, cr, 5, 10)
    assert np.array_equal(c, oc) and np.array_equal(v, ov)


def test_module_interface_and_caps(dev):
    cfg = synth.CL_CONFIG
    vox = Voxelization(list(cfg["voxel_size"]), list(cfg["point_cloud_range"]), cfg["max_num_points"],
                       list(cfg["max_voxels"]))
    assert vox.grid_size.tolist() == [1440, 1440, 40] and [int(t) for t in vox.pcd_shape] == [1440, 1440, 1]
    assert vox.max_voxels == (120000, 160000)
    pts = synth.lidar_points(seed=1)
    vox.eval()
    v, c, n = vox(torch.from_numpy(pts).to(dev))
    ov, oc, on = oracle.hard_voxelize(pts, cfg["voxel_size"], cfg["point_cloud_range"], 10, 160000)
    assert v.shape[0] == 160000
    assert np.array_equal(c.cpu().numpy(), oc) and np.array_equal(n.cpu().numpy(), on)
    assert np.array_equal(v.cpu().numpy(), ov)
    vox.train()
    v, c, n = vox(torch.from_numpy(pts).to(dev))
    assert v.shape[0] == 120000 and np.array_equal(c.cpu().numpy(), oc[:120000])


def test_voxelize_batch_matches_bevfusion_voxelize(dev):
    """bevfusion.py:169-197: per-sample voxelize, (b,x,y,z) coords, mean over points."""
    cfg = synth.CL_CONFIG
    clouds = [synth.lidar_points(seed=s, sweeps=3) for s in (0, 1, 2)]
    feats, coords, sizes = voxelize_batch([torch.from_numpy(p).to(dev) for p in clouds], cfg["voxel_size"],
                                          cfg["point_cloud_range"], 10, 160000)
    of, oc, os_ = oracle.voxelize_batch(clouds, cfg["voxel_size"], cfg["point_cloud_range"], 10, 160000)
    assert np.array_equal(coords.cpu().numpy(), oc) and np.array_equal(sizes.cpu().numpy(), os_)
    assert np.max(np.abs(feats.cpu().numpy() - of)) <= 1e-5 * max(1.0, np.abs(of).max())
    # and against the three-step reference formulation built from the drop-in op
    v, c, n = voxelization(torch.from_numpy(clouds[1]).to(dev), list(cfg["voxel_size"]),
                           list(cfg["point_cloud_range"]), 10, 160000, True)
    ref = v.sum(dim=1) / n.type_as(v).view(-1, 1)
    sel = coords[:, 0] == 1
    assert torch.allclose(feats[sel], ref, atol=1e-4, rtol=1e-6)


def test_flagship_size_properties(dev):
    """~310k points on the 1440x1440x40 grid: permutation sensitivity and idempotence properties."""
    cfg = synth.CL_CONFIG
    pts = synth.lidar_points(seed=0)
    args = (list(cfg["voxel_size"]), list(cfg["point_cloud_range"]), 10, 400000, True)
    v, c, n = voxelization(torch.from_numpy(pts).to(dev), *args)
    m = v.shape[0]
    assert int(n.sum()) <= pts.shape[0] and int(n.max()) <= 10
    # uncapped voxel SET does not depend on the point order; the ORDER does
    perm = np.random.default_rng(1).permutation(pts.shape[0])
    v2, c2, n2 = voxelization(torch.from_numpy(pts[perm]).to(dev), *args)
    assert v2.shape[0] == m
    key = lambda t: (t[:, 0].long() * 1440 + t[:, 1].long()) * 40 + t[:, 2].long()
    assert torch.equal(torch.sort(key(c))[0], torch.sort(key(c2))[0])
    # idempotence: voxelizing the first point of every voxel recreates the same voxels in the same order
    v3, c3, n3 = voxelization(v[:, 0, :].contiguous(), *args)
    assert torch.equal(c3, c) and torch.all(n3 == 1)


def test_batch_packing_without_host_sync(dev):
    """voxelize_batch_device == voxelize_batch (host-synchronised concatenation) on the live rows, for B = 1 and B = 3."""
    from bevfusion_amd.voxel import voxelize_batch, voxelize_batch_device

    vs, pr = [0.5, 0.5, 0.5], [0.0, 0.0, 0.0, 20.0, 20.0, 4.0]
    for B in (1, 3):
        pts = []
        for b in range(B):
            rng = np.random.default_rng(100 + b)
            p = rng.random((3000 + 500 * b, 5)).astype(np.float32) * np.array([22, 22, 4.4, 1, 1], np.float32) - 1.0
            pts.append(torch.from_numpy(p).to(dev))
        f0, c0, s0 = voxelize_batch(pts, vs, pr, 5, 1500)
        f1, c1, s1, tot = voxelize_batch_device(pts, vs, pr, 5, 1500)
        n = int(tot)
        assert n == f0.shape[0] and f1.shape[0] == B * 1500
        assert torch.equal(f1[:n], f0) and torch.equal(c1[:n], c0) and torch.equal(s1[:n], s0)


def _per_sample_then_cat(pts, vs, pr, mp, mv):
    """The reference formulation through the one-sample entry: bevfusion.py:176-191."""
    from bevfusion_amd import voxel as V

    f, c, s, counts = V._voxelize_mean_lanes(pts, vs, pr, mp, mv)
    cnt = counts.tolist()
    cat = lambda t: torch.cat([t[k, : cnt[k]] for k in range(len(pts))], 0)
    return cat(f), cat(c), cat(s), cnt


@pytest.mark.parametrize("case", ["ragged", "empty_middle", "all_empty", "capped", "one", "tile_edges"])
def test_batched_entry_is_bit_identical_to_per_sample_calls(dev, case):
    """`bevamd_voxelize_mean_batch` (one segmented sort for the batch) == one `bevamd_voxelize_mean` per sample: features,
    coordinates, counts and the first-appearance voxel order, padded and packed layouts; ragged / empty samples, samples
    beyond max_voxels, sizes on the sort's 1024-key tile edges."""
    from bevfusion_amd import voxel as V

    vs, pr, mp = [0.5, 0.5, 0.5], [0.0, 0.0, 0.0, 20.0, 20.0, 4.0], 5
    sizes = {"ragged": [3000, 1, 4097, 777, 2048], "empty_middle": [500, 0, 0, 900], "all_empty": [0, 0],
             "capped": [6000, 200, 6000], "one": [2500], "tile_edges": [1024, 1023, 1025, 2048, 1]}[case]
    mv = 300 if case == "capped" else 1500
    pts = []
    for b, n in enumerate(sizes):
        rng = np.random.default_rng(7 * b + n)
        p = rng.random((n, 5)).astype(np.float32) * np.array([22, 22, 4.4, 1, 1], np.float32) - 1.0   # some out of range
        pts.append(torch.from_numpy(p).to(dev))
    f0, c0, s0, cnt = _per_sample_then_cat(pts, vs, pr, mp, mv)
    if case == "capped":
        assert cnt[0] == mv and cnt[1] < mv

    f, c, s, counts, total = V._voxelize_mean_batch(pts, vs, pr, mp, mv, packed=True)
    assert counts.tolist() == cnt and int(total) == sum(cnt)
    n = sum(cnt)
    assert torch.equal(c[:n], c0) and torch.equal(s[:n], s0) and torch.equal(f[:n], f0)

    f, c, s, counts, total = V._voxelize_mean_batch(pts, vs, pr, mp, mv, packed=False)
    assert counts.tolist() == cnt and int(total) == sum(cnt)
    row = 0
    for b, k in enumerate(cnt):
        sl = slice(b * mv, b * mv + k)
        assert torch.equal(c[sl], c0[row:row + k]) and torch.equal(s[sl], s0[row:row + k])
        assert torch.equal(f[sl], f0[row:row + k])
        row += k


def test_batched_entry_rejects_bad_arguments(dev):
    from bevfusion_amd import voxel as V

    vs, pr = [0.5, 0.5, 0.5], [0.0, 0.0, 0.0, 20.0, 20.0, 4.0]
    good = torch.zeros((4, 5), device=dev)
    with pytest.raises(ValueError):
        V._voxelize_mean_batch([good, torch.zeros((4, 4), device=dev)], vs, pr, 5, 100, packed=True)
    with pytest.raises(RuntimeError):
        V._voxelize_mean_batch([good] * 65, vs, pr, 5, 100, packed=True)
