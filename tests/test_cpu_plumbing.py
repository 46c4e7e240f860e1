"""BASELINE configs[0]: camera-only LSS, 1 camera, 256x704 image, 64x64 BEV on a box WITHOUT a GPU — the host-tensor route of
`bev_pool()` (the reference's device-agnostic QuickCumsum formulation, ops/bev_pool/bev_pool.py:8-34 + :83-97, in torch) and
`LSSTransform` end to end on it (models/vtransforms/lss.py:13-86, base.py:141-176).

The bar is the float64 oracle (<= 1e-4 abs, BASELINE.json north_star).  The host route is not a fallback of the GPU path: it is
only ever taken for host tensors (tests/test_host_mirror.py::test_product_has_no_oracle_import guards the other direction)."""
import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import synth
from bevfusion_amd.bev_pool import BevPoolPlan, QuickCumsum, bev_pool
from bevfusion_amd.vtransforms import LSSTransform

CFG = synth.LSS_SMALL_CONFIG


def small_inputs(seed=0, batch=1):
    inp = synth.bev_pool_inputs(CFG, batch=batch, channels=80, seed=seed)
    inp["feats"] *= 0.25
    coords, kept = oracle.bev_cell_index(inp["geom"], batch, inp["origin"], inp["dx"], inp["nx"])
    return inp, coords, kept


@pytest.mark.parametrize("batch", [1, 2])
def test_bev_pool_on_host_tensors_matches_the_float64_oracle(batch):
    inp, coords, kept = small_inputs(batch=batch)
    H, W, D = (int(v) for v in inp["nx"])
    ref = oracle.bev_pool(inp["feats"][kept], coords[kept], batch, D, H, W)
    x = torch.from_numpy(inp["feats"][kept]).requires_grad_(True)
    for c in (torch.from_numpy(coords[kept]), torch.from_numpy(coords[kept]).int()):     # int64 like base.py, int32 like the ext
        out = bev_pool(x, c, batch, D, H, W)
        assert tuple(out.shape) == (batch, 80, D, H, W) and out.dtype == torch.float32
        assert float(np.abs(out.detach().numpy() - ref).max()) <= 1e-4
    # backward of an interval sum: every kept row receives its cell's gradient (bev_pool_cuda.cu:61-84)
    g = torch.randn(out.shape, generator=torch.Generator().manual_seed(1))
    out.backward(g)
    c = coords[kept]
    want = g.numpy()[c[:, 3], :, c[:, 2], c[:, 0], c[:, 1]]
    assert np.array_equal(x.grad.numpy(), want)


def test_bev_pool_host_route_handles_unsorted_empty_and_single_rows():
    rng = np.random.default_rng(3)
    B, D, H, W, C = 2, 1, 5, 7, 8
    for n in (0, 1, 50):
        coords = np.stack([rng.integers(0, H, n), rng.integers(0, W, n), np.zeros(n, np.int64), rng.integers(0, B, n)], 1)
        feats = rng.standard_normal((n, C)).astype(np.float32)
        ref = oracle.bev_pool(feats, coords, B, D, H, W) if n else np.zeros((B, C, D, H, W))
        out = bev_pool(torch.from_numpy(feats), torch.from_numpy(coords), B, D, H, W)
        assert float(np.abs(out.numpy() - ref).max()) <= 1e-5


def test_quickcumsum_contract():
    """Sorted rows + ranks in, one (sum, coordinate) row per distinct rank out, in rank order; gradient = gather."""
    ranks = torch.tensor([2, 2, 5, 7, 7, 7])
    x = torch.arange(12, dtype=torch.float32).view(6, 2).requires_grad_(True)
    geom = torch.arange(24).view(6, 4)
    sums, cells = QuickCumsum.apply(x, geom, ranks)
    assert torch.equal(sums.detach(), torch.tensor([[2.0, 4.0], [4.0, 5.0], [24.0, 27.0]]))
    assert torch.equal(cells, geom[[1, 2, 5]])                         # the LAST row of each run (bev_pool.py:12-15)
    (sums * torch.tensor([[1.0], [10.0], [100.0]])).sum().backward()
    assert torch.equal(x.grad[:, 0], torch.tensor([1.0, 1.0, 10.0, 100.0, 100.0, 100.0]))


def test_gpu_objects_are_refused_on_the_host_route():
    inp, coords, kept = small_inputs()
    H, W, D = (int(v) for v in inp["nx"])
    with pytest.raises(RuntimeError, match="GPU"):
        BevPoolPlan.from_coords(torch.from_numpy(coords[kept]), 1, D, H, W)


def camera_matrices(batch):
    rig = synth.camera_rig(1)

    def m4(rot, trans):
        m = np.tile(np.eye(4, dtype=np.float32), (batch, 1, 1, 1))
        m[:, :, :3, :3], m[:, :, :3, 3] = rot, trans
        return torch.from_numpy(m)

    return (m4(rig["camera2lidar_rots"], rig["camera2lidar_trans"]), m4(rig["intrins"], 0.0),
            m4(rig["post_rots"], rig["post_trans"]), torch.eye(4).repeat(batch, 1, 1))


@pytest.mark.parametrize("batch", [1, 2])
def test_lss_transform_runs_without_a_gpu(batch):
    """configs[0] end to end: image features [B, 1, 256, 32, 88] -> [B, 80, 64, 64]; factored and materialised camera features
    agree, and both agree with a float64 segment sum of the explicit outer product."""
    torch.manual_seed(0)
    vt = LSSTransform(256, 80, CFG["image_size"], CFG["feature_size"], CFG["xbound"], CFG["ybound"], CFG["zbound"],
                      CFG["dbound"]).eval()
    c2l, K, ia, la = camera_matrices(batch)
    img = torch.randn(batch, 1, 256, 32, 88)
    outs = []
    with torch.no_grad():
        for fused in (True, False):
            vt.fused_cam_feats = fused
            outs.append(vt(img, None, None, c2l, None, None, None, K, c2l, ia, la))
        assert tuple(outs[0].shape) == (batch, 80, 64, 64)
        assert float((outs[0] - outs[1]).abs().max()) <= 1e-5
        # float64 reference from the module's own depthnet output and geometry
        geom = vt.get_geometry(c2l[..., :3, :3], c2l[..., :3, 3], K[..., :3, :3], ia[..., :3, :3], ia[..., :3, 3],
                               extra_rots=la[..., :3, :3], extra_trans=la[..., :3, 3])
        vt.fused_cam_feats = False
        vol = vt.get_cam_feats(img)                                       # [B, 1, D, fH, fW, C]
    g = geom.reshape(-1, 3).numpy()
    origin = (vt.bx - vt.dx / 2.0).numpy()
    coords, kept = oracle.bev_cell_index(g, batch, origin, vt.dx.numpy(), vt.nx.numpy())
    rows = vol.reshape(-1, 80).numpy()
    ref = oracle.bev_pool(rows[kept], coords[kept], batch, 1, 64, 64)[:, :, 0]
    assert float(np.abs(outs[0].numpy() - ref).max()) <= 1e-4


def test_lss_transform_on_host_is_differentiable():
    torch.manual_seed(1)
    vt = LSSTransform(16, 8, CFG["image_size"], (8, 22), CFG["xbound"], CFG["ybound"], CFG["zbound"], (1.0, 60.0, 4.0))
    c2l, K, ia, la = camera_matrices(1)
    img = torch.randn(1, 1, 16, 8, 22, requires_grad=True)
    y = vt(img, None, None, c2l, None, None, None, K, c2l, ia, la)
    y.square().mean().backward()
    assert img.grad is not None and float(img.grad.abs().sum()) > 0 and vt.depthnet.weight.grad is not None
