"""GPU: module-level parity — the view transforms and the SparseEncoder, built with the flagship config's
constants, against (a) the oracle substituted for the native ops inside the SAME module tree and (b) a
pure-torch restatement of the reference's PyTorch glue."""
import numpy as np
import pytest
import torch

import oracle
from bevfusion_amd import spconv, synth
from bevfusion_amd.sparse_encoder import SparseEncoder
from bevfusion_amd.spconv import functional as Fsp
from bevfusion_amd.spconv import ops as sops
from bevfusion_amd.vtransforms import FactoredCamFeats, DepthLSSTransform, LSSTransform

pytestmark = pytest.mark.gpu


def _rig_tensors(n_cam, B, dev):
    rig = synth.camera_rig(n_cam)
    eye = torch.eye(4, device=dev)

    def mat4(rot, trans):
        m = eye.repeat(B, n_cam, 1, 1).clone()
        m[:, :, :3, :3] = torch.from_numpy(rot).to(dev)
        m[:, :, :3, 3] = torch.from_numpy(trans).to(dev)
        return m

    camera2lidar = mat4(rig["camera2lidar_rots"], rig["camera2lidar_trans"])
    intr = mat4(rig["intrins"], np.zeros((n_cam, 3), np.float32))
    aug = mat4(rig["post_rots"], rig["post_trans"])
    lidar_aug = eye.repeat(B, 1, 1).clone()
    lidar2camera = torch.inverse(camera2lidar)
    lidar2image = intr @ lidar2camera
    return dict(camera2ego=camera2lidar.clone(), lidar2ego=eye.repeat(B, 1, 1), lidar2camera=lidar2camera,
                lidar2image=lidar2image, camera_intrinsics=intr, camera2lidar=camera2lidar, img_aug_matrix=aug,
                lidar_aug_matrix=lidar_aug)


def _reference_bev_pool_torch(vt, geom, x):
    """BaseTransform.bev_pool (base.py:141-176) restated with index_add_ in float64 instead of the extension."""
    B, N, D, H, W, C = x.shape
    Nprime = B * N * D * H * W
    x = x.reshape(Nprime, C).double()
    g = ((geom - (vt.bx - vt.dx / 2.0)) / vt.dx).long().view(Nprime, 3)
    batch_ix = torch.arange(B, device=x.device).repeat_interleave(Nprime // B).view(-1, 1)
    g = torch.cat((g, batch_ix), 1)
    kept = ((g[:, 0] >= 0) & (g[:, 0] < vt.nx[0]) & (g[:, 1] >= 0) & (g[:, 1] < vt.nx[1]) & (g[:, 2] >= 0)
            & (g[:, 2] < vt.nx[2]))
    x, g = x[kept], g[kept]
    nx, ny, nz = (int(v) for v in vt.nx)
    out = torch.zeros(B * nz * nx * ny, C, dtype=torch.float64, device=x.device)
    lin = ((g[:, 3] * nz + g[:, 2]) * nx + g[:, 0]) * ny + g[:, 1]
    out.index_add_(0, lin, x)
    out = out.view(B, nz, nx, ny, C).permute(0, 4, 1, 2, 3)
    return torch.cat(out.unbind(dim=2), 1)


def test_lss_transform_forward_config1(dev):
    """BASELINE config 1: camera-only LSS, 1 camera, 256x704, 64x64 BEV."""
    cfg = synth.LSS_SMALL_CONFIG
    torch.manual_seed(0)
    vt = LSSTransform(256, 80, cfg["image_size"], cfg["feature_size"], cfg["xbound"], cfg["ybound"], cfg["zbound"],
                      cfg["dbound"], downsample=1).to(dev).eval()
    B, n_cam = 2, 1
    mats = _rig_tensors(n_cam, B, dev)
    img = torch.randn(B, n_cam, 256, 32, 88, device=dev)
    with torch.no_grad():
        out = vt(img, None, None, **mats)
        assert tuple(out.shape) == (B, 80, 64, 64)
        geom = vt.get_geometry(mats["camera2lidar"][..., :3, :3], mats["camera2lidar"][..., :3, 3],
                               mats["camera_intrinsics"][..., :3, :3], mats["img_aug_matrix"][..., :3, :3],
                               mats["img_aug_matrix"][..., :3, 3], extra_rots=mats["lidar_aug_matrix"][..., :3, :3],
                               extra_trans=mats["lidar_aug_matrix"][..., :3, 3])
        vt.fused_cam_feats = False                       # the reference's materialised [B,N,D,fH,fW,C] volume
        feats = vt.get_cam_feats(img)
        ref = _reference_bev_pool_torch(vt, geom, feats)
        out_unfused = vt(img, None, None, **mats)
    assert float((out.double() - ref).abs().max()) <= 1e-4          # `out` ran the fused depth (x) context kernel
    assert float((out_unfused.double() - ref).abs().max()) <= 1e-4


def test_depth_lss_transform_forward_flagship_shapes(dev):
    """C+L config camera branch: 6 cameras, D=118, 360x360 -> downsample 2 -> 180x180; cached plan == fresh plan."""
    cfg = synth.CL_CONFIG
    torch.manual_seed(0)
    vt = DepthLSSTransform(256, 80, cfg["image_size"], cfg["feature_size"], cfg["xbound"], cfg["ybound"],
                           cfg["zbound"], cfg["dbound"], downsample=2).to(dev).eval()
    B, n_cam = 1, 6
    mats = _rig_tensors(n_cam, B, dev)
    mats["sensor2ego"] = mats.pop("camera2ego")
    mats["cam_intrinsic"] = mats.pop("camera_intrinsics")
    img = torch.randn(B, n_cam, 256, 32, 88, device=dev)
    pts = [torch.from_numpy(synth.lidar_points(seed=0, sweeps=2)).to(dev)]
    with torch.no_grad():
        out = vt(img, pts, None, metas=None, **mats)
        assert tuple(out.shape) == (B, 80, 180, 180) and torch.isfinite(out).all()
        depth = vt.depth_raster(img, pts, mats["lidar2image"], mats["img_aug_matrix"], mats["lidar_aug_matrix"])
        assert tuple(depth.shape) == (B, 6, 1, 256, 704) and int((depth > 0).sum()) > 1000
        geom = vt.get_geometry(mats["camera2lidar"][..., :3, :3], mats["camera2lidar"][..., :3, 3],
                               mats["cam_intrinsic"][..., :3, :3], mats["img_aug_matrix"][..., :3, :3],
                               mats["img_aug_matrix"][..., :3, 3], extra_rots=mats["lidar_aug_matrix"][..., :3, :3],
                               extra_trans=mats["lidar_aug_matrix"][..., :3, 3])
        fz = vt.get_cam_feats(img, depth)                # inference default: depth / context kept factored
        assert isinstance(fz, FactoredCamFeats) and tuple(fz.depth.shape) == (B, 6, 118, 32, 88)
        vt.fused_cam_feats = False
        feats = vt.get_cam_feats(img, depth)
        assert tuple(feats.shape) == (B, 6, 118, 32, 88, 80)
        assert torch.allclose(fz.materialize(), feats, rtol=1e-4, atol=1e-6)   # two passes through the conv stack (MIOpen)
        pooled = vt.bev_pool(geom, feats)
        ref = _reference_bev_pool_torch(vt, geom, feats)
        assert tuple(pooled.shape) == (B, 80, 360, 360)
        assert float((pooled.double() - ref).abs().max()) <= 1e-4
        # fused depth (x) context -> BEV: same sums without the 638 MB volume (fp32 and bf16 context)
        pooled_f = vt.bev_pool(geom, fz)
        assert tuple(pooled_f.shape) == (B, 80, 360, 360)
        assert float((pooled_f.double() - ref).abs().max()) <= 1e-4
        plan = vt.make_plan(geom, B)
        ctx16 = fz.ctx.permute(0, 1, 3, 4, 2).contiguous().bfloat16()
        ref16 = _reference_bev_pool_torch(vt, geom, FactoredCamFeats(fz.depth, ctx16.float().permute(0, 1, 4, 2, 3)).materialize())
        got16 = plan.launch_fused(fz.depth.contiguous(), ctx16.view(-1, 80), 118, 32, 88).permute(0, 4, 1, 2, 3)[:, :, 0]
        assert float((got16.double() - ref16).abs().max()) <= 1e-4
        # static calibration: the precompute is built once and reused; same bits as a fresh plan
        vt.cache_geometry = True
        a = vt.bev_pool(geom, feats)
        plan = vt._plan
        b = vt.bev_pool(geom, feats)
        assert vt._plan is plan and torch.equal(a, b) and torch.equal(a, pooled)
        # (whole-module outputs are not compared bit-for-bit: the depth raster is last-writer-wins on
        #  colliding LiDAR points, in the reference as here)


# ---- SparseEncoder with the oracle substituted for the native ops -------------------------------------------
class _OracleRulebook:
    def __init__(self, oi, pairs, num, n_in, K, subm, oshape, dev):
        self.out_indices = torch.from_numpy(oi).to(dev)
        self.pairs, self.num = pairs, num
        self.num_out, self.num_in, self.kernel_volume, self.subm = oi.shape[0], n_in, K, subm
        self.out_spatial_shape = list(oshape)


def _oracle_build(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1, subm=False,
                  transpose=False, out_padding=0):
    assert not transpose
    ks, st, pd = (sops._as_list(v, 3) for v in (ksize, stride, padding))
    ind = indices.cpu().numpy()
    oi, pairs, num, oshape = oracle.get_indice_pairs(ind, batch_size, spatial_shape, ks, st, pd, [1, 1, 1], int(subm),
                                                     order="cuda")
    return _OracleRulebook(oi, pairs, num, ind.shape[0], int(np.prod(ks)), subm, oshape, indices.device)


def _oracle_conv(features, filters, rb):
    out = oracle.indice_conv(features.detach().float().cpu().numpy(), filters.detach().float().cpu().numpy(), rb.pairs,
                             rb.num, rb.num_out)
    return torch.from_numpy(out).to(features.device).to(features.dtype)


def _small_encoder(dev, dtype):
    torch.manual_seed(1)
    enc = SparseEncoder(5, [40, 40, 41], order=["conv", "norm", "act"], output_channels=32,
                        encoder_channels=[[16, 16, 32], [32, 32, 64], [64, 64, 64], [64, 64]],
                        encoder_paddings=[[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]], block_type="basicblock")
    for m in enc.modules():                       # non-trivial BN statistics
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    return enc.to(dev).to(dtype).eval()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float16, 3e-2)])
def test_sparse_encoder_vs_oracle_substitution(dev, monkeypatch, dtype, tol):
    rng = np.random.default_rng(0)
    B, shape = 2, (40, 40, 41)
    idx = []
    for b in range(B):
        lin = rng.choice(int(np.prod(shape)), size=2500, replace=False)
        idx.append(np.concatenate([np.full((2500, 1), b), np.stack(np.unravel_index(lin, shape), 1)], 1))
    coors = np.concatenate(idx).astype(np.int32)
    rng.shuffle(coors, axis=0)
    feats = rng.standard_normal((coors.shape[0], 5)).astype(np.float32)
    enc = _small_encoder(dev, dtype)
    x, c = torch.from_numpy(feats).to(dev).to(dtype), torch.from_numpy(coors).to(dev)
    with torch.no_grad():
        got = enc(x, c, B)
    assert tuple(got.shape) == (B, 32 * 2, 5, 5)
    # reference: the module-by-module path with the CPU oracle substituted for the native ops
    # (`got` above ran the fused inference path for fp16, the module path for fp32)
    enc.fused_inference = False
    monkeypatch.setattr(sops, "build_rulebook", _oracle_build)
    monkeypatch.setattr(Fsp, "rulebook_conv", _oracle_conv)
    with torch.no_grad():
        ref = enc(x, c, B)
    err = float((got.float() - ref.float()).abs().max())
    assert err <= tol * (1 + float(ref.float().abs().max())), err


def test_sparse_encoder_flagship_size_indices_and_dtypes(dev):
    """~300k LiDAR points -> 160k voxels -> the 21-conv VoxelNet at 1440x1440x41: per-stage active sets equal the
    oracle's (bit-exact indices at full size), fp16 output tracks fp32, dense output is [B, 256, 180, 180]."""
    from bevfusion_amd.voxel import voxelize_batch

    cfg = synth.CL_CONFIG
    pts = synth.lidar_points(seed=0)
    feats, coords, sizes = voxelize_batch([torch.from_numpy(pts).to(dev)], cfg["voxel_size"], cfg["point_cloud_range"],
                                          10, 160000)
    assert feats.shape[0] == 160000
    torch.manual_seed(0)
    enc = SparseEncoder(5, list(cfg["sparse_shape"]), order=["conv", "norm", "act"], output_channels=128,
                        encoder_channels=[[16, 16, 32], [32, 32, 64], [64, 64, 128], [128, 128]],
                        encoder_paddings=[[0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]],
                        block_type="basicblock").to(dev).eval()
    with torch.no_grad():
        out32 = enc(feats, coords, 1)
        enc.half()
        out16 = enc(feats.half(), coords, 1)
    assert tuple(out32.shape) == (1, 256, 180, 180) and torch.isfinite(out32).all()
    rel = float((out16.float() - out32).abs().max() / (out32.abs().max() + 1e-6))
    assert rel < 5e-2, rel
    # stage-by-stage active sets vs the oracle
    ind = coords.cpu().numpy()
    shape = list(cfg["sparse_shape"])
    for ks, st, pd in [((3, 3, 3), (2, 2, 2), (1, 1, 1)), ((3, 3, 3), (2, 2, 2), (1, 1, 1)),
                       ((3, 3, 3), (2, 2, 2), (1, 1, 0)), ((1, 1, 3), (1, 1, 2), (0, 0, 0))]:
        rb = spconv.build_rulebook(torch.from_numpy(ind).to(dev), 1, shape, list(ks), list(st), list(pd), 1, False)
        oi, opairs, onum, oshape = oracle.get_indice_pairs(ind, 1, shape, ks, st, pd, [1, 1, 1], 0, order="cuda")
        assert np.array_equal(rb.out_indices.cpu().numpy(), oi)
        _, gnum = rb.indice_pairs()
        assert np.array_equal(gnum.cpu().numpy(), onum)
        ind, shape = oi, list(oshape)
    assert shape == [180, 180, 2]


def test_plan_cache_is_keyed_on_the_calibration(dev):
    """`cache_geometry` must not reuse a pooling plan across a change of calibration / augmentation matrices
    (same N', same B), must be ignored while training, and can be dropped explicitly."""
    cfg = synth.LSS_SMALL_CONFIG
    torch.manual_seed(0)
    vt = LSSTransform(256, 80, cfg["image_size"], cfg["feature_size"], cfg["xbound"], cfg["ybound"], cfg["zbound"],
                      cfg["dbound"], downsample=1).to(dev).eval()
    B, n_cam = 2, 1
    mats = _rig_tensors(n_cam, B, dev)
    img = torch.randn(B, n_cam, 256, 32, 88, device=dev)
    with torch.no_grad():
        fresh = vt(img, None, None, **mats)
        vt.cache_geometry = True
        a = vt(img, None, None, **mats)
        plan = vt._plan
        assert plan is not None and torch.equal(a, fresh)
        b = vt(img, None, None, **mats)                                  # unchanged calibration: plan reused, no geometry pass
        assert vt._plan is plan and torch.equal(a, b)
        c = vt(img, None, None, calibration_id="rig-0", **mats)          # caller-supplied identity: its own key
        plan_id = vt._plan
        assert torch.equal(c, fresh)
        vt(img, None, None, calibration_id="rig-0", **mats)
        assert vt._plan is plan_id
        # a LiDAR augmentation (same shapes): stale plan must NOT be used
        mats2 = dict(mats)
        la = mats["lidar_aug_matrix"].clone()
        ang = 0.3
        la[:, :3, :3] = torch.tensor([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]],
                                     dtype=torch.float32, device=dev)
        mats2["lidar_aug_matrix"] = la
        vt.cache_geometry = False
        want = vt(img, None, None, **mats2)
        vt.cache_geometry = True
        got = vt(img, None, None, **mats2)
        assert vt._plan is not plan_id and torch.equal(got, want) and not torch.equal(got, fresh)
        vt.invalidate_plan()
        assert vt._plan is None
    vt.train()
    out = vt(img, None, None, **mats)                                    # training: never cached
    assert vt._plan is None and out.requires_grad
