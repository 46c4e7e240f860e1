"""Camera -> BEV view transforms — mirror of `mmdet3d/models/vtransforms/{base,lss,depth_lss}.py`.

Same class names, constructor kwargs, attributes (`dx, bx, nx, frustum, D, C`), sub-module names (`depthnet`,
`dtransform`, `downsample`) and therefore state-dict keys; registered as VTRANSFORMS entries `LSSTransform` and
`DepthLSSTransform`, so the reference's YAML configs build them unchanged.

What changed underneath (`BaseTransform.bev_pool`, base.py:141-176):
  * index computation + batch index + range mask + rank + sort + interval search happen in ONE device pipeline
    (`BevPoolPlan.from_geometry`) with no boolean gather of the feature volume and no host sync; with
    `cache_geometry=True` (static calibration at inference; keyed on the calibration / augmentation matrices or a
    caller-supplied `calibration_id`, ignored while training) it is done once and reused;
  * the reduction reads the feature rows through the sort permutation (no `feats[indices]` copy) and writes
    every BEV cell once; the result is returned as a channels-last VIEW of that buffer ([B, C*D, H, W] values
    identical to the reference's `torch.cat(x.unbind(dim=2), 1)`).
Reference defects handled as SURVEY.md lists them: D2 (`get_cam_feats` arity) and D3 (DepthLSSTransform gets
scalar 1-channel depth, no height expansion — the semantics the released checkpoint's `Conv2d(1, 8, 1)` needs).
"""
import ctypes
import warnings
from typing import Tuple

import torch
from torch import nn

from . import _capi
from .bev_pool import BevPoolPlan, bev_pool
from .registry import register_everywhere

__all__ = ["BaseTransform", "BaseDepthTransform", "LSSTransform", "DepthLSSTransform", "gen_dx_bx"]


def gen_dx_bx(xbound, ybound, zbound):
    """base.py:15-21."""
    dx = torch.Tensor([row[2] for row in [xbound, ybound, zbound]])
    bx = torch.Tensor([row[0] + row[2] / 2.0 for row in [xbound, ybound, zbound]])
    nx = torch.LongTensor([(row[1] - row[0]) / row[2] for row in [xbound, ybound, zbound]])
    return dx, bx, nx


_RASTER_MAPS = {}          # (device, bytes, stream) -> zeroed map; insertion-ordered, at most _RASTER_MAPS_MAX entries
_RASTER_MAPS_MAX = 8

class FactoredCamFeats:
    """`get_cam_feats` result kept factored: depth [B, N, D, fH, fW] (softmax) and context [B, N, C, fH, fW].  Equivalent
    to the reference's [B, N, D, fH, fW, C] outer product (depth_lss.py:92-97), which `bev_pool` then never builds."""

    def __init__(self, depth, ctx):
        self.depth, self.ctx = depth, ctx

    def materialize(self):
        B, N, D, fH, fW = self.depth.shape
        x = self.depth.unsqueeze(2) * self.ctx.unsqueeze(3)          # [B, N, C, D, fH, fW]
        return x.permute(0, 1, 3, 4, 5, 2)


def _fp32(*tensors):
    """mmcv's @force_fp32: half/bf16 tensor arguments are cast to float."""
    return [t.float() if torch.is_tensor(t) and t.is_floating_point() and t.dtype != torch.float32 else t for t in tensors]


class BaseTransform(nn.Module):
    _warned_sync = False   # the "fingerprinting the calibration costs a device-to-host copy" warning is issued once per process

    def __init__(self, in_channels: int, out_channels: int, image_size: Tuple[int, int], feature_size: Tuple[int, int],
                 xbound: Tuple[float, float, float], ybound: Tuple[float, float, float],
                 zbound: Tuple[float, float, float], dbound: Tuple[float, float, float], use_points="lidar",
                 depth_input="scalar", height_expand=True, add_depth_features=True) -> None:
        super().__init__()
        self.in_channels = in_channels
        self.image_size = image_size
        self.feature_size = feature_size
        self.xbound = xbound
        self.ybound = ybound
        self.zbound = zbound
        self.dbound = dbound
        self.use_points = use_points
        assert use_points in ["radar", "lidar"]
        self.depth_input = depth_input
        assert depth_input in ["scalar", "one-hot"]
        self.height_expand = height_expand
        self.add_depth_features = add_depth_features

        dx, bx, nx = gen_dx_bx(self.xbound, self.ybound, self.zbound)
        self.dx = nn.Parameter(dx, requires_grad=False)
        self.bx = nn.Parameter(bx, requires_grad=False)
        self.nx = nn.Parameter(nx, requires_grad=False)

        self.C = out_channels
        self.frustum = self.create_frustum()
        self.D = self.frustum.shape[0]
        self.fp16_enabled = False
        # MI355X-native knobs (not in the reference): reuse the bev_pool precompute across frames; keep depth and
        # context factored and fold their outer product into the pooling kernel, forward and backward
        self.cache_geometry = False     # eval only; keyed on the calibration (see _calibration_key), never on shapes alone
        self._plan = None
        self._plan_key = None
        self._pending_key = None
        self._plan_geom = None          # the geometry tensor a directly-keyed plan was built from (kept alive with the plan)
        self.fused_cam_feats = True
        # 3x3 inverses of the calibration on the device path: False = bevamd_lss_camera_matrices / bevamd_mat3_inverse
        # (fp64 adjugate, no LAPACK call, no host sync); True = torch.inverse like the reference, call for call
        self.lapack_inverse = False

    def create_frustum(self):
        """base.py:66-89."""
        iH, iW = self.image_size
        fH, fW = self.feature_size
        ds = torch.arange(*self.dbound, dtype=torch.float).view(-1, 1, 1).expand(-1, fH, fW)
        D, _, _ = ds.shape
        xs = torch.linspace(0, iW - 1, fW, dtype=torch.float).view(1, 1, fW).expand(D, fH, fW)
        ys = torch.linspace(0, iH - 1, fH, dtype=torch.float).view(1, fH, 1).expand(D, fH, fW)
        frustum = torch.stack((xs, ys, ds), -1)
        return nn.Parameter(frustum, requires_grad=False)

    def get_geometry(self, camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans, **kwargs):
        """base.py:92-135 -> [B, N, D, fH, fW, 3] frustum points in the lidar frame."""
        camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans = _fp32(
            camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans)
        B, N, _ = camera2lidar_trans.shape
        if camera2lidar_trans.is_cuda and not torch.is_grad_enabled():
            return self._get_geometry_native(camera2lidar_rots, camera2lidar_trans, intrins, post_rots, post_trans, **kwargs)
        # host tensors / autograd: the reference's chain of broadcast matmuls, same operations in the same order (torch picks
        # the same kernels, so the bits match the reference on CPU: tests/test_oracle_vtransform.py)
        per_cam = (B, N, 1, 1, 1)
        inv_aug = torch.inverse(post_rots).view(*per_cam, 3, 3)
        pix = (self.frustum - post_trans.view(*per_cam, 3)).unsqueeze(-1)           # undo the image augmentation ...
        pix = inv_aug.matmul(pix)
        cam = torch.cat((pix[..., :2, :] * pix[..., 2:3, :], pix[..., 2:3, :]), 5)   # ... (u*d, v*d, d) ...
        to_lidar = camera2lidar_rots.matmul(torch.inverse(intrins)).view(*per_cam, 3, 3)
        points = to_lidar.matmul(cam).squeeze(-1)                                    # ... unproject and move to the lidar frame
        points += camera2lidar_trans.view(*per_cam, 3)
        if "extra_rots" in kwargs:                                                   # LiDAR augmentation (base.py:122-133)
            rot = _fp32(kwargs["extra_rots"])[0].view(B, 1, 1, 1, 1, 3, 3).repeat(1, N, 1, 1, 1, 1, 1)
            points = rot.matmul(points.unsqueeze(-1)).squeeze(-1)
        if "extra_trans" in kwargs:
            points += _fp32(kwargs["extra_trans"])[0].view(B, 1, 1, 1, 1, 3).repeat(1, N, 1, 1, 1, 1)
        return points

    def _get_geometry_native(self, c2l_rots, c2l_trans, intrins, post_rots, post_trans, **kwargs):
        """Same result from two launches (csrc/vtransform.hip): `lss_camera_matrices_kernel` (the B*N 3x3 inverses and the
        rot @ inverse(intrinsics) products, base.py:106/118 — no torch.inverse, no host sync) + `lss_geometry_kernel`
        (the [B,N,D,fH,fW,3] point cloud, fp32 op for op in the reference's order)."""
        lib = _capi.load()
        B, N, _ = c2l_trans.shape
        dev = c2l_trans.device
        if self.lapack_inverse:
            post_rot_inv = torch.inverse(post_rots).reshape(B * N, 3, 3).contiguous()
            combine = c2l_rots.matmul(torch.inverse(intrins)).reshape(B * N, 3, 3).contiguous()
        else:
            post_rot_inv = torch.empty((B * N, 3, 3), dtype=torch.float32, device=dev)
            combine = torch.empty((B * N, 3, 3), dtype=torch.float32, device=dev)
            pr, cr, kr = (m.reshape(B * N, 3, 3).contiguous() for m in (post_rots, c2l_rots, intrins))   # 54 floats each
            with torch.cuda.device(dev):
                rc = lib.bevamd_lss_camera_matrices(_capi.ptr(pr), _capi.ptr(cr), _capi.ptr(kr), 9, 3, B * N,
                                                    _capi.ptr(post_rot_inv), _capi.ptr(combine), _capi.stream_ptr(dev))
            _capi.check(rc, "lss_camera_matrices")
        extra_rots = _fp32(kwargs["extra_rots"])[0].reshape(B, 3, 3).contiguous() if "extra_rots" in kwargs else None
        extra_trans = _fp32(kwargs["extra_trans"])[0].reshape(B, 3).contiguous() if "extra_trans" in kwargs else None
        return self.geometry_from_camera_matrices(post_rot_inv, post_trans.reshape(B * N, 3).contiguous(), combine,
                                                  c2l_trans.reshape(B * N, 3).contiguous(), extra_rots, extra_trans, B, N)

    def geometry_from_camera_matrices(self, post_rot_inv, post_trans, combine, c2l_trans, extra_rots, extra_trans, B, N):
        """bevamd_lss_geometry on prepared per-camera matrices ([B*N,3,3] / [B*N,3], extra_* per sample or None)."""
        lib = _capi.load()
        dev = c2l_trans.device
        frustum = self.frustum.detach().to(dev).float().contiguous()
        D, fH, fW, _ = frustum.shape
        geom = torch.empty((B, N, D, fH, fW, 3), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.bevamd_lss_geometry(_capi.ptr(frustum), D * fH * fW, _capi.ptr(post_rot_inv), _capi.ptr(post_trans),
                                         _capi.ptr(combine), _capi.ptr(c2l_trans), _capi.ptr(extra_rots),
                                         _capi.ptr(extra_trans), B, N, _capi.ptr(geom), _capi.stream_ptr(dev))
        _capi.check(rc, "lss_geometry")
        return geom

    def get_cam_feats(self, x, *args):
        raise NotImplementedError

    # -- the hot path ---------------------------------------------------------------------------------
    def make_plan(self, geom_feats, B):
        """bev_pool precompute for this geometry (base.py:149-169 + bev_pool.py:83-93 + :39-46)."""
        origin = (self.bx - self.dx / 2.0).detach().cpu().tolist()
        return BevPoolPlan.from_geometry(geom_feats.reshape(-1, 3), B, origin, self.dx.detach().cpu().tolist(),
                                         self.nx.detach().cpu().tolist())

    # -- plan cache: a pooling plan is valid for ONE set of calibration + augmentation matrices --------------
    def invalidate_plan(self):
        """Drop the cached pooling plan (new calibration, new device, ...)."""
        self._plan = self._plan_key = self._pending_key = self._plan_geom = None

    @staticmethod
    def _calibration_key(tensors, calibration_id=None):
        """Hashable identity of the matrices a geometry depends on.  A caller-supplied `calibration_id` is used as is
        (zero device work); otherwise the matrices' bytes are fingerprinted — host tensors directly, device tensors through
        ONE small read-back (a few hundred bytes; pass `calibration_id` to stay sync-free)."""
        if calibration_id is not None:
            return ("id", calibration_id)
        flat = torch.cat([t.detach().reshape(-1).float() for t in tensors])
        if flat.is_cuda and not BaseTransform._warned_sync:
            BaseTransform._warned_sync = True
            warnings.warn("cache_geometry=True without calibration_id: the calibration matrices are fingerprinted through a blocking "
                          "device-to-host copy on every forward (and cannot be captured into a HIP graph); pass calibration_id=<any "
                          "hashable that changes with the calibration> to stay sync-free", RuntimeWarning, stacklevel=4)
        return ("bytes", str(flat.device), flat.cpu().numpy().tobytes())

    def _geometry_or_cached(self, c2l_rots, c2l_trans, intrins, post_rots, post_trans, extra_rots, extra_trans,
                            calibration_id=None):
        """forward()'s geometry step: with `cache_geometry` (eval only) and an unchanged calibration the cached plan is still
        valid and NOTHING is recomputed (returns None); otherwise the geometry, with the key its plan will be stored under."""
        self._pending_key = None
        if self.cache_geometry and not self.training:
            B, N, _ = c2l_trans.shape
            D, fH, fW, _ = self.frustum.shape
            key = self._calibration_key((c2l_rots, c2l_trans, intrins, post_rots, post_trans, extra_rots, extra_trans),
                                        calibration_id)
            self._pending_key = (key, B * N * D * fH * fW, B, str(c2l_trans.device))
            if self._plan is not None and self._plan_key == self._pending_key:
                return None
        return self.get_geometry(c2l_rots, c2l_trans, intrins, post_rots, post_trans, extra_rots=extra_rots,
                                 extra_trans=extra_trans)

    def _cached_plan(self, geom_feats, Nprime, B):
        use_cache = self.cache_geometry and not self.training      # train-time augmentation changes the geometry per step
        key, self._pending_key = self._pending_key, None
        if geom_feats is None:                                     # forward() found the calibration unchanged
            assert use_cache and self._plan is not None and self._plan_key == key
            return self._plan
        if use_cache:
            by_tensor = key is None
            if by_tensor:     # bev_pool() called directly: the geometry tensor itself is the identity
                key = (("geom", geom_feats.data_ptr(), geom_feats._version, tuple(geom_feats.shape)), Nprime, B,
                       str(geom_feats.device))
            if self._plan is not None and self._plan_key == key:
                return self._plan
        plan = self.make_plan(geom_feats, B)
        if use_cache:
            self._plan, self._plan_key = plan, key
            # (address, version) only identify the tensor while it is alive: the cache keeps it, so the allocator cannot hand its
            # address to the geometry of another calibration (a recycled address matched the stale plan)
            self._plan_geom = geom_feats if by_tensor else None
        return plan

    def bev_pool(self, geom_feats, x):
        """base.py:141-176: [B,N,D,H,W,3] geometry + [B,N,D,H,W,C] features -> [B, C*nz, nx, ny]."""
        factored = x if isinstance(x, FactoredCamFeats) else None
        if factored is not None:
            if geom_feats is not None:
                (geom_feats,) = _fp32(geom_feats)
            B, N, D, H, W = factored.depth.shape
            C = factored.ctx.shape[2]
        else:
            (x,) = _fp32(x)
            if geom_feats is not None:
                (geom_feats,) = _fp32(geom_feats)
            B, N, D, H, W, C = x.shape
        Nprime = B * N * D * H * W
        on_host = not (factored.depth if factored is not None else x).is_cuda
        if on_host:
            return self._bev_pool_host(geom_feats, factored.materialize() if factored is not None else x)
        plan = self._cached_plan(geom_feats, Nprime, B)
        if factored is not None:
            ctx_cl = factored.ctx.float().permute(0, 1, 3, 4, 2).contiguous()     # [B, N, fH, fW, C] (5 MB)
            out = plan.fused(factored.depth.float().contiguous(), ctx_cl.view(-1, C), D, H, W)   # autograd-aware
        else:
            out = plan.forward(x.reshape(Nprime, C))   # [B, nz, nx, ny, C] fp32
        out = out.permute(0, 4, 1, 2, 3)            # [B, C, nz, nx, ny] view
        # collapse Z (reference: torch.cat(x.unbind(dim=2), 1))
        if out.shape[2] == 1:
            return out[:, :, 0]                     # channels-last view, no copy
        return torch.cat(out.unbind(dim=2), 1)

    def _bev_pool_host(self, geom_feats, x):
        """Host tensors (BASELINE configs[0], a box without a GPU): base.py:141-176 step by step in torch — cell index by
        truncation, batch column, range mask — and `bev_pool()`'s torch QuickCumsum route.  No plan, no cache."""
        if geom_feats is None:
            raise RuntimeError("bev_pool on host tensors needs the geometry (cache_geometry plans are GPU objects)")
        B, N, D, H, W, C = x.shape
        Nprime = B * N * D * H * W
        cell = ((geom_feats - (self.bx - self.dx / 2.0)) / self.dx).long().view(Nprime, 3)
        batch_ix = torch.arange(B, dtype=torch.long).repeat_interleave(Nprime // B).view(Nprime, 1)
        cell = torch.cat((cell, batch_ix), 1)
        inside = ((cell[:, :3] >= 0) & (cell[:, :3] < self.nx.view(1, 3))).all(1)
        nx, ny, nz = (int(v) for v in self.nx)
        out = bev_pool(x.reshape(Nprime, C)[inside], cell[inside], B, nz, nx, ny)   # [B, C, nz, nx, ny]
        return torch.cat(out.unbind(dim=2), 1)

    def _split_mats(self, camera2ego, lidar2ego, camera_intrinsics, camera2lidar, img_aug_matrix, lidar_aug_matrix):
        intrins = camera_intrinsics[..., :3, :3]
        post_rots = img_aug_matrix[..., :3, :3]
        post_trans = img_aug_matrix[..., :3, 3]
        camera2lidar_rots = camera2lidar[..., :3, :3]
        camera2lidar_trans = camera2lidar[..., :3, 3]
        extra_rots = lidar_aug_matrix[..., :3, :3]
        extra_trans = lidar_aug_matrix[..., :3, 3]
        return intrins, post_rots, post_trans, camera2lidar_rots, camera2lidar_trans, extra_rots, extra_trans

    def forward(self, img, points, radar, camera2ego, lidar2ego, lidar2camera, lidar2image, camera_intrinsics,
                camera2lidar, img_aug_matrix, lidar_aug_matrix, **kwargs):
        """base.py:178-235."""
        intrins, post_rots, post_trans, c2l_rots, c2l_trans, extra_rots, extra_trans = self._split_mats(
            camera2ego, lidar2ego, camera_intrinsics, camera2lidar, img_aug_matrix, lidar_aug_matrix)
        geom = self._geometry_or_cached(c2l_rots, c2l_trans, intrins, post_rots, post_trans, extra_rots, extra_trans,
                                        kwargs.get("calibration_id"))
        mats_dict = {"intrin_mats": camera_intrinsics, "ida_mats": img_aug_matrix, "bda_mat": lidar_aug_matrix,
                     "sensor2ego_mats": camera2ego}
        x = self.get_cam_feats(img, mats_dict)
        use_depth = False
        if type(x) == tuple:
            x, depth = x
            use_depth = True
        x = self.bev_pool(geom, x)
        if use_depth:
            return x, depth
        return x


class BaseDepthTransform(BaseTransform):
    def depth_raster(self, img, points, lidar2image, img_aug_matrix, lidar_aug_matrix):
        """base.py:269-329 for depth_input='scalar' without extra features: project every LiDAR point into every
        camera and write its range at the truncated pixel -> [B, N, 1, iH, iW].  One scatter per sample for all
        cameras (the reference loops over cameras with boolean indexing); colliding points are last-writer-wins
        in both."""
        batch_size = len(points)
        n_cam = img.shape[1]
        iH, iW = self.image_size
        if points[0].is_cuda:
            return self._depth_raster_native(points, n_cam, lidar2image, img_aug_matrix, lidar_aug_matrix)
        depth = torch.zeros(batch_size, n_cam, 1, iH, iW, device=points[0].device)
        for b in range(batch_size):
            cur_coords = points[b][:, :3].float()
            cur_img_aug_matrix = img_aug_matrix[b].float()
            cur_lidar_aug_matrix = lidar_aug_matrix[b].float()
            cur_lidar2image = lidar2image[b].float()
            # inverse aug
            cur_coords = cur_coords - cur_lidar_aug_matrix[:3, 3]
            # row-major inverse: MKL's sgemm arithmetic depends on the left operand's layout (tests/golden/make_vtransform_golden.py)
            cur_coords = torch.inverse(cur_lidar_aug_matrix[:3, :3]).contiguous().matmul(cur_coords.transpose(1, 0))
            # lidar2image
            cur_coords = cur_lidar2image[:, :3, :3].matmul(cur_coords)
            cur_coords += cur_lidar2image[:, :3, 3].reshape(-1, 3, 1)
            # get 2d coords
            dist = cur_coords[:, 2, :]
            cur_coords[:, 2, :] = torch.clamp(cur_coords[:, 2, :], 1e-5, 1e5)
            cur_coords[:, :2, :] /= cur_coords[:, 2:3, :]
            # imgaug
            cur_coords = cur_img_aug_matrix[:, :3, :3].matmul(cur_coords)
            cur_coords += cur_img_aug_matrix[:, :3, 3].reshape(-1, 3, 1)
            cur_coords = cur_coords[:, :2, :].transpose(1, 2)
            cur_coords = cur_coords[..., [1, 0]]  # (row, col)
            on_img = ((cur_coords[..., 0] < iH) & (cur_coords[..., 0] >= 0) & (cur_coords[..., 1] < iW)
                      & (cur_coords[..., 1] >= 0))
            pix = cur_coords.long()                                   # truncation (base.py:317)
            cam = torch.arange(n_cam, device=pix.device).view(-1, 1).expand_as(on_img)
            lin = (cam * iH + pix[..., 0]) * iW + pix[..., 1]
            depth[b].view(-1)[lin[on_img]] = dist[on_img]
        return depth

    def _depth_raster_native(self, points, n_cam, lidar2image, img_aug_matrix, lidar_aug_matrix):
        """csrc/vtransform.hip: one (packed atomicMax, unpack) kernel pair for the whole batch, all cameras at once, no host sync; a pixel hit
        by several points keeps the LAST one in input order (the reference's assignment order on CPU; its GPU
        index_put is unordered)."""
        lib = _capi.load()
        iH, iW = self.image_size
        dev = points[0].device
        B = len(points)
        depth = torch.empty((B, n_cam, 1, iH, iW), dtype=torch.float32, device=dev)
        if self.lapack_inverse:
            trans = lidar_aug_matrix[:, :3, 3].float().contiguous()
            inv_rot = torch.inverse(lidar_aug_matrix[:, :3, :3].float()).contiguous()
        else:   # inverse rotation and packed translation column from one launch
            la = lidar_aug_matrix.float().contiguous()                     # [B, 4, 4]
            inv_rot = torch.empty((B, 3, 3), dtype=torch.float32, device=dev)
            trans = torch.empty((B, 3), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _capi.check(lib.bevamd_mat3_inverse_with_column(_capi.ptr(la), 16, 4, B, _capi.ptr(inv_rot), _capi.ptr(trans),
                                                                _capi.stream_ptr(dev)), "mat3_inverse_with_column")
        l2i = lidar2image.float().contiguous()
        ia = img_aug_matrix.float().contiguous()
        pts = [p if (p.dtype == torch.float32 and p.is_contiguous()) else p.float().contiguous() for p in points]
        nfeat = pts[0].shape[1]
        if any(p.shape[1] != nfeat for p in pts):
            raise RuntimeError("depth_raster: every sample must have the same number of point features")
        ptrs = (ctypes.c_void_p * B)(*[p.data_ptr() for p in pts])
        counts = (ctypes.c_int * B)(*[int(p.shape[0]) for p in pts])
        with torch.cuda.device(dev):
            wsb = lib.bevamd_depth_raster_workspace_bytes(n_cam, iH, iW) * B
            # one zero-initialised (winner, depth) map per device and size, left zero by every raster (the unpack pass clears what
            # it reads): no fill launch per call — the rasters of a device are issued on one stream at a time (eager calls and
            # the replays of a captured graph included).  A first use under graph capture takes a fresh map and the fill.
            # Keyed by the issuing stream as well: rasters on two streams — an eager call beside a graph replay, two
            # models — never share a map; a call that fails between scatter and unpack drops its map (it may hold stale words).
            key = (dev.index, int(wsb), int(torch.cuda.current_stream(dev).cuda_stream))
            ws = _RASTER_MAPS.get(key)
            if ws is None and not torch.cuda.is_current_stream_capturing():
                while len(_RASTER_MAPS) >= _RASTER_MAPS_MAX:      # bounded (69 MB per map at 8 frames, one per stream ever used):
                    _RASTER_MAPS.pop(next(iter(_RASTER_MAPS)))    # the oldest goes; a stream that comes back takes a fresh, zeroed map
                ws = _RASTER_MAPS[key] = torch.zeros(wsb, dtype=torch.uint8, device=dev)
            if ws is not None:
                rc = lib.bevamd_depth_raster_batch_zero_ws(ptrs, counts, B, nfeat, _capi.ptr(inv_rot), _capi.ptr(trans), 3,
                                                           _capi.ptr(l2i), _capi.ptr(ia), n_cam, iH, iW, _capi.ptr(depth),
                                                           _capi.ptr(ws), wsb, _capi.stream_ptr(dev))
                if rc != 0:
                    _RASTER_MAPS.pop(key, None)
            else:
                ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
                rc = lib.bevamd_depth_raster_batch(ptrs, counts, B, nfeat, _capi.ptr(inv_rot), _capi.ptr(trans), 3, _capi.ptr(l2i),
                                                   _capi.ptr(ia), n_cam, iH, iW, _capi.ptr(depth), _capi.ptr(ws), wsb,
                                                   _capi.stream_ptr(dev))
        _capi.check(rc, "depth_raster_batch")
        return depth

    def forward(self, img, points, radar, sensor2ego, lidar2ego, lidar2camera, lidar2image, cam_intrinsic,
                camera2lidar, img_aug_matrix, lidar_aug_matrix, metas=None, **kwargs):
        """base.py:240-361."""
        intrins, post_rots, post_trans, c2l_rots, c2l_trans, extra_rots, extra_trans = self._split_mats(
            sensor2ego, lidar2ego, cam_intrinsic, camera2lidar, img_aug_matrix, lidar_aug_matrix)
        if self.use_points == "radar":
            points = radar
        if self.height_expand or self.add_depth_features or self.depth_input != "scalar":
            raise NotImplementedError("only scalar 1-channel depth input is implemented (what DepthLSSTransform's "
                                      "Conv2d(1, 8, 1) stem and the released checkpoint use; SURVEY.md D3)")
        depth = self.depth_raster(img, points, lidar2image, img_aug_matrix, lidar_aug_matrix)
        geom = self._geometry_or_cached(c2l_rots, c2l_trans, intrins, post_rots, post_trans, extra_rots, extra_trans,
                                        kwargs.get("calibration_id"))
        mats_dict = {"intrin_mats": intrins, "ida_mats": img_aug_matrix, "bda_mat": lidar_aug_matrix,
                     "sensor2ego_mats": sensor2ego}
        x = self.get_cam_feats(img, depth, mats_dict)
        use_depth = False
        if type(x) == tuple:
            x, depth = x
            use_depth = True
        x = self.bev_pool(geom, x)
        if use_depth:
            return x, depth
        return x


def _downsample_block(out_channels, downsample):
    if downsample > 1:
        assert downsample == 2, downsample
        return nn.Sequential(
            nn.Conv2d(out_channels, out_channels, 3, padding=1, bias=False), nn.BatchNorm2d(out_channels), nn.ReLU(True),
            nn.Conv2d(out_channels, out_channels, 3, stride=downsample, padding=1, bias=False),
            nn.BatchNorm2d(out_channels), nn.ReLU(True),
            nn.Conv2d(out_channels, out_channels, 3, padding=1, bias=False), nn.BatchNorm2d(out_channels), nn.ReLU(True))
    return nn.Identity()


class LSSTransform(BaseTransform):
    """lss.py:13-86."""

    def __init__(self, in_channels, out_channels, image_size, feature_size, xbound, ybound, zbound, dbound,
                 downsample: int = 1) -> None:
        super().__init__(in_channels=in_channels, out_channels=out_channels, image_size=image_size,
                         feature_size=feature_size, xbound=xbound, ybound=ybound, zbound=zbound, dbound=dbound)
        self.depthnet = nn.Conv2d(in_channels, self.D + self.C, 1)
        self.downsample = _downsample_block(out_channels, downsample)

    def get_cam_feats(self, x, *args):
        (x,) = _fp32(x)
        B, N, C, fH, fW = x.shape
        x = x.view(B * N, C, fH, fW)
        x = self.depthnet(x)
        depth = x[:, : self.D].softmax(dim=1)
        if self.fused_cam_feats:
            return FactoredCamFeats(depth.view(B, N, self.D, fH, fW), x[:, self.D: (self.D + self.C)].view(B, N, self.C, fH, fW))
        x = depth.unsqueeze(1) * x[:, self.D: (self.D + self.C)].unsqueeze(2)
        x = x.view(B, N, self.C, self.D, fH, fW)
        x = x.permute(0, 1, 3, 4, 5, 2)
        return x

    def forward(self, *args, **kwargs):
        x = super().forward(*args, **kwargs)
        x = self.downsample(x)
        return x


class DepthLSSTransform(BaseDepthTransform):
    """depth_lss.py:14-102."""

    def __init__(self, in_channels, out_channels, image_size, feature_size, xbound, ybound, zbound, dbound,
                 downsample: int = 1) -> None:
        super().__init__(in_channels=in_channels, out_channels=out_channels, image_size=image_size,
                         feature_size=feature_size, xbound=xbound, ybound=ybound, zbound=zbound, dbound=dbound,
                         depth_input="scalar", height_expand=False, add_depth_features=False)
        self.dtransform = nn.Sequential(
            nn.Conv2d(1, 8, 1), nn.BatchNorm2d(8), nn.ReLU(True),
            nn.Conv2d(8, 32, 5, stride=4, padding=2), nn.BatchNorm2d(32), nn.ReLU(True),
            nn.Conv2d(32, 64, 5, stride=2, padding=2), nn.BatchNorm2d(64), nn.ReLU(True))
        self.depthnet = nn.Sequential(
            nn.Conv2d(in_channels + 64, in_channels, 3, padding=1), nn.BatchNorm2d(in_channels), nn.ReLU(True),
            nn.Conv2d(in_channels, in_channels, 3, padding=1), nn.BatchNorm2d(in_channels), nn.ReLU(True),
            nn.Conv2d(in_channels, self.D + self.C, 1))
        self.downsample = _downsample_block(out_channels, downsample)

    def get_cam_feats(self, x, d, *args):
        x, d = _fp32(x, d)
        B, N, C, fH, fW = x.shape
        d = d.view(B * N, *d.shape[2:])
        x = x.view(B * N, C, fH, fW)
        d = self.dtransform(d)
        x = torch.cat([d, x], dim=1)
        x = self.depthnet(x)
        depth = x[:, : self.D].softmax(dim=1)
        if self.fused_cam_feats:
            return FactoredCamFeats(depth.view(B, N, self.D, fH, fW), x[:, self.D: (self.D + self.C)].view(B, N, self.C, fH, fW))
        x = depth.unsqueeze(1) * x[:, self.D: (self.D + self.C)].unsqueeze(2)
        x = x.view(B, N, self.C, self.D, fH, fW)
        x = x.permute(0, 1, 3, 4, 5, 2)
        return x

    def forward(self, *args, **kwargs):
        x = super().forward(*args, **kwargs)
        x = self.downsample(x)
        return x


register_everywhere("vtransform", LSSTransform)
register_everywhere("vtransform", DepthLSSTransform)
