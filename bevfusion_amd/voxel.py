"""Voxelization — host-side mirror of `mmdet3d/ops/voxel/voxelize.py` over the HIP C ABI.

Reference interface reproduced (same names, ctor kwargs, attributes, return shapes):
  * `voxel_layer.hard_voxelize / dynamic_voxelize`   (voxel/src/voxelization.cpp:6-11)
  * `_Voxelization`, `voxelization`                  (voxelize.py:10-74)
  * `Voxelization(voxel_size, point_cloud_range, max_num_points, max_voxels=20000, deterministic=True)`
    with `.grid_size` / `.pcd_shape`                 (voxelize.py:77-138)

MI355X-native addition: `voxelize_batch()` = `BEVFusion.voxelize` (bevfusion.py:169-197) for a whole
batch with the mean-reduce fused, no `[max_voxels, max_points, F]` intermediate and one host sync
for the whole batch (the reference syncs once per sample).
"""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn.modules.utils import _pair

from . import _capi

__all__ = ["Voxelization", "voxelization", "voxel_layer", "voxelize_batch"]


def _check_points(points):
    if not points.is_cuda:
        raise RuntimeError("points must be a GPU tensor: the HIP extension has no CPU path")
    if points.dtype != torch.float32:
        # voxelization_cuda.cu:339-347 hard-codes float; bevfusion.py:170 forces fp32 before the call
        raise RuntimeError(f"points must be float32 (got {points.dtype})")
    if points.dim() != 2 or points.shape[1] < 3:
        raise RuntimeError("points must be [N, >=3]")
    return points.contiguous()


class _VoxelLayer:
    """Drop-in for the reference's pybind module `voxel_layer`."""

    @staticmethod
    def hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range, max_points, max_voxels,
                      NDim=3, deterministic=True):
        """Fills the caller's buffers in place and returns voxel_num as a host int (blocking), like
        voxelization_cuda.cu:369-372."""
        points = _check_points(points)
        lib = _capi.load()
        n, f = points.shape
        for t, nm in ((voxels, "voxels"), (coors, "coors"), (num_points_per_voxel, "num_points_per_voxel")):
            if not t.is_contiguous():
                raise RuntimeError(f"{nm} must be contiguous")
        if coors.dtype != torch.int32 or num_points_per_voxel.dtype != torch.int32 or voxels.dtype != torch.float32:
            raise RuntimeError("voxels must be float32, coors / num_points_per_voxel int32")
        import ctypes

        count_dev = torch.zeros(1, dtype=torch.int32, device=points.device)
        host = ctypes.c_int(0)
        with torch.cuda.device(points.device):
            wsb = lib.bevamd_hard_voxelize_workspace_bytes(n)
            ws = torch.empty(wsb, dtype=torch.uint8, device=points.device)
            rc = lib.bevamd_hard_voxelize(
                _capi.ptr(points), _capi.ptr(voxels), _capi.ptr(coors), _capi.ptr(num_points_per_voxel),
                _capi.floats(voxel_size), _capi.floats(coors_range), int(max_points), int(max_voxels), n, f, int(NDim),
                int(bool(deterministic)), _capi.ptr(count_dev), ctypes.byref(host), _capi.ptr(ws), wsb,
                _capi.stream_ptr(points.device))
        _capi.check(rc, "hard_voxelize")
        return int(host.value)

    @staticmethod
    def dynamic_voxelize(points, coors, voxel_size, coors_range, NDim=3):
        points = _check_points(points)
        lib = _capi.load()
        n, f = points.shape
        with torch.cuda.device(points.device):
            rc = lib.bevamd_dynamic_voxelize(_capi.ptr(points), _capi.ptr(coors), _capi.floats(voxel_size),
                                             _capi.floats(coors_range), n, f, int(NDim), _capi.stream_ptr(points.device))
        _capi.check(rc, "dynamic_voxelize")


voxel_layer = _VoxelLayer()
hard_voxelize = voxel_layer.hard_voxelize
dynamic_voxelize = voxel_layer.dynamic_voxelize


class _Voxelization(Function):
    @staticmethod
    def forward(ctx, points, voxel_size, coors_range, max_points=35, max_voxels=20000, deterministic=True):
        """Same contract as voxelize.py:12-71.  Returns (voxels [M,max_points,F], coors [M,3] int32 (x,y,z),
        num_points_per_voxel [M] int32), or per-point coors when max_points == -1 or max_voxels == -1."""
        if max_points == -1 or max_voxels == -1:
            coors = points.new_zeros(size=(points.size(0), 3), dtype=torch.int)
            dynamic_voxelize(points, coors, voxel_size, coors_range, 3)
            return coors
        # the kernel writes every row it creates in full -> no zero fill needed (reference: new_zeros)
        voxels = points.new_empty(size=(max_voxels, max_points, points.size(1)))
        coors = points.new_empty(size=(max_voxels, 3), dtype=torch.int)
        num_points_per_voxel = points.new_empty(size=(max_voxels,), dtype=torch.int)
        voxel_num = hard_voxelize(points, voxels, coors, num_points_per_voxel, voxel_size, coors_range, max_points,
                                  max_voxels, 3, deterministic)
        return voxels[:voxel_num], coors[:voxel_num], num_points_per_voxel[:voxel_num]


voxelization = _Voxelization.apply


class Voxelization(nn.Module):
    def __init__(self, voxel_size, point_cloud_range, max_num_points, max_voxels=20000, deterministic=True):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.max_num_points = max_num_points
        if isinstance(max_voxels, tuple):
            self.max_voxels = max_voxels
        elif isinstance(max_voxels, list):
            self.max_voxels = tuple(max_voxels)
        else:
            self.max_voxels = _pair(max_voxels)
        self.deterministic = deterministic

        point_cloud_range = torch.tensor(point_cloud_range, dtype=torch.float32)
        voxel_size = torch.tensor(voxel_size, dtype=torch.float32)
        grid_size = (point_cloud_range[3:] - point_cloud_range[:3]) / voxel_size
        grid_size = torch.round(grid_size).long()
        input_feat_shape = grid_size[:2]
        self.grid_size = grid_size
        self.pcd_shape = [*input_feat_shape, 1]

    def forward(self, input):
        max_voxels = self.max_voxels[0] if self.training else self.max_voxels[1]
        return voxelization(input, self.voxel_size, self.point_cloud_range, self.max_num_points, max_voxels,
                            self.deterministic)

    def __repr__(self):
        tmpstr = self.__class__.__name__ + "("
        tmpstr += "voxel_size=" + str(self.voxel_size)
        tmpstr += ", point_cloud_range=" + str(self.point_cloud_range)
        tmpstr += ", max_num_points=" + str(self.max_num_points)
        tmpstr += ", max_voxels=" + str(self.max_voxels)
        tmpstr += ", deterministic=" + str(self.deterministic)
        tmpstr += ")"
        return tmpstr


@torch.no_grad()
def voxelize_batch(points_list, voxel_size, point_cloud_range, max_num_points, max_voxels, sync=True):
    """`BEVFusion.voxelize` (bevfusion.py:169-197, hard voxelization + voxelize_reduce) for a batch.

    points_list: list of [N_k, F] fp32 GPU tensors.  Returns (feats [M, F] = per-voxel mean,
    coords [M, 4] int32 = (batch_idx, x, y, z), sizes [M] int32).
    With sync=True the outputs are exactly sized (one D2H copy of all counts for the whole batch);
    with sync=False they are (padded buffers, counts_dev) for callers that stay on the device."""
    lib = _capi.load()
    B = len(points_list)
    dev = points_list[0].device
    F = points_list[0].shape[1]
    feats = torch.empty((B, max_voxels, F), dtype=torch.float32, device=dev)
    coords = torch.empty((B, max_voxels, 4), dtype=torch.int32, device=dev)
    sizes = torch.empty((B, max_voxels), dtype=torch.int32, device=dev)
    counts = torch.zeros(B, dtype=torch.int32, device=dev)
    vs, cr = _capi.floats(voxel_size), _capi.floats(point_cloud_range)
    nmax = max(int(p.shape[0]) for p in points_list)
    with torch.cuda.device(dev):
        wsb = lib.bevamd_hard_voxelize_workspace_bytes(nmax)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        for k, pts in enumerate(points_list):
            pts = _check_points(pts)
            rc = lib.bevamd_voxelize_mean(_capi.ptr(pts), _capi.ptr(feats[k]), _capi.ptr(coords[k]), _capi.ptr(sizes[k]),
                                          vs, cr, int(max_num_points), int(max_voxels), pts.shape[0], F, k,
                                          _capi.ptr(counts[k:]), _capi.ptr(ws), wsb, _capi.stream_ptr(dev))
            _capi.check(rc, "voxelize_mean")
    if not sync:
        return feats, coords, sizes, counts

    cnt = counts.tolist()  # the single host sync of the batch
    feats = torch.cat([feats[k, : cnt[k]] for k in range(B)], 0)
    coords = torch.cat([coords[k, : cnt[k]] for k in range(B)], 0)
    sizes = torch.cat([sizes[k, : cnt[k]] for k in range(B)], 0)
    return feats, coords, sizes


@torch.no_grad()
def voxelize_batch_device(points_list, voxel_size, point_cloud_range, max_num_points, max_voxels):
    """`voxelize_batch` that never touches the host: returns (feats [B*max_voxels, F], coords [B*max_voxels, 4],
    sizes [B*max_voxels], total [1] int32 on the device) with the batch packed sample after sample in the first `total`
    rows — what `SparseEncoder(..., num_voxels=total)` consumes on its sync-free path."""
    lib = _capi.load()
    feats, coords, sizes, counts = voxelize_batch(points_list, voxel_size, point_cloud_range, max_num_points, max_voxels,
                                                  sync=False)
    B, cap, F = feats.shape
    if B == 1:
        return feats[0], coords[0], sizes[0], counts
    dev = feats.device
    of = torch.empty((B * cap, F), dtype=torch.float32, device=dev)
    oc = torch.empty((B * cap, 4), dtype=torch.int32, device=dev)
    osz = torch.empty((B * cap,), dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.bevamd_voxel_compact(_capi.ptr(feats), _capi.ptr(coords), _capi.ptr(sizes), _capi.ptr(counts), B, cap, F,
                                      _capi.ptr(of), _capi.ptr(oc), _capi.ptr(osz), _capi.ptr(total), _capi.stream_ptr(dev))
    _capi.check(rc, "voxel_compact")
    return of, oc, osz, total
