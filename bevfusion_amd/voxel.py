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
import contextlib
import ctypes
import os

import torch
from torch import nn
from torch.autograd import Function
from torch.nn.modules.utils import _pair

from . import _capi

__all__ = ["Voxelization", "voxelization", "voxel_layer", "voxelize_batch", "DynamicScatter", "dynamic_scatter"]

_REDUCE = {"sum": 0, "mean": 1, "max": 2}   # reduce_t, scatter_points_cuda.cu:7 / voxelization.h:83-92


_VOXEL_LANES = int(os.environ.get("BEVAMD_VOXEL_STREAMS", "4"))   # concurrent per-sample voxelizations (1 = serial)
_lanes = {}


def _voxel_lanes(dev, n):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), n)
    if key not in _lanes:
        _lanes[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
    return _lanes[key]


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


    @staticmethod
    def dynamic_point_to_voxel_forward(feats, coors, reduce_type):
        """voxelization.h:108-120 -> scatter_points_cuda.cu:197-250.  feats [N, C] fp32, coors [N, ndim] int ->
        [reduced_feats [M, C], out_coors [M, ndim] (ascending lexicographic order, dtype of `coors`),
        coors_map [N] int32 (-1 for rows with a negative coordinate), reduce_count [M] int32]."""
        if reduce_type not in _REDUCE:
            raise RuntimeError(f"do not support reduce type {reduce_type}")       # voxelization.h:91
        if not feats.is_cuda or not coors.is_cuda:
            raise RuntimeError("do not support cpu yet")                          # voxelization.h:118
        if feats.dtype != torch.float32:
            raise RuntimeError(f"feats must be float32 (got {feats.dtype})")
        if not feats.is_contiguous() or not coors.is_contiguous():
            raise RuntimeError("feats and coors must be contiguous")             # CHECK_INPUT, :201-202
        n, c = feats.shape
        if n == 0:                                                               # :207-211
            return [feats.clone().detach(), coors.clone().detach(), coors.new_empty((0,), dtype=torch.int32),
                    coors.new_empty((0,), dtype=torch.int32)]
        ndim = coors.shape[1]
        c32 = coors if coors.dtype == torch.int32 else coors.to(torch.int32)
        dev = feats.device
        lib = _capi.load()
        import ctypes

        out_coors = torch.empty((n, ndim), dtype=torch.int32, device=dev)
        coors_map = torch.empty((n,), dtype=torch.int32, device=dev)
        count = torch.empty((n,), dtype=torch.int32, device=dev)
        order = torch.empty((n,), dtype=torch.int32, device=dev)
        seg = torch.empty((n + 1,), dtype=torch.int32, device=dev)
        m_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        host = ctypes.c_int(0)
        with torch.cuda.device(dev):
            wsb = lib.bevamd_dynamic_scatter_workspace_bytes(n)
            ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            rc = lib.bevamd_dynamic_scatter_index(_capi.ptr(c32), n, ndim, _capi.ptr(out_coors), _capi.ptr(coors_map),
                                                  _capi.ptr(count), _capi.ptr(order), _capi.ptr(seg), _capi.ptr(m_dev),
                                                  ctypes.byref(host), _capi.ptr(ws), wsb, _capi.stream_ptr(dev))
            _capi.check(rc, "dynamic_scatter_index")
            m = int(host.value)
            reduced = torch.empty((m, c), dtype=torch.float32, device=dev)
            rc = lib.bevamd_dynamic_scatter_reduce(_capi.ptr(feats), c, _capi.ptr(order), _capi.ptr(seg), m,
                                                   _REDUCE[reduce_type], _capi.ptr(reduced), _capi.stream_ptr(dev))
            _capi.check(rc, "dynamic_scatter_reduce")
        out_coors = out_coors[:m]
        if coors.dtype != torch.int32:
            out_coors = out_coors.to(coors.dtype)
        return [reduced, out_coors, coors_map, count[:m]]

    @staticmethod
    def dynamic_point_to_voxel_backward(grad_feats, grad_reduced_feats, feats, reduced_feats, coors_idx, reduce_count,
                                        reduce_type):
        """voxelization.h:122-140 -> scatter_points_cuda.cu:252-330: fills `grad_feats` [N, C] in place."""
        if reduce_type not in _REDUCE:
            raise RuntimeError(f"do not support reduce type {reduce_type}")
        if not grad_feats.is_cuda:
            raise RuntimeError("do not support cpu yet")
        for t, nm in ((grad_feats, "grad_feats"), (grad_reduced_feats, "grad_reduced_feats"), (feats, "feats"),
                      (reduced_feats, "reduced_feats"), (coors_idx, "coors_idx"), (reduce_count, "reduce_count")):
            if not t.is_contiguous():
                raise RuntimeError(f"{nm} must be contiguous")
        if grad_feats.dtype != torch.float32 or grad_reduced_feats.dtype != torch.float32:
            raise RuntimeError("gradients must be float32")
        n, c = feats.shape
        m = reduced_feats.shape[0]
        dev = grad_feats.device
        lib = _capi.load()
        mode = _REDUCE[reduce_type]
        with torch.cuda.device(dev):
            scratch = torch.empty((max(m * c, 1),), dtype=torch.int32, device=dev) if mode == 2 else None
            rc = lib.bevamd_dynamic_scatter_backward(
                _capi.ptr(grad_feats), _capi.ptr(grad_reduced_feats), _capi.ptr(feats), _capi.ptr(reduced_feats),
                _capi.ptr(coors_idx.int() if coors_idx.dtype != torch.int32 else coors_idx),
                _capi.ptr(reduce_count.int() if reduce_count.dtype != torch.int32 else reduce_count), n, m, c, mode,
                _capi.ptr(scratch), _capi.stream_ptr(dev))
        _capi.check(rc, "dynamic_scatter_backward")


voxel_layer = _VoxelLayer()
hard_voxelize = voxel_layer.hard_voxelize
dynamic_voxelize = voxel_layer.dynamic_voxelize
dynamic_point_to_voxel_forward = voxel_layer.dynamic_point_to_voxel_forward
dynamic_point_to_voxel_backward = voxel_layer.dynamic_point_to_voxel_backward


class _dynamic_scatter(Function):
    """scatter_points.py:8-47: (feats [N, C], coors [N, ndim], reduce_type) -> (voxel_feats [M, C], voxel_coors [M, ndim])."""

    @staticmethod
    def forward(ctx, feats, coors, reduce_type="max"):
        voxel_feats, voxel_coors, point2voxel_map, voxel_points_count = dynamic_point_to_voxel_forward(feats, coors,
                                                                                                       reduce_type)
        ctx.reduce_type = reduce_type
        ctx.save_for_backward(feats, voxel_feats, point2voxel_map, voxel_points_count)
        ctx.mark_non_differentiable(voxel_coors)
        return voxel_feats, voxel_coors

    @staticmethod
    def backward(ctx, grad_voxel_feats, grad_voxel_coors=None):
        feats, voxel_feats, point2voxel_map, voxel_points_count = ctx.saved_tensors
        grad_feats = torch.empty_like(feats)       # every element is written by the call below
        dynamic_point_to_voxel_backward(grad_feats, grad_voxel_feats.contiguous(), feats, voxel_feats, point2voxel_map,
                                        voxel_points_count, ctx.reduce_type)
        return grad_feats, None, None


dynamic_scatter = _dynamic_scatter.apply


class DynamicScatter(nn.Module):
    """scatter_points.py:53-108: mean (average_points) or max reduction of the points of every voxel, per sample when
    the coordinates carry a batch column."""

    def __init__(self, voxel_size, point_cloud_range, average_points: bool):
        super().__init__()
        self.voxel_size = voxel_size
        self.point_cloud_range = point_cloud_range
        self.average_points = average_points

    def forward_single(self, points, coors):
        return dynamic_scatter(points.contiguous(), coors.contiguous(), "mean" if self.average_points else "max")

    def forward(self, points, coors):
        if coors.size(-1) == 3:
            return self.forward_single(points, coors)
        batch_size = int(coors[-1, 0]) + 1          # scatter_points.py:89: samples are concatenated in order
        voxels, voxel_coors = [], []
        for i in range(batch_size):
            inds = torch.where(coors[:, 0] == i)
            voxel, voxel_coor = self.forward_single(points[inds], coors[inds][:, 1:])
            voxel_coors.append(nn.functional.pad(voxel_coor, (1, 0), mode="constant", value=i))
            voxels.append(voxel)
        return torch.cat(voxels, dim=0), torch.cat(voxel_coors, dim=0)

    def __repr__(self):
        return (f"{self.__class__.__name__}(voxel_size={self.voxel_size}, point_cloud_range={self.point_cloud_range}, "
                f"average_points={self.average_points})")


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


_VOXEL_BATCHED = os.environ.get("BEVAMD_VOXEL_BATCHED", "1") != "0"   # one segmented sort per batch (0 = one per sample)
_VOXEL_MAX_BATCH = 64


_ORDERS = {"first": 0, "appearance": 0, None: 0, "key": 1, "linear": 1}


def _voxelize_mean_batch(points_list, voxel_size, point_cloud_range, max_num_points, max_voxels, packed, order="first",
                         rows16_dtype=None):
    """All samples in the launches of one (`bevamd_voxelize_mean_batch_ex`): returns (feats [B*cap, F], coords [B*cap, 4],
    sizes [B*cap], counts [B], total [1]); rows packed sample after sample when `packed`, else sample b at row b*cap.
    order: "first" = first-appearance rows (the reference's numbering), "key" = the same rows in ascending linear cell index."""
    if order not in _ORDERS:
        raise ValueError(f"voxelize: order must be 'first' or 'key', got {order!r}")
    lib = _capi.load()
    B = len(points_list)
    dev = points_list[0].device
    F = points_list[0].shape[1]
    points_list = [_check_points(p) for p in points_list]
    for p in points_list:
        if p.shape[1] != F or p.device != dev:
            raise ValueError("voxelize_batch: samples must share the device and the feature count")
    feats = torch.empty((B * max_voxels, F), dtype=torch.float32, device=dev)
    coords = torch.empty((B * max_voxels, 4), dtype=torch.int32, device=dev)
    sizes = torch.empty((B * max_voxels,), dtype=torch.int32, device=dev)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.int32, device=dev)
    ptrs = (ctypes.c_void_p * B)(*[p.data_ptr() if p.shape[0] else None for p in points_list])
    nums = (ctypes.c_int * B)(*[int(p.shape[0]) for p in points_list])
    with torch.cuda.device(dev):
        wsb = lib.bevamd_voxelize_mean_batch_workspace_bytes(nums, B)
        ws = torch.empty(max(int(wsb), 1), dtype=torch.uint8, device=dev)
        rows16 = None
        if rows16_dtype is not None:
            if F != 5 or rows16_dtype not in (torch.float16, torch.bfloat16):
                raise ValueError("voxelize: 16-bit encoder rows are written for 5 point features, fp16 or bf16")
            rows16 = torch.empty((B * max_voxels, 8), dtype=rows16_dtype, device=dev)
        rc = lib.bevamd_voxelize_mean_batch_rows16(ptrs, nums, B, F, _capi.floats(voxel_size), _capi.floats(point_cloud_range),
                                                   int(max_num_points), int(max_voxels), 1 if packed else 0, _ORDERS[order],
                                                   _capi.ptr(feats), _capi.ptr(coords), _capi.ptr(sizes), _capi.ptr(counts),
                                                   _capi.ptr(total), _capi.ptr(rows16),
                                                   0 if rows16 is None else (1 if rows16_dtype == torch.float16 else 2),
                                                   8 if rows16 is not None else 0, _capi.ptr(ws), wsb, _capi.stream_ptr(dev))
    _capi.check(rc, "voxelize_mean_batch")
    if rows16 is not None:
        return feats, coords, sizes, counts, total, rows16
    return feats, coords, sizes, counts, total


@torch.no_grad()
def voxelize_batch(points_list, voxel_size, point_cloud_range, max_num_points, max_voxels, sync=True, order="first"):
    """`BEVFusion.voxelize` (bevfusion.py:169-197, hard voxelization + voxelize_reduce) for a batch.

    points_list: list of [N_k, F] fp32 GPU tensors.  Returns (feats [M, F] = per-voxel mean,
    coords [M, 4] int32 = (batch_idx, x, y, z), sizes [M] int32).
    With sync=True the outputs are exactly sized (one D2H copy of all counts for the whole batch);
    with sync=False they are (padded buffers, counts_dev) for callers that stay on the device.
    order="key": each sample's rows in ascending linear cell index instead of first appearance (same set, same values)."""
    lib = _capi.load()
    B = len(points_list)
    dev = points_list[0].device
    F = points_list[0].shape[1]
    if _VOXEL_BATCHED and B <= _VOXEL_MAX_BATCH:
        feats, coords, sizes, counts, _ = _voxelize_mean_batch(points_list, voxel_size, point_cloud_range, max_num_points,
                                                               max_voxels, packed=False, order=order)
        feats, coords, sizes = feats.view(B, max_voxels, F), coords.view(B, max_voxels, 4), sizes.view(B, max_voxels)
    else:
        if _ORDERS.get(order, -1) != 0:
            raise ValueError("voxelize_batch: order='key' needs the batched entry (<= 64 samples, BEVAMD_VOXEL_BATCHED=1)")
        feats, coords, sizes, counts = _voxelize_mean_lanes(points_list, voxel_size, point_cloud_range, max_num_points,
                                                            max_voxels)
    if not sync:
        return feats, coords, sizes, counts

    cnt = counts.tolist()  # the single host sync of the batch
    if any(c < 0 for c in cnt):   # a single-pass kernel's bounded spin expired (csrc/single_pass.h): flagged, never silent
        raise RuntimeError("voxelize_batch: the single-pass sort / scan kernels reported a stalled predecessor tile; "
                           "results are invalid (set BEVAMD_SINGLE_PASS=0 for the multi-launch kernels)")
    feats = torch.cat([feats[k, : cnt[k]] for k in range(B)], 0)
    coords = torch.cat([coords[k, : cnt[k]] for k in range(B)], 0)
    sizes = torch.cat([sizes[k, : cnt[k]] for k in range(B)], 0)
    return feats, coords, sizes


def _voxelize_mean_lanes(points_list, voxel_size, point_cloud_range, max_num_points, max_voxels):
    """One `bevamd_voxelize_mean` per sample, spread over a few HIP streams (batches beyond the batched entry's 64
    samples, or BEVAMD_VOXEL_BATCHED=0): padded slabs + device counts."""
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
    points_list = [_check_points(p) for p in points_list]
    with torch.cuda.device(dev):
        wsb = lib.bevamd_hard_voxelize_workspace_bytes(nmax)
        # The samples are independent and each one's ~20 kernels (300 k points) fill a fraction of the GPU: spread them
        # over a few HIP streams (fork from / join into the caller's stream, so the call stays graph-capturable).
        lanes = _voxel_lanes(dev, min(B, _VOXEL_LANES)) if B > 1 and _VOXEL_LANES > 1 else [None]
        ws = [torch.empty(wsb, dtype=torch.uint8, device=dev) for _ in lanes]
        main = torch.cuda.current_stream(dev)
        for lane in lanes:
            if lane is not None:
                lane.wait_stream(main)
        try:
            for k, pts in enumerate(points_list):
                lane = lanes[k % len(lanes)]
                with torch.cuda.stream(lane) if lane is not None else contextlib.nullcontext():
                    rc = lib.bevamd_voxelize_mean(_capi.ptr(pts), _capi.ptr(feats[k]), _capi.ptr(coords[k]), _capi.ptr(sizes[k]),
                                                  vs, cr, int(max_num_points), int(max_voxels), pts.shape[0], F, k,
                                                  _capi.ptr(counts[k:]), _capi.ptr(ws[k % len(lanes)]), wsb,
                                                  _capi.stream_ptr(dev))
                _capi.check(rc, "voxelize_mean")
        finally:
            # always join the lanes back (also when a launch raised): the buffers above were allocated on `main`, and the
            # caching allocator may hand them out again as soon as `main` moves on
            for lane in lanes:
                if lane is not None:
                    main.wait_stream(lane)
    return feats, coords, sizes, counts


@torch.no_grad()
def voxelize_batch_device(points_list, voxel_size, point_cloud_range, max_num_points, max_voxels, order="first", encoder_rows=None):
    """`voxelize_batch` that never touches the host: returns (feats [B*max_voxels, F], coords [B*max_voxels, 4],
    sizes [B*max_voxels], total [1] int32 on the device) with the batch packed sample after sample in the first `total`
    rows — what `SparseEncoder(..., num_voxels=total)` consumes on its sync-free path.  order="key": rows in ascending
    linear index over the whole packed batch (pass `coors_order="linear"` to the encoder: level 1 then runs on the
    staged-rows kernels with sorted-key neighbour search).  encoder_rows=torch.float16 | torch.bfloat16 (5 point features): the first
    element of the result is instead the [B*max_voxels, 8] 16-bit zero-padded rows the SparseEncoder's first convolution reads —
    the fp32 means rounded once, exactly what the encoder's own pad-and-cast pass would produce — written by the mean kernel
    itself; pass it to the encoder as `voxel_features`."""
    lib = _capi.load()
    if _VOXEL_BATCHED and len(points_list) <= _VOXEL_MAX_BATCH:
        if encoder_rows is not None:
            _, oc, osz, _, total, rows16 = _voxelize_mean_batch(points_list, voxel_size, point_cloud_range, max_num_points,
                                                                max_voxels, packed=True, order=order, rows16_dtype=encoder_rows)
            return rows16, oc, osz, total
        of, oc, osz, _, total = _voxelize_mean_batch(points_list, voxel_size, point_cloud_range, max_num_points, max_voxels,
                                                     packed=True, order=order)
        return of, oc, osz, total
    if encoder_rows is not None:
        raise ValueError("voxelize_batch_device: encoder_rows needs the batched entry (<= 64 samples, BEVAMD_VOXEL_BATCHED=1)")
    if _ORDERS.get(order, -1) != 0:
        raise ValueError("voxelize_batch_device: order='key' needs the batched entry (<= 64 samples, BEVAMD_VOXEL_BATCHED=1)")
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
