"""Rotated-box BEV IoU and NMS — host-side mirror of `mmdet3d/ops/iou3d/iou3d_utils.py` and of the pybind module
`iou3d_cuda` (src/iou3d.cpp:182-210) over the HIP C ABI (csrc/iou3d.hip).

Same function names, arguments and return values: `boxes_iou_bev(boxes_a, boxes_b)`, `nms_gpu(boxes, scores, thresh,
pre_maxsize, post_max_size)`, `nms_normal_gpu(boxes, scores, thresh)`; boxes are [N, 5] = (x1, y1, x2, y2, ry).
What differs underneath: the suppression mask never leaves the GPU and the greedy sweep runs there too, so an NMS call
costs one 4-byte read-back (the count the API returns a correctly sized tensor from) instead of a mask-sized copy, a
host loop and a malloc/free pair."""
import ctypes

import torch

from . import _capi


def _check(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor: the HIP extension has no CPU path")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    if t.dim() != 2 or t.shape[1] != 5:
        raise RuntimeError(f"{name} must have shape [N, 5] (x1, y1, x2, y2, ry), got {tuple(t.shape)}")
    return t.contiguous()


def _pairwise(fn_name, boxes_a, boxes_b, out=None):
    lib = _capi.load()
    a, b = _check(boxes_a, "boxes_a"), _check(boxes_b, "boxes_b")
    if out is None:
        out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        rc = getattr(lib, fn_name)(_capi.ptr(a), a.shape[0], _capi.ptr(b), b.shape[0], _capi.ptr(out),
                                   _capi.stream_ptr(a.device))
    _capi.check(rc, fn_name)
    return out


def boxes_overlap_bev(boxes_a, boxes_b):
    """[M, N] overlap areas of rotated boxes (iou3d_kernel.cu:126-222)."""
    return _pairwise("bevamd_iou3d_boxes_overlap_bev", boxes_a, boxes_b)


def boxes_iou_bev(boxes_a, boxes_b):
    """iou3d_utils.py:6-20: [M, N] IoU of rotated boxes in the bird view."""
    return _pairwise("bevamd_iou3d_boxes_iou_bev", boxes_a, boxes_b)


def nms_sorted(boxes, thresh, normal=False, sync=True):
    """NMS over boxes already sorted by descending score.  sync=True -> kept indices [K] int64 (one 4-byte read-back);
    sync=False -> (keep [N] int64 with the first `count` entries valid, count int32 device tensor)."""
    lib = _capi.load()
    boxes = _check(boxes, "boxes")
    n = boxes.shape[0]
    dev = boxes.device
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    host = ctypes.c_int(0)
    with torch.cuda.device(dev):
        wsb = lib.bevamd_iou3d_nms_workspace_bytes(n)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
        rc = lib.bevamd_iou3d_nms(_capi.ptr(boxes), n, float(thresh), int(bool(normal)), _capi.ptr(keep), _capi.ptr(count),
                                  ctypes.byref(host) if sync else None, _capi.ptr(ws), wsb, _capi.stream_ptr(dev))
    _capi.check(rc, "iou3d_nms")
    if not sync:
        return keep[:n], count
    return keep[: int(host.value)]


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, post_max_size=None):
    """iou3d_utils.py:23-48: rotated NMS; returns indices into `boxes` in descending score order."""
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep = order[nms_sorted(boxes[order], thresh, normal=False)].contiguous()
    if post_max_size is not None:
        keep = keep[:post_max_size]
    return keep


def nms_normal_gpu(boxes, scores, thresh):
    """iou3d_utils.py:51-68: axis-aligned NMS (the angle column is ignored)."""
    order = scores.sort(0, descending=True)[1]
    return order[nms_sorted(boxes[order], thresh, normal=True)].contiguous()


class _Iou3dCuda:
    """Drop-in for the pybind module `iou3d_cuda` (iou3d.cpp:182-210): outputs are written into caller tensors, the NMS
    entry points fill a (CPU) `keep` tensor and return the number kept."""

    @staticmethod
    def boxes_overlap_bev_gpu(boxes_a, boxes_b, ans_overlap):
        _pairwise("bevamd_iou3d_boxes_overlap_bev", boxes_a, boxes_b, out=ans_overlap)
        return 1

    @staticmethod
    def boxes_iou_bev_gpu(boxes_a, boxes_b, ans_iou):
        _pairwise("bevamd_iou3d_boxes_iou_bev", boxes_a, boxes_b, out=ans_iou)
        return 1

    @staticmethod
    def _nms(boxes, keep, thresh, normal):
        kept = nms_sorted(boxes, thresh, normal=normal)
        keep[: kept.shape[0]] = kept.to(keep.device)
        return int(kept.shape[0])

    @staticmethod
    def nms_gpu(boxes, keep, nms_overlap_thresh, device_id):
        return _Iou3dCuda._nms(boxes, keep, nms_overlap_thresh, False)

    @staticmethod
    def nms_normal_gpu(boxes, keep, nms_overlap_thresh, device_id):
        return _Iou3dCuda._nms(boxes, keep, nms_overlap_thresh, True)


iou3d_cuda = _Iou3dCuda()
