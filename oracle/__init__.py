"""ORACLE — test infrastructure only (CPU restatement of the reference's hot path).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package.  Nothing under `bevfusion_amd/` imports it; the product path has no CPU fallback.

`oracle/*.c` is plain C (gcc); `build()` compiles it into `oracle/_build/liboracle.so`.
The numpy wrappers below cite the reference lines they follow.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_LIB = os.path.join(_BUILD, "liboracle.so")
_lib = None


def _sources():
    return sorted(os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith(".c"))


def build(force=False):
    os.makedirs(_BUILD, exist_ok=True)
    srcs = _sources()
    if not force and os.path.exists(_LIB) and all(os.path.getmtime(s) <= os.path.getmtime(_LIB) for s in srcs):
        return _LIB
    cmd = ["gcc", "-O2", "-std=c11", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall",
           "-o", _LIB] + srcs + ["-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    return _LIB


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _i64(v):
    return ctypes.c_int64(int(v))


# --------------------------------------------------------------------------------------------
# bev_pool
# --------------------------------------------------------------------------------------------
def bev_pool_ranks(coords, B, D, H, W):
    """bev_pool.py:86-91.  coords [N,4] (x,y,z,b) -> int64 ranks."""
    coords = np.ascontiguousarray(coords, dtype=np.int64)
    ranks = np.empty(coords.shape[0], dtype=np.int64)
    lib().oracle_bev_pool_ranks(_p(coords), _i64(coords.shape[0]), _i64(B), _i64(D), _i64(H), _i64(W), _p(ranks))
    return ranks


def bev_cell_index(geom, batch, origin, dx, nx):
    """vtransforms/base.py:149-169: truncated cell index + batch index + range mask.
    geom [N',3] fp32 batch-major; origin = (bx - dx/2) as fp32; returns (coords int64 [N',4], kept bool)."""
    geom = np.ascontiguousarray(geom, dtype=np.float32)
    n = geom.shape[0]
    origin = np.ascontiguousarray(origin, dtype=np.float32)
    dx = np.ascontiguousarray(dx, dtype=np.float32)
    nx = np.ascontiguousarray(nx, dtype=np.int64)
    coords = np.empty((n, 4), dtype=np.int64)
    kept = np.empty(n, dtype=np.uint8)
    lib().oracle_bev_cell_index(_p(geom), _i64(n), _i64(n // batch), _p(origin), _p(dx), _p(nx), _p(coords), _p(kept))
    return coords, kept.astype(bool)


def bev_pool_intervals(ranks_sorted):
    """bev_pool.py:39-46 -> (interval_starts int32, interval_lengths int32)."""
    ranks_sorted = np.ascontiguousarray(ranks_sorted, dtype=np.int64)
    n = ranks_sorted.shape[0]
    starts = np.empty(max(n, 1), dtype=np.int32)
    lengths = np.empty(max(n, 1), dtype=np.int32)
    lib().oracle_bev_pool_intervals.restype = ctypes.c_int64
    k = lib().oracle_bev_pool_intervals(_p(ranks_sorted), _i64(n), _p(starts), _p(lengths))
    return starts[:k].copy(), lengths[:k].copy()


def bev_pool_forward_sorted(x, geom, starts, lengths, B, D, H, W, dtype=np.float64):
    """bev_pool_cuda.cu:20-42 on sorted rows -> out [B,D,H,W,C] (float64 or float32-sequential)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    geom = np.ascontiguousarray(geom, dtype=np.int32)
    starts = np.ascontiguousarray(starts, dtype=np.int32)
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    c = x.shape[1] if x.ndim == 2 else 0
    out = np.empty((B, D, H, W, c), dtype=dtype)
    fn = lib().oracle_bev_pool_forward_f64 if dtype == np.float64 else lib().oracle_bev_pool_forward_f32
    fn(_p(x), _p(geom), _p(starts), _p(lengths), _i64(starts.shape[0]), _i64(c), _i64(B), _i64(D), _i64(H), _i64(W),
       _p(out))
    return out


def bev_pool_backward_sorted(out_grad, geom, starts, lengths, n, B, D, H, W):
    """bev_pool_cuda.cu:61-84 -> x_grad [n, C] fp32 (rows in sorted order)."""
    out_grad = np.ascontiguousarray(out_grad, dtype=np.float32)
    geom = np.ascontiguousarray(geom, dtype=np.int32)
    starts = np.ascontiguousarray(starts, dtype=np.int32)
    lengths = np.ascontiguousarray(lengths, dtype=np.int32)
    c = out_grad.shape[-1]
    x_grad = np.empty((n, c), dtype=np.float32)
    lib().oracle_bev_pool_backward(_p(out_grad), _p(geom), _p(starts), _p(lengths), _i64(starts.shape[0]), _i64(n),
                                   _i64(c), _i64(B), _i64(D), _i64(H), _i64(W), _p(x_grad))
    return x_grad


def bev_pool_prologue(coords, B, D, H, W):
    """bev_pool.py:86-93 + :39-46: ranks, STABLE argsort (ties in input order — the reference's
    argsort leaves tie order unspecified), sorted ranks/coords, interval arrays."""
    coords = np.ascontiguousarray(coords, dtype=np.int64)
    ranks = bev_pool_ranks(coords, B, D, H, W)
    order = np.argsort(ranks, kind="stable")
    ranks_sorted = ranks[order]
    geom_sorted = coords[order].astype(np.int32)
    starts, lengths = bev_pool_intervals(ranks_sorted)
    return dict(ranks=ranks, order=order, ranks_sorted=ranks_sorted, geom_sorted=geom_sorted,
                interval_starts=starts, interval_lengths=lengths)


def bev_pool(feats, coords, B, D, H, W, dtype=np.float64):
    """The whole op, bev_pool.py:83-97: -> [B, C, D, H, W]."""
    pro = bev_pool_prologue(coords, B, D, H, W)
    x = np.ascontiguousarray(feats, dtype=np.float32)[pro["order"]]
    out = bev_pool_forward_sorted(x, pro["geom_sorted"], pro["interval_starts"], pro["interval_lengths"], B, D, H, W,
                                  dtype=dtype)
    return np.ascontiguousarray(out.transpose(0, 4, 1, 2, 3))


# --------------------------------------------------------------------------------------------
# hard voxelization
# --------------------------------------------------------------------------------------------
def dynamic_voxelize(points, voxel_size, coors_range):
    """voxelization_cpu.cpp:8-44 -> coors [N,3] int32 (x,y,z) or -1."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    n, f = points.shape
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    cr = np.ascontiguousarray(coors_range, dtype=np.float32)
    coors = np.empty((n, 3), dtype=np.int32)
    lib().oracle_dynamic_voxelize(_p(points), _i64(n), _i64(f), _p(vs), _p(cr), _p(coors))
    return coors


def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """voxelization_cpu.cpp:46-101 (+ voxelize.py:52-71 allocation/slicing) ->
    (voxels [M,max_points,F], coors [M,3] int32, num_points_per_voxel [M] int32)."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    n, f = points.shape
    vs = np.ascontiguousarray(voxel_size, dtype=np.float32)
    cr = np.ascontiguousarray(coors_range, dtype=np.float32)
    voxels = np.zeros((max_voxels, max_points, f), dtype=np.float32)
    coors = np.zeros((max_voxels, 3), dtype=np.int32)
    npv = np.zeros((max_voxels,), dtype=np.int32)
    fn = lib().oracle_hard_voxelize
    fn.restype = ctypes.c_int32
    m = fn(_p(points), _i64(n), _i64(f), _p(vs), _p(cr), ctypes.c_int32(max_points), ctypes.c_int32(max_voxels),
           _p(voxels), _p(coors), _p(npv))
    return voxels[:m].copy(), coors[:m].copy(), npv[:m].copy()


def voxel_mean(voxels, num_points_per_voxel):
    """bevfusion.py:192-195: voxels.sum(1) / count (fp32, slot order)."""
    voxels = np.ascontiguousarray(voxels, dtype=np.float32)
    npv = np.ascontiguousarray(num_points_per_voxel, dtype=np.int32)
    m, mp, f = voxels.shape
    feats = np.empty((m, f), dtype=np.float32)
    lib().oracle_voxel_mean(_p(voxels), _p(npv), _i64(m), _i64(mp), _i64(f), _p(feats))
    return feats


def voxelize_batch(points_list, voxel_size, coors_range, max_points, max_voxels):
    """bevfusion.py:169-197: per-sample hard voxelize, batch index prepended to coords, mean reduce."""
    feats, coords, sizes = [], [], []
    for k, pts in enumerate(points_list):
        v, c, n = hard_voxelize(pts, voxel_size, coors_range, max_points, max_voxels)
        feats.append(voxel_mean(v, n))
        coords.append(np.concatenate([np.full((c.shape[0], 1), k, np.int32), c], 1))
        sizes.append(n)
    return np.concatenate(feats), np.concatenate(coords), np.concatenate(sizes)
