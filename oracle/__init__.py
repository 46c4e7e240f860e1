"""ORACLE — test infrastructure only (CPU restatement of the reference's hot path).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package.  Nothing under `bevfusion_amd/` imports it; the product path has no CPU fallback.

`oracle/*.c` is plain C (gcc); `build()` compiles it into `oracle/_build/liboracle.so`.
The numpy wrappers below cite the reference lines they follow.

Parity pins (tests/golden/, each with the script that generated it from the reference's own code):
  bev_pool            bev_pool_ref_small.npz   reference kernel (hipified, run on an MI355X)     test_oracle_bev_pool.py
  hard voxelization   voxel_ref_{a,b,c}.npz    reference hard_voxelize_cpu                       test_oracle_voxel.py
  spconv (encoder)    spconv_ref_*.npz         reference CPU functors                            test_oracle_spconv.py
  spconv (ext rest)   spconv_ext_*.npz         reference CPU functors (transposed, dilated, 2D,
                                               inverse conv, max pooling)                        test_oracle_spconv_ext.py
  dynamic scatter     scatter_ref.npz          reference GPU kernels (hipified, run on an MI355X) test_oracle_scatter.py
  iou3d               iou3d_ref.npz            reference GPU kernels (hipified, run on an MI355X) test_oracle_iou3d.py
  view-transform glue vtransform_ref.npz       the reference's own Python function bodies (base.py) exec'd on CPU torch
                                               under mmcv stubs (tests/golden/make_vtransform_golden.py)
                                                                                                 test_oracle_vtransform.py
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


def dynamic_scatter(feats, coors, reduce_type, dtype=np.float64):
    """voxel_layer.dynamic_point_to_voxel_forward (scatter_points_cuda.cu:197-250) in numpy: rows of `coors` with a
    negative entry are dropped (:213); the distinct rows in ascending lexicographic order are the voxels
    (at::unique_dim(sorted), :215-222); features reduce by sum / mean (= sum / count, :237-238) / max.
    Returns (reduced [M, C] `dtype`, out_coors [M, ndim], coors_map [N] int32 with -1 for dropped rows,
    reduce_count [M] int32).  Sums accumulate in `dtype` in ascending point order (the reference's atomicAdd order is
    unspecified, so sum / mean carry a tolerance; max is exact)."""
    feats = np.asarray(feats, dtype=np.float32)
    coors = np.asarray(coors)
    n, c = feats.shape
    valid = ~(coors < 0).any(1)
    coors_map = np.full(n, -1, dtype=np.int32)
    if not valid.any():
        return (np.zeros((0, c), dtype), coors[:0].copy(), coors_map, np.zeros(0, np.int32))
    out_coors, inv, count = np.unique(coors[valid], axis=0, return_inverse=True, return_counts=True)
    inv = inv.reshape(-1)
    coors_map[valid] = inv.astype(np.int32)
    m = out_coors.shape[0]
    f = feats[valid].astype(dtype)
    if reduce_type == "max":
        red = np.full((m, c), -np.inf, dtype)
        np.maximum.at(red, inv, f)
    elif reduce_type in ("sum", "mean"):
        red = np.zeros((m, c), dtype)
        np.add.at(red, inv, f)                       # unbuffered, in index order
        if reduce_type == "mean":
            red = red / count[:, None].astype(dtype)
    else:
        raise ValueError(reduce_type)
    return red, out_coors, coors_map, count.astype(np.int32)


def dynamic_scatter_backward(grad_reduced, feats, reduced, coors_map, reduce_count, reduce_type):
    """voxel_layer.dynamic_point_to_voxel_backward (scatter_points_cuda.cu:252-330) -> grad_feats [N, C] float32."""
    feats = np.asarray(feats, dtype=np.float32)
    g = np.asarray(grad_reduced, dtype=np.float32)
    n, c = feats.shape
    out = np.zeros((n, c), np.float32)
    keep = coors_map >= 0
    if reduce_type == "sum":
        out[keep] = g[coors_map[keep]]
    elif reduce_type == "mean":
        out[keep] = g[coors_map[keep]] / reduce_count[coors_map[keep]][:, None].astype(np.float32)
    elif reduce_type == "max":
        red = np.asarray(reduced, dtype=np.float32)
        first = np.full(red.shape, n, dtype=np.int64)                    # lowest point id attaining the maximum (:144-170)
        for p in np.nonzero(keep)[0][::-1]:
            hit = feats[p] == red[coors_map[p]]
            first[coors_map[p]][hit] = p
        vv, cc = np.nonzero(first < n)
        out[first[vv, cc], cc] = g[vv, cc]
    else:
        raise ValueError(reduce_type)
    return out


# --------------------------------------------------------------------------------------------
# spconv: rulebook + sparse convolution
# --------------------------------------------------------------------------------------------
def _i32arr(v):
    return np.ascontiguousarray(v, dtype=np.int32)


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    """spconv/ops.py:20-31."""
    return [(input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
            for i in range(len(input_size))]


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    """spconv/ops.py:34-42."""
    return [(input_size[i] - 1) * stride[i] - 2 * padding[i] + kernel_size[i] + output_padding[i]
            for i in range(len(input_size))]


def get_indice_pairs(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm, order="cuda",
                     transpose=False, out_padding=None):
    """spconv_ops.h:27-141 (getIndicePair<NDim>) on CPU semantics, NDim = 2 or 3 (a 2D problem is run as the 3D one with
    a unit last axis — same offsets, same row order — and the padding column is dropped again).
    indices [N,1+NDim] int32 (b, spatial...).  Returns (out_indices [M,1+NDim], indice_pairs [K,2,N] (-1 padded),
    indice_num [K], out_shape).  transpose: the rulebook of a transposed convolution (geometry.h:196-245).
    order="cpu":  output rows of a strided conv in first-appearance order (geometry.h:181-187);
    order="cuda": rows renumbered by ascending linear index b*vol + x*Y*Z + y*Z + z, which is what the
                  CUDA path produces (torch::_unique at spconv_ops.h:130, indice.cu.h:112-145).  D8."""
    indices = _i32arr(indices)
    ndim = indices.shape[1] - 1
    if ndim == 2:
        lifted = np.concatenate([indices, np.zeros((indices.shape[0], 1), np.int32)], 1)
        op = list(out_padding) + [0] if out_padding is not None else None
        oi, pairs, num, oshape = get_indice_pairs(lifted, batch_size, list(spatial_shape) + [1], list(ksize) + [1],
                                                  list(stride) + [1], list(padding) + [0], list(dilation) + [1], subm,
                                                  order, transpose, op)
        return np.ascontiguousarray(oi[:, :3]), pairs, num, oshape[:2]
    n = indices.shape[0]
    ks, st, pd, dl = _i32arr(ksize), _i32arr(stride), _i32arr(padding), _i32arr(dilation)
    K = int(np.prod(ks))
    pairs = np.full((K, 2, max(n, 1)), -1, dtype=np.int32)
    num = np.zeros(K, dtype=np.int32)
    if subm:
        shape = _i32arr(spatial_shape)
        lib().oracle_subm_indice_pairs(_p(indices), _i64(n), _p(shape), _p(ks), _p(dl), _p(pairs), _p(num))
        return indices.copy(), pairs[:, :, :n] if n else pairs[:, :, :0], num, list(spatial_shape)
    if transpose:
        out_shape = get_deconv_output_size(list(spatial_shape), list(ks), list(st), list(pd), list(dl),
                                           list(out_padding) if out_padding is not None else [0, 0, 0])
    else:
        out_shape = get_conv_output_size(list(spatial_shape), list(ks), list(st), list(pd), list(dl))
    oshape = _i32arr(out_shape)
    out_inds = np.zeros((max(n * K, 1), 4), dtype=np.int32)
    fn = lib().oracle_deconv_indice_pairs if transpose else lib().oracle_conv_indice_pairs
    fn.restype = ctypes.c_int64
    m = fn(_p(indices), _i64(n), _p(ks), _p(st), _p(pd), _p(dl), _p(oshape), _p(out_inds), _p(pairs), _p(num))
    out_inds = out_inds[:m].copy()
    pairs = pairs[:, :, :n] if n else pairs[:, :, :0]
    if order == "cuda" and m > 0:
        lin = ((out_inds[:, 0].astype(np.int64) * out_shape[0] + out_inds[:, 1]) * out_shape[1] + out_inds[:, 2]) \
            * out_shape[2] + out_inds[:, 3]
        perm = np.argsort(lin, kind="stable")          # new row r holds old row perm[r]
        inv = np.empty(m, dtype=np.int32)
        inv[perm] = np.arange(m, dtype=np.int32)
        out_inds = out_inds[perm]
        pairs = pairs.copy()
        sel = pairs[:, 1, :] >= 0
        pairs[:, 1, :][sel] = inv[pairs[:, 1, :][sel]]
    return out_inds, pairs, num, out_shape


def _subm_shortcut(pairs, num, n_rows):
    """spconv_ops.h:272-276,300-303,309: with subM set, the offset holding the most pairs (first maximum) is taken to be
    the identity map and runs as one dense GEMM over all rows instead of through its pairs.  That IS the identity for an
    undilated odd (or even) kernel; for a dilated SubM rulebook no offset is, and the reference still does it — the
    restatement follows the reference."""
    pairs, num = pairs.copy(), num.copy()
    kmax = int(np.argmax(num))                      # first maximum, like std::max_element
    if pairs.shape[2] < n_rows:
        pairs = np.concatenate([pairs, np.full((pairs.shape[0], 2, n_rows - pairs.shape[2]), -1, np.int32)], 2)
    pairs[kmax, :, :] = -1
    pairs[kmax, :, :n_rows] = np.arange(n_rows, dtype=np.int32)
    num[kmax] = n_rows
    return np.ascontiguousarray(pairs), num


def indice_conv(features, filters, indice_pairs, indice_num, num_act_out, inverse=False, subm=False):
    """spconv_ops.h:260-361, float64 accumulation.  filters [kx,ky,kz,Cin,Cout]. -> [M, Cout] float64."""
    features = np.ascontiguousarray(features, dtype=np.float32)
    filters = np.ascontiguousarray(filters, dtype=np.float32)
    pairs = _i32arr(indice_pairs)
    num = _i32arr(indice_num)
    if subm:
        pairs, num = _subm_shortcut(pairs, num, num_act_out)
    K, _, L = pairs.shape
    cin, cout = filters.shape[-2], filters.shape[-1]
    out = np.empty((num_act_out, cout), dtype=np.float64)
    lib().oracle_indice_conv_f64(_p(features), _p(filters), _p(pairs), _p(num), _i64(K), _i64(L), _i64(num_act_out),
                                 _i64(cin), _i64(cout), ctypes.c_int32(int(inverse)), _p(out))
    return out


def indice_conv_backward(features, filters, out_grad, indice_pairs, indice_num, inverse=False, subm=False):
    """spconv_ops.h:363-456, float64 accumulation -> (in_grad [N,Cin], filter_grad like filters)."""
    features = np.ascontiguousarray(features, dtype=np.float32)
    filters = np.ascontiguousarray(filters, dtype=np.float32)
    out_grad = np.ascontiguousarray(out_grad, dtype=np.float32)
    pairs = _i32arr(indice_pairs)
    num = _i32arr(indice_num)
    if subm:
        pairs, num = _subm_shortcut(pairs, num, features.shape[0])
    K, _, L = pairs.shape
    cin, cout = filters.shape[-2], filters.shape[-1]
    gi = np.empty((features.shape[0], cin), dtype=np.float64)
    gw = np.empty((K, cin, cout), dtype=np.float64)
    lib().oracle_indice_conv_backward_f64(_p(features), _p(filters), _p(out_grad), _p(pairs), _p(num), _i64(K), _i64(L),
                                          _i64(features.shape[0]), _i64(cin), _i64(cout), ctypes.c_int32(int(inverse)),
                                          _p(gi), _p(gw))
    return gi, gw.reshape(filters.shape)


def indice_maxpool(features, indice_pairs, indice_num, num_act_out):
    """pool_ops.h:25-58 with the CPU functor's arithmetic (maxpool_cpu.cc:22-40), fp32 -> [M, C] float32."""
    features = np.ascontiguousarray(features, dtype=np.float32)
    pairs, num = _i32arr(indice_pairs), _i32arr(indice_num)
    K, _, L = pairs.shape
    out = np.empty((num_act_out, features.shape[1]), dtype=np.float32)
    lib().oracle_indice_maxpool_f32(_p(features), _p(pairs), _p(num), _i64(K), _i64(L), _i64(num_act_out),
                                    _i64(features.shape[1]), _p(out))
    return out


def indice_maxpool_backward(features, out_features, out_grad, indice_pairs, indice_num):
    """pool_ops.h:60-97 / maxpool_cpu.cc:43-66, fp32 -> [N, C] float32."""
    features = np.ascontiguousarray(features, dtype=np.float32)
    out_features = np.ascontiguousarray(out_features, dtype=np.float32)
    out_grad = np.ascontiguousarray(out_grad, dtype=np.float32)
    pairs, num = _i32arr(indice_pairs), _i32arr(indice_num)
    K, _, L = pairs.shape
    gi = np.empty_like(features)
    lib().oracle_indice_maxpool_backward_f32(_p(features), _p(out_features), _p(out_grad), _p(pairs), _p(num), _i64(K),
                                             _i64(L), _i64(features.shape[0]), _i64(features.shape[1]), _p(gi))
    return gi


def pairs_as_sets(indice_pairs, indice_num):
    """Canonical form of a rulebook: per offset, the sorted list of (in,out) pairs — the CUDA path
    fills each offset's list in atomicAdd order (indice.cu.h:62,195), so only the SET is defined."""
    out = []
    for k in range(indice_pairs.shape[0]):
        m = int(indice_num[k])
        pr = np.stack([indice_pairs[k, 0, :m], indice_pairs[k, 1, :m]], 1)
        out.append(pr[np.lexsort((pr[:, 0], pr[:, 1]))] if m else pr.reshape(0, 2))
    return out


# --------------------------------------------------------------------------------------------
# view-transform glue (mmdet3d/models/vtransforms/base.py — Python in the reference; restated in C, vtransform_oracle.c,
# with the rounding of the torch CPU kernels those lines dispatch to; pinned by tests/golden/vtransform_ref.npz)
# --------------------------------------------------------------------------------------------
def mat3_inverse(m):
    """Stand-in for `torch.inverse` on [..., 3, 3] fp32 matrices when the caller has no reference-computed inverse: float64
    LAPACK rounded to fp32.  NOT bit-identical to the reference's fp32 LAPACK call (MKL, un-vendored): the golden fixture
    carries the reference's own inverses for the bit-exact checks."""
    return np.linalg.inv(np.asarray(m, np.float64)).astype(np.float32)


def lss_geometry(frustum, post_rots, post_trans, camera2lidar_rots, camera2lidar_trans, intrins, extra_rots=None,
                 extra_trans=None, inv_post_rots=None, combine=None):
    """BaseTransform.get_geometry (base.py:92-135).  frustum [D,fH,fW,3]; per-camera matrices [B,N,3,3] / [B,N,3];
    extra_rots [B,3,3], extra_trans [B,3] -> [B, N, D, fH, fW, 3] float32.
    `inv_post_rots` = torch.inverse(post_rots), `combine` = camera2lidar_rots.matmul(torch.inverse(intrins)) as the reference
    computed them (fixture); when absent they come from `mat3_inverse` / a float64 product."""
    f32 = np.float32
    frustum = np.ascontiguousarray(frustum, f32)
    B, N = np.asarray(camera2lidar_trans).shape[:2]
    if inv_post_rots is None:
        inv_post_rots = mat3_inverse(post_rots)
    if combine is None:
        combine = (np.asarray(camera2lidar_rots, np.float64) @ np.linalg.inv(np.asarray(intrins, np.float64))).astype(f32)
    c = lambda a, shp: np.ascontiguousarray(np.asarray(a, f32).reshape(shp))  # noqa: E731
    ipr, pt = c(inv_post_rots, (B * N, 9)), c(post_trans, (B * N, 3))
    cmb, ct = c(combine, (B * N, 9)), c(camera2lidar_trans, (B * N, 3))
    er = c(extra_rots, (B, 9)) if extra_rots is not None else None
    et = c(extra_trans, (B, 3)) if extra_trans is not None else None
    npts = frustum.size // 3
    out = np.empty((B, N) + frustum.shape, f32)
    lib().oracle_lss_geometry(_p(frustum), _i64(npts), _p(ipr), _p(pt), _p(cmb), _p(ct), _p(er) if er is not None else None,
                              _p(et) if et is not None else None, _i64(B), _i64(N), _p(out))
    return out


def depth_raster(points, lidar2image, img_aug_matrix, lidar_aug_matrix, image_size, inv_lidar_aug_rot=None):
    """One sample of BaseDepthTransform.forward's raster (base.py:283-329): points [n, >=3], lidar2image / img_aug_matrix
    [N,4,4], lidar_aug_matrix [4,4] -> (depth [N, 1, iH, iW] float32, winner [N, iH, iW] int32 = index of the point
    written at each pixel or -1).  Colliding points: the last one in input order wins (sequential assignment).
    `inv_lidar_aug_rot` = torch.inverse(lidar_aug_matrix[:3,:3]) as the reference computed it (fixture), else `mat3_inverse`."""
    f32 = np.float32
    iH, iW = image_size
    pts = np.ascontiguousarray(points, f32)
    lam = np.asarray(lidar_aug_matrix, f32)
    l2i = np.ascontiguousarray(lidar2image, f32)
    ia = np.ascontiguousarray(img_aug_matrix, f32)
    inv = np.ascontiguousarray(mat3_inverse(lam[:3, :3]) if inv_lidar_aug_rot is None else inv_lidar_aug_rot, f32)
    trans = np.ascontiguousarray(lam[:3, 3])
    n_cam = l2i.shape[0]
    depth = np.empty((n_cam, 1, iH, iW), f32)
    winner = np.empty((n_cam, iH, iW), np.int32)
    lib().oracle_depth_raster(_p(pts), _i64(pts.shape[0]), _i64(pts.shape[1] if pts.ndim == 2 else 3), _p(inv), _p(trans),
                              _p(l2i), _p(ia), _i64(n_cam), _i64(iH), _i64(iW), _p(depth), _p(winner))
    return depth, winner


# --------------------------------------------------------------------------------------------
# iou3d (mmdet3d/ops/iou3d): rotated BEV overlap / IoU / NMS
# --------------------------------------------------------------------------------------------
def iou3d_pairwise(boxes_a, boxes_b, mode):
    """mode 'overlap' | 'iou' | 'iou_normal' -> [M, N] float32 (iou3d_kernel.cu:126-229, 291-299)."""
    a = np.ascontiguousarray(boxes_a, dtype=np.float32).reshape(-1, 5)
    b = np.ascontiguousarray(boxes_b, dtype=np.float32).reshape(-1, 5)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().iou3d_pairwise(_p(a), _i64(a.shape[0]), _p(b), _i64(b.shape[0]),
                         ctypes.c_int({"overlap": 0, "iou": 1, "iou_normal": 2}[mode]), _p(out))
    return out


def iou3d_nms(boxes_sorted, thresh, normal=False):
    """Greedy NMS over boxes sorted by descending score (iou3d.cpp:96-180) -> kept indices int64."""
    b = np.ascontiguousarray(boxes_sorted, dtype=np.float32).reshape(-1, 5)
    keep = np.zeros(max(b.shape[0], 1), np.int64)
    fn = lib().iou3d_nms
    fn.restype = ctypes.c_int64
    k = fn(_p(b), _i64(b.shape[0]), ctypes.c_float(thresh), ctypes.c_int(int(normal)), _p(keep))
    return keep[: int(k)].copy()


def rotated_overlap_float64(box_a, box_b):
    """Independent check of the overlap area: Sutherland-Hodgman clipping of the two rotated rectangles in float64
    (not the reference's algorithm; agrees with it away from degenerate contacts)."""
    def corners(bx):
        x1, y1, x2, y2, ang = [float(v) for v in bx]
        cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
        c, s = np.cos(ang), np.sin(ang)
        pts = [(x1, y1), (x2, y1), (x2, y2), (x1, y2)]
        # rotate_around_center (iou3d_kernel.cu:109-118): x' = dx*c + dy*s, y' = -dx*s + dy*c
        return [((px - cx) * c + (py - cy) * s + cx, -(px - cx) * s + (py - cy) * c + cy) for px, py in pts]

    def area(poly):
        return 0.5 * sum(poly[i][0] * poly[(i + 1) % len(poly)][1] - poly[(i + 1) % len(poly)][0] * poly[i][1]
                         for i in range(len(poly)))

    subj, clip = corners(box_a), corners(box_b)
    if area(clip) < 0:
        clip = clip[::-1]
    for i in range(4):
        p, q = clip[i], clip[(i + 1) % 4]
        if not subj:
            break
        inside = lambda r: (q[0] - p[0]) * (r[1] - p[1]) - (q[1] - p[1]) * (r[0] - p[0]) >= 0  # noqa: E731
        out = []
        for j in range(len(subj)):
            cur, prev = subj[j], subj[j - 1]
            if inside(cur) != inside(prev):
                d1 = (q[0] - p[0], q[1] - p[1])
                d2 = (cur[0] - prev[0], cur[1] - prev[1])
                den = d1[0] * d2[1] - d1[1] * d2[0]
                t = ((prev[0] - p[0]) * d2[1] - (prev[1] - p[1]) * d2[0]) / den
                out.append((p[0] + t * d1[0], p[1] + t * d1[1]))
            if inside(cur):
                out.append(cur)
        subj = out
    return abs(area(subj)) if len(subj) >= 3 else 0.0
