// Dynamic scatter: reduce point features into the voxels they fall in (no point / voxel caps), gfx950.
//
// Replaces (reference, /root/reference/mmdet3d/ops/voxel/src/scatter_points_cuda.cu):
//   :197-250  dynamic_point_to_voxel_forward_gpu   coors with a negative entry are dropped; at::unique_dim(sorted,
//             inverse, counts) numbers the voxels in ascending lexicographic order; feats_reduce_kernel (:88-111)
//             reduces with float atomics (sum / mean = sum / count / max)
//   :252-330  dynamic_point_to_voxel_backward_gpu  sum: g[voxel]; mean: g[voxel] / count; max: the lowest-numbered
//             point that attains the maximum takes the gradient (atomicMin over point indices, :144-170)
//
// Native formulation (integer work + one pass over the features; nothing here is MFMA- or HBM-peak bound):
//   * the lexicographic order is one stable LSD radix sort of (packed row key, point id): columns are packed into
//     32-bit keys from the last column backwards using only the bits the data needs (column maxima are read back once),
//     e.g. (1440, 1440, 41) voxel grids sort on 29 bits in a single 4-pass sort; rows that do not fit 31 bits fall back to
//     one stable sort per column group;
//   * heads of equal-key runs + a prefix sum give voxel ids, coordinates, the point->voxel map, and segment starts;
//   * the reduction walks each voxel's segment of the SORTED order: no float atomics, fixed (ascending point id)
//     summation order, bit-reproducible — the reference's atomicAdd order is not;
//   * the backward works from (coors_map, reduce_count) alone, which is all the reference interface hands over.
#include "common.h"

#include <math.h>

namespace bevamd {

constexpr int DS_MAX_DIM = 4;

struct DsCols {
  int ndim;
  int first, last;         // columns [first, last] packed into this key, `last` in the low bits
  int shift[DS_MAX_DIM];   // left shift of column c inside the key
  uint32_t invalid_key;    // key of rows with a negative coordinate: above every valid key
};

__device__ __forceinline__ bool ds_row_valid(const int* __restrict__ row, int ndim) {
  bool ok = true;
  for (int c = 0; c < ndim; ++c) ok = ok && row[c] >= 0;
  return ok;
}

// column maxima over the valid rows + number of invalid rows
__global__ __launch_bounds__(256) void ds_colmax_kernel(const int* __restrict__ coors, int n, int ndim,
                                                        int* __restrict__ stats /*[DS_MAX_DIM + 1]*/) {
  int mx[DS_MAX_DIM] = {0, 0, 0, 0};
  int bad = 0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int* row = coors + (size_t)i * ndim;
    if (ds_row_valid(row, ndim)) {
      for (int c = 0; c < ndim; ++c) mx[c] = max(mx[c], row[c]);
    } else {
      ++bad;
    }
  }
  for (int c = 0; c < DS_MAX_DIM; ++c) {
    int v = mx[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_down(v, o, 64));
    if ((threadIdx.x & 63) == 0 && v > 0) atomicMax(&stats[c], v);
  }
  bad = wave_reduce_add(bad);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(&stats[DS_MAX_DIM], bad);
}

// keys of one column group, read through the current order (nullptr = identity); vals = point ids
__global__ __launch_bounds__(256) void ds_keys_kernel(const int* __restrict__ coors, const uint32_t* __restrict__ order,
                                                      int n, DsCols g, uint32_t* __restrict__ keys,
                                                      uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t p = order ? order[i] : (uint32_t)i;
  const int* row = coors + (size_t)p * g.ndim;
  uint32_t key = g.invalid_key;
  if (ds_row_valid(row, g.ndim)) {
    key = 0;
    for (int c = g.first; c <= g.last; ++c) key |= (uint32_t)row[c] << g.shift[c];
  }
  keys[i] = key;
  vals[i] = p;
}

// flag[i] = 1 where sorted position i starts a new voxel (valid rows only; invalid rows are sorted to the tail)
__global__ __launch_bounds__(256) void ds_heads_kernel(const int* __restrict__ coors, const uint32_t* __restrict__ order,
                                                       int n, int ndim, const int* __restrict__ stats,
                                                       uint32_t* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int n_valid = n - stats[DS_MAX_DIM];
  uint32_t f = 0;
  if (i < n_valid) {
    if (i == 0) {
      f = 1;
    } else {
      const int* a = coors + (size_t)order[i] * ndim;
      const int* b = coors + (size_t)order[i - 1] * ndim;
      for (int c = 0; c < ndim; ++c) f |= (a[c] != b[c]) ? 1u : 0u;
    }
  }
  flags[i] = f;
}

__global__ __launch_bounds__(256) void ds_emit_kernel(const int* __restrict__ coors, const uint32_t* __restrict__ order,
                                                      int n, int ndim, const int* __restrict__ stats,
                                                      const uint32_t* __restrict__ flags, const uint32_t* __restrict__ rank,
                                                      int* __restrict__ coors_map, int* __restrict__ out_coors,
                                                      int* __restrict__ seg_start, int* __restrict__ order_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int n_valid = n - stats[DS_MAX_DIM];
  if (i == 0 && n_valid == 0) seg_start[0] = 0;
  if (i >= n) return;
  const uint32_t p = order[i];
  order_out[i] = (int)p;
  if (i >= n_valid) {
    coors_map[p] = -1;
    return;
  }
  const int v = (int)(rank[i] + flags[i]) - 1;   // rank = exclusive scan of the head flags
  coors_map[p] = v;
  if (flags[i]) {
    seg_start[v] = i;
    for (int c = 0; c < ndim; ++c) out_coors[(size_t)v * ndim + c] = coors[(size_t)p * ndim + c];
  }
  if (i == n_valid - 1) seg_start[v + 1] = n_valid;
}

__global__ __launch_bounds__(256) void ds_count_kernel(const int* __restrict__ seg_start, const uint32_t* __restrict__ total,
                                                       int n, int* __restrict__ reduce_count, int* __restrict__ num_voxels) {
  const int m = (int)*total;
  for (int v = blockIdx.x * 256 + threadIdx.x; v < n; v += gridDim.x * 256)
    if (v < m) reduce_count[v] = seg_start[v + 1] - seg_start[v];
  if (blockIdx.x == 0 && threadIdx.x == 0) *num_voxels = m;
}

enum { DS_SUM = 0, DS_MEAN = 1, DS_MAX = 2 };   // reduce_t of scatter_points_cuda.cu:7

template <int MODE>
__global__ __launch_bounds__(256) void ds_reduce_kernel(const float* __restrict__ feats, int F, const int* __restrict__ order,
                                                        const int* __restrict__ seg_start, int num_voxels,
                                                        float* __restrict__ reduced) {
  const long long total = (long long)num_voxels * F;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int v = (int)(e / F), c = (int)(e - (long long)v * F);
    const int s = seg_start[v], t = seg_start[v + 1];
    float acc = MODE == DS_MAX ? -INFINITY : 0.f;
    for (int j = s; j < t; ++j) {
      const float x = feats[(size_t)order[j] * F + c];
      acc = MODE == DS_MAX ? fmaxf(acc, x) : acc + x;
    }
    if (MODE == DS_MEAN) acc = acc / (float)(t - s);   // reduced_feats /= count.to(float) (:237-238)
    reduced[e] = acc;
  }
}

// sum / mean backward: every point takes its voxel's gradient (:113-142)
template <int MODE>
__global__ __launch_bounds__(256) void ds_bwd_add_kernel(float* __restrict__ grad_feats, const float* __restrict__ grad_reduced,
                                                         const int* __restrict__ coors_map, const int* __restrict__ reduce_count,
                                                         int n, int F) {
  const long long total = (long long)n * F;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int p = (int)(e / F), c = (int)(e - (long long)p * F);
    const int v = coors_map[p];
    float g = 0.f;
    if (v >= 0) {
      g = grad_reduced[(size_t)v * F + c];
      if (MODE == DS_MEAN) g = g / (float)reduce_count[v];
    }
    grad_feats[e] = g;
  }
}

// max backward, pass 1: lowest point id attaining the maximum, per (voxel, channel) (:144-170)
__global__ __launch_bounds__(256) void ds_bwd_argmax_kernel(const float* __restrict__ feats, const float* __restrict__ reduced,
                                                            const int* __restrict__ coors_map, int n, int F,
                                                            int* __restrict__ reduce_from) {
  const long long total = (long long)n * F;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int p = (int)(e / F), c = (int)(e - (long long)p * F);
    const int v = coors_map[p];
    if (v < 0) continue;
    if (feats[e] == reduced[(size_t)v * F + c]) atomicMin(&reduce_from[(size_t)v * F + c], p);
  }
}

// pass 2 (:172-190); a (voxel, channel) nobody matched (NaN maximum) is skipped instead of written out of bounds
__global__ __launch_bounds__(256) void ds_bwd_max_scatter_kernel(float* __restrict__ grad_feats, const float* __restrict__ grad_reduced,
                                                                 const int* __restrict__ reduce_from, int n, int num_voxels,
                                                                 int F) {
  const long long total = (long long)num_voxels * F;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e % F);
    const int p = reduce_from[e];
    if (p >= 0 && p < n) grad_feats[(size_t)p * F + c] = grad_reduced[e];
  }
}

static unsigned ds_grid(long long total) {
  long long b = (total + 255) / 256;
  return (unsigned)(b < 1 ? 1 : b > 8192 ? 8192 : b);
}

}  // namespace bevamd

using namespace bevamd;

extern "C" {

size_t bevamd_dynamic_scatter_workspace_bytes(int num_points) {
  const size_t n = (size_t)(num_points > 0 ? num_points : 1);
  return 6 * align_up(n * 4, 256) + align_up(radix_sort_workspace_bytes(n), 256) + align_up(scan_workspace_bytes(n), 256) +
         1024;
}

int bevamd_dynamic_scatter_index(const int* coors, int num_points, int ndim, int* out_coors, int* coors_map,
                                 int* reduce_count, int* order, int* seg_start, int* num_voxels_dev, int* num_voxels_host,
                                 void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(num_points >= 0 && ndim >= 1 && ndim <= DS_MAX_DIM, "dynamic_scatter_index: ndim %d (1..%d)", ndim, DS_MAX_DIM);
  BEVAMD_REQUIRE(num_voxels_dev != nullptr, "dynamic_scatter_index: num_voxels_dev is null");
  if (num_points == 0) {
    BEVAMD_HIP_CHECK(hipMemsetAsync(num_voxels_dev, 0, sizeof(int), stream));
    if (num_voxels_host) { BEVAMD_HIP_CHECK(hipStreamSynchronize(stream)); *num_voxels_host = 0; }
    return BEVAMD_OK;
  }
  BEVAMD_REQUIRE(coors && out_coors && coors_map && reduce_count && order && seg_start, "dynamic_scatter_index: null buffer");
  const size_t need = bevamd_dynamic_scatter_workspace_bytes(num_points);
  if (!ws || ws_bytes < need) {
    set_error("dynamic_scatter_index: workspace too small (%zu < %zu)", ws_bytes, need);
    return BEVAMD_ERR_WORKSPACE;
  }
  const size_t n = (size_t)num_points;
  Carver cv(ws, ws_bytes);
  uint32_t* keys_a = cv.take<uint32_t>(n);
  uint32_t* vals_a = cv.take<uint32_t>(n);
  uint32_t* keys_b = cv.take<uint32_t>(n);
  uint32_t* vals_b = cv.take<uint32_t>(n);
  uint32_t* flags = cv.take<uint32_t>(n);
  uint32_t* rank = cv.take<uint32_t>(n);
  const size_t sort_bytes = align_up(radix_sort_workspace_bytes(n), 256);
  void* sort_ws = cv.take<char>(sort_bytes);
  const size_t scan_bytes = align_up(scan_workspace_bytes(n), 256);
  void* scan_ws = cv.take<char>(scan_bytes);
  int* stats = cv.take<int>(DS_MAX_DIM + 4);   // column maxima, invalid-row count, [DS_MAX_DIM + 1] = voxel total

  // 1. how many bits does each column need?  (one small read-back; the caller needs the voxel count on the host anyway)
  BEVAMD_HIP_CHECK(hipMemsetAsync(stats, 0, (DS_MAX_DIM + 4) * sizeof(int), stream));
  ds_colmax_kernel<<<dim3(ds_grid(num_points) > 512 ? 512 : ds_grid(num_points)), dim3(256), 0, stream>>>(coors, num_points, ndim, stats);
  BEVAMD_LAUNCH_CHECK("ds_colmax");
  int host_stats[DS_MAX_DIM + 1];
  BEVAMD_HIP_CHECK(hipMemcpyAsync(host_stats, stats, sizeof(host_stats), hipMemcpyDeviceToHost, stream));
  BEVAMD_HIP_CHECK(hipStreamSynchronize(stream));
  int bits[DS_MAX_DIM];
  for (int c = 0; c < ndim; ++c) bits[c] = bits_for((uint64_t)host_stats[c] + 1);

  // 2. stable LSD sort over column groups, last column first; each group packs as many columns as fit 31 bits
  //    (bit 31 side: the invalid-row key sits just above the largest valid key of the group)
  const uint32_t* cur_order = nullptr;
  uint32_t *ka = keys_a, *va = vals_a, *kb = keys_b, *vb = vals_b;
  int last = ndim - 1;
  while (last >= 0) {
    DsCols g;
    g.ndim = ndim;
    g.last = last;
    int used = 0, first = last;
    for (int c = last; c >= 0; --c) {
      if (c != last && used + bits[c] > 31) break;
      g.shift[c] = used;
      used += bits[c];
      first = c;
    }
    BEVAMD_REQUIRE(used <= 31, "dynamic_scatter_index: a coordinate needs more than 31 bits");
    g.first = first;
    g.invalid_key = 1u << used;
    ds_keys_kernel<<<dim3(cdiv(num_points, 256)), dim3(256), 0, stream>>>(coors, cur_order, num_points, g, ka, va);
    BEVAMD_LAUNCH_CHECK("ds_keys");
    uint32_t *rk = nullptr, *rv = nullptr;
    int rc = radix_sort_pairs_u32_ex(ka, va, kb, vb, n, used + 1, sort_ws, sort_bytes, stream, &rk, &rv);
    if (rc) return rc;
    cur_order = rv;
    if (rv == va) { uint32_t* t = ka; ka = kb; kb = t; t = va; va = vb; vb = t; }   // next round writes the other pair
    last = first - 1;
  }

  // 3. heads -> voxel ids, coordinates, point->voxel map, segment starts, counts
  ds_heads_kernel<<<dim3(cdiv(num_points, 256)), dim3(256), 0, stream>>>(coors, cur_order, num_points, ndim, stats, flags);
  BEVAMD_LAUNCH_CHECK("ds_heads");
  uint32_t* total = (uint32_t*)(stats + DS_MAX_DIM + 1);
  int rc = exclusive_scan_u32(flags, rank, n, total, scan_ws, scan_bytes, stream);
  if (rc) return rc;
  ds_emit_kernel<<<dim3(cdiv(num_points, 256)), dim3(256), 0, stream>>>(coors, cur_order, num_points, ndim, stats, flags, rank,
                                                                       coors_map, out_coors, seg_start, order);
  BEVAMD_LAUNCH_CHECK("ds_emit");
  ds_count_kernel<<<dim3(ds_grid(num_points) > 1024 ? 1024 : ds_grid(num_points)), dim3(256), 0, stream>>>(seg_start, total, num_points,
                                                                                                 reduce_count, num_voxels_dev);
  BEVAMD_LAUNCH_CHECK("ds_count");
  if (num_voxels_host) {
    BEVAMD_HIP_CHECK(hipMemcpyAsync(num_voxels_host, num_voxels_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
    BEVAMD_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return BEVAMD_OK;
}

int bevamd_dynamic_scatter_reduce(const float* feats, int num_feats, const int* order, const int* seg_start, int num_voxels,
                                  int reduce_type, float* reduced, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(reduce_type >= DS_SUM && reduce_type <= DS_MAX, "dynamic_scatter_reduce: reduce_type %d (0 sum, 1 mean, 2 max)", reduce_type);
  BEVAMD_REQUIRE(num_voxels >= 0 && num_feats >= 0, "dynamic_scatter_reduce: bad sizes");
  if (num_voxels == 0 || num_feats == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(feats && order && seg_start && reduced, "dynamic_scatter_reduce: null buffer");
  const dim3 grid(ds_grid((long long)num_voxels * num_feats)), block(256);
  if (reduce_type == DS_SUM) ds_reduce_kernel<DS_SUM><<<grid, block, 0, stream>>>(feats, num_feats, order, seg_start, num_voxels, reduced);
  else if (reduce_type == DS_MEAN) ds_reduce_kernel<DS_MEAN><<<grid, block, 0, stream>>>(feats, num_feats, order, seg_start, num_voxels, reduced);
  else ds_reduce_kernel<DS_MAX><<<grid, block, 0, stream>>>(feats, num_feats, order, seg_start, num_voxels, reduced);
  BEVAMD_LAUNCH_CHECK("ds_reduce");
  return BEVAMD_OK;
}

int bevamd_dynamic_scatter_backward(float* grad_feats, const float* grad_reduced, const float* feats, const float* reduced,
                                    const int* coors_map, const int* reduce_count, int num_points, int num_voxels,
                                    int num_feats, int reduce_type, int* reduce_from_ws, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(reduce_type >= DS_SUM && reduce_type <= DS_MAX, "dynamic_scatter_backward: reduce_type %d (0 sum, 1 mean, 2 max)", reduce_type);
  BEVAMD_REQUIRE(num_points >= 0 && num_voxels >= 0 && num_feats >= 0, "dynamic_scatter_backward: bad sizes");
  if (num_points == 0 || num_feats == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(grad_feats != nullptr, "dynamic_scatter_backward: grad_feats is null");
  const long long ne = (long long)num_points * num_feats;
  if (num_voxels == 0) {   // grad_feats.fill_(0) and return (:268-270)
    BEVAMD_HIP_CHECK(hipMemsetAsync(grad_feats, 0, (size_t)ne * sizeof(float), stream));
    return BEVAMD_OK;
  }
  BEVAMD_REQUIRE(grad_reduced && coors_map, "dynamic_scatter_backward: null buffer");
  if (reduce_type != DS_MAX) {
    BEVAMD_REQUIRE(reduce_type == DS_SUM || reduce_count, "dynamic_scatter_backward: mean needs reduce_count");
    if (reduce_type == DS_SUM) ds_bwd_add_kernel<DS_SUM><<<dim3(ds_grid(ne)), dim3(256), 0, stream>>>(grad_feats, grad_reduced, coors_map, reduce_count, num_points, num_feats);
    else ds_bwd_add_kernel<DS_MEAN><<<dim3(ds_grid(ne)), dim3(256), 0, stream>>>(grad_feats, grad_reduced, coors_map, reduce_count, num_points, num_feats);
    BEVAMD_LAUNCH_CHECK("ds_bwd_add");
    return BEVAMD_OK;
  }
  BEVAMD_REQUIRE(feats && reduced && reduce_from_ws, "dynamic_scatter_backward: max needs feats, reduced and the reduce_from workspace");
  const size_t ve = (size_t)num_voxels * num_feats;
  BEVAMD_HIP_CHECK(hipMemsetAsync(grad_feats, 0, (size_t)ne * sizeof(float), stream));
  int rc = device_fill_u32((uint32_t*)reduce_from_ws, ve, (uint32_t)num_points, stream);   // at::full(..., num_input) (:300)
  if (rc) return rc;
  ds_bwd_argmax_kernel<<<dim3(ds_grid(ne)), dim3(256), 0, stream>>>(feats, reduced, coors_map, num_points, num_feats, reduce_from_ws);
  BEVAMD_LAUNCH_CHECK("ds_bwd_argmax");
  ds_bwd_max_scatter_kernel<<<dim3(ds_grid((long long)ve)), dim3(256), 0, stream>>>(grad_feats, grad_reduced, reduce_from_ws, num_points,
                                                                                  num_voxels, num_feats);
  BEVAMD_LAUNCH_CHECK("ds_bwd_max_scatter");
  return BEVAMD_OK;
}

}  // extern "C"
