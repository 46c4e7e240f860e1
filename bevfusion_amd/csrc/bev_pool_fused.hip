// Fused depth (x) context -> BEV for gfx950.
//
// The reference materialises the camera feature volume before pooling it:
//   models/vtransforms/depth_lss.py:92-97   x = depth.unsqueeze(1) * ctx.unsqueeze(2)   -> [BN, C, D, fH, fW] (638 MB fp32)
//                                           .view().permute(0,1,3,4,5,2)                 -> a strided view
//   models/vtransforms/base.py:141-176      reshape (copy), mask gather, sort gather, interval sum
// i.e. ~5 GB of HBM traffic per frame around an op whose inputs are 13 MB.  Here
//   out[cell, c] = sum over the frustum points p of the cell of  depth[p] * ctx[pixel(p), c]
// is evaluated directly from depth [B*N, D, fH, fW] (softmax output, fp32) and the channels-last context
// [B*N*fH*fW, C] (fp32 or bf16): the volume never exists.  The pooling plan (order / cell_start: stable sort of the
// points by BEV cell, bev_pool.hip) is the same one the unfused op uses, so the summation order per cell — and with it
// the rounding — is the reference's (row order inside an interval = point order).
//
// Not the same byte denominator as bev_pool: algorithmic bytes = N_kept*4 (depth) + n_pixels*C*s
// (context, read once) + B*D*H*W*C*4 (output) = 54 MB at the flagship frame; the context rows are re-read ~107x
// each from L2, so this kernel is L2-bandwidth-bound, not HBM-bound.
//
// One 64-lane wave per BEV cell; a context row (C channels) is covered by C/VEC lanes holding 16 bytes each, so
// floor(64 / lanes-per-row) points are processed per wave instruction; the point's depth is a same-address (broadcast)
// load for the lanes of its row slot; U independent (index -> depth, context) load pairs in flight per lane; row slots
// are folded with __shfl; every cell (empty ones as zeros) is stored exactly once.
#include <stdlib.h>

#include "common.h"

namespace bevamd {

struct alignas(16) FU4 { uint32_t x, y, z, w; };
struct FusedDims { int B, D, H, W, C; };

template <int VEC> struct FAcc { float v[VEC]; };

__device__ __forceinline__ void fma_row(FAcc<4>& a, float d, const float4& f) {
  a.v[0] = fmaf(d, f.x, a.v[0]); a.v[1] = fmaf(d, f.y, a.v[1]); a.v[2] = fmaf(d, f.z, a.v[2]); a.v[3] = fmaf(d, f.w, a.v[3]);
}
__device__ __forceinline__ void fma_row(FAcc<8>& a, float d, const FU4& u) {
  a.v[0] = fmaf(d, __uint_as_float(u.x << 16), a.v[0]); a.v[1] = fmaf(d, __uint_as_float(u.x & 0xFFFF0000u), a.v[1]);
  a.v[2] = fmaf(d, __uint_as_float(u.y << 16), a.v[2]); a.v[3] = fmaf(d, __uint_as_float(u.y & 0xFFFF0000u), a.v[3]);
  a.v[4] = fmaf(d, __uint_as_float(u.z << 16), a.v[4]); a.v[5] = fmaf(d, __uint_as_float(u.z & 0xFFFF0000u), a.v[5]);
  a.v[6] = fmaf(d, __uint_as_float(u.w << 16), a.v[6]); a.v[7] = fmaf(d, __uint_as_float(u.w & 0xFFFF0000u), a.v[7]);
}

// frustum point p = ((cam * D + d) * fH + h) * fW + w  ->  context row  cam * fH*fW + h*fW + w
__device__ __forceinline__ uint32_t pixel_of(uint32_t p, uint32_t dfhw, uint32_t fhw) {
  const uint32_t cam = p / dfhw;
  const uint32_t rem = p - cam * dfhw;
  return cam * fhw + rem % fhw;
}

// one wave, one BEV cell
template <typename VecT, int VEC, int U>
__device__ __forceinline__ void fused_cell(const float* __restrict__ depth, const VecT* __restrict__ ctx,
                                           const uint32_t* __restrict__ order, const uint32_t* __restrict__ cell_start,
                                           uint32_t cell, float* __restrict__ out, int lpr, int rpi, uint32_t dfhw, uint32_t fhw,
                                           const FusedDims& s) {
  const int lane = threadIdx.x & 63;
  const int slot = lane / lpr;
  const int cv = lane - slot * lpr;
  const uint32_t start = cell_start[cell];
  const int len = (int)(cell_start[cell + 1] - start);
  FAcc<VEC> acc;
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc.v[j] = 0.f;
  if (len > 0) {  // wave-uniform
    if (slot < rpi) {
      // Every trip handles U points per row slot with all loads independent; the LAST trip is predicated (a missing point is
      // the cell's first point with weight 0: acc + 0 * ctx is exact), so a 40-point cell costs 4 batched round trips and no
      // one-point-at-a-time tail; the indices of trip t+1 are requested before the depth / context loads of trip t.
      const uint32_t* ord = order + start;
      uint32_t pn[U];
#pragma unroll
      for (int u = 0; u < U; ++u) pn[u] = ord[slot + u * rpi < len ? slot + u * rpi : 0];
      for (int r = slot; r < len; r += U * rpi) {
        uint32_t p[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { p[u] = pn[u]; ok[u] = r + u * rpi < len; }
        const int rn = r + U * rpi;
#pragma unroll
        for (int u = 0; u < U; ++u) pn[u] = ord[rn + u * rpi < len ? rn + u * rpi : 0];
        float d[U];
        VecT a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          d[u] = depth[p[u]];
          a[u] = ctx[(size_t)pixel_of(p[u], dfhw, fhw) * lpr + cv];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) fma_row(acc, ok[u] ? d[u] : 0.f, a[u]);
      }
    }
    // fold the row slots into slot 0 (wave-uniform trip count)
    FAcc<VEC> tot = acc;
    for (int sl = 1; sl < rpi; ++sl) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) tot.v[j] += __shfl(acc.v[j], lane + sl * lpr, 64);
    }
    acc = tot;
  }
  if (slot == 0) {
    // rank = x*(W*D*B) + y*(D*B) + z*B + b (bev_pool.py:86-91) -> out[b, z, x, y, :] (bev_pool_cuda.cu:33-35)
    uint32_t r = cell;
    const int gb = r % s.B; r /= s.B;
    const int gz = r % s.D; r /= s.D;
    const int gy = r % s.W; r /= s.W;
    const int gx = (int)r;
    float4* o = (float4*)(out + ((((size_t)gb * s.D + gz) * s.H + gx) * s.W + gy) * (size_t)s.C + (size_t)cv * VEC);
#pragma unroll
    for (int j = 0; j < VEC / 4; ++j) o[j] = make_float4(acc.v[4 * j], acc.v[4 * j + 1], acc.v[4 * j + 2], acc.v[4 * j + 3]);
  }
}

template <typename VecT, int VEC, int U>
__global__ __launch_bounds__(256) void bev_pool_fused_cells_kernel(
    const float* __restrict__ depth, const VecT* __restrict__ ctx, const uint32_t* __restrict__ order,
    const uint32_t* __restrict__ cell_start, uint32_t ncells, float* __restrict__ out, int lpr, int rpi, uint32_t dfhw,
    uint32_t fhw, FusedDims s) {
  // Cells are numbered b-fastest (the reference's rank); waves walk them FRAME-major instead: neighbouring waves then share
  // one frame's context (5.4 MB, L2-resident) rather than all B of them, and write neighbouring output rows of one frame.
  const uint32_t lin = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (lin >= ncells) return;
  const uint32_t per_frame = ncells / (uint32_t)s.B;
  const uint32_t cell = (lin % per_frame) * (uint32_t)s.B + lin / per_frame;
  fused_cell<VecT, VEC, U>(depth, ctx, order, cell_start, cell, out, lpr, rpi, dfhw, fhw, s);
}

// ---- camera-sector schedule -----------------------------------------------------------------------------------------------
// PMC of the kernel above at 8 frames (profiles/r03_pmc_infer_per_kernel.txt): 1450 MB of HBM traffic for 432 MB of
// algorithmic bytes, L2 hit 80 %.  Workgroups go to the 8 XCDs round-robin, so every XCD sweeps the whole BEV and needs the
// context rows of every camera whose rays cross the x-band being swept: more than its 4 MB L2 holds.  The schedule below gives
// each XCD a compact SECTOR instead: cells are ordered by (frame, camera of the cell's first point, cell) — a permutation built
// once per plan with one radix sort — and cut into 8 contiguous chunks of equal WORK (points + 1 per cell).  An XCD then reads
// about one camera's context at a time (0.9 MB) and neighbouring cells of a sector — neighbouring output rows — leave from the
// same L2.  Empty cells (zeros to write) are spread over the frame's camera buckets by x-band.
__global__ __launch_bounds__(256) void bev_fused_schedule_keys_kernel(const uint32_t* __restrict__ order,
                                                                      const uint32_t* __restrict__ cell_start, uint32_t ncells,
                                                                      int B, uint32_t dfhw, int ncam, uint32_t* __restrict__ keys,
                                                                      uint32_t* __restrict__ vals, uint32_t* __restrict__ work) {
  const uint32_t cell = blockIdx.x * 256u + threadIdx.x;
  if (cell >= ncells) return;
  const uint32_t per_frame = ncells / (uint32_t)B;
  const uint32_t b = cell % (uint32_t)B, local = cell / (uint32_t)B;
  const uint32_t start = cell_start[cell], len = cell_start[cell + 1] - start;
  uint32_t bucket;
  if (len > 0) {
    bucket = order[start] / dfhw;                                   // (frame, camera) of the cell's first point
    const uint32_t top = (uint32_t)(B * ncam);
    bucket = bucket < top ? bucket : top - 1u;
  } else {
    bucket = b * (uint32_t)ncam + (uint32_t)(((unsigned long long)local * (unsigned)ncam) / per_frame);
  }
  keys[cell] = bucket * per_frame + local;
  vals[cell] = cell;
  work[cell] = len + 1u;
}

// work of the permuted cells (to be scanned) / the 9 chunk boundaries from the scan
__global__ __launch_bounds__(256) void bev_fused_schedule_work_kernel(const uint32_t* __restrict__ perm,
                                                                      const uint32_t* __restrict__ work, uint32_t ncells,
                                                                      uint32_t* __restrict__ pwork) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < ncells) pwork[i] = work[perm[i]];
}
__global__ void bev_fused_schedule_cuts_kernel(const uint32_t* __restrict__ scan, const uint32_t* __restrict__ total,
                                               uint32_t ncells, uint32_t* __restrict__ xcd_start) {
  const int j = threadIdx.x;
  if (j > 8) return;
  if (j == 0) { xcd_start[0] = 0; return; }
  if (j == 8) { xcd_start[8] = ncells; return; }
  const uint32_t target = (uint32_t)(((unsigned long long)(*total) * (unsigned)j) / 8ull);
  uint32_t lo = 0, hi = ncells;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (scan[mid] < target) lo = mid + 1; else hi = mid;
  }
  xcd_start[j] = lo;
}

template <typename VecT, int VEC, int U>
__global__ __launch_bounds__(256) void bev_pool_fused_sched_kernel(
    const float* __restrict__ depth, const VecT* __restrict__ ctx, const uint32_t* __restrict__ order,
    const uint32_t* __restrict__ cell_start, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ xcd_start,
    float* __restrict__ out, int lpr, int rpi, uint32_t dfhw, uint32_t fhw, FusedDims s) {
  // XCD x = blockIdx.x % 8 walks its chunk of the permuted cells, one wave per cell, the XCD's waves striding together
  const uint32_t xcd = blockIdx.x & 7u, wave = (blockIdx.x >> 3) * 4u + (threadIdx.x >> 6);
  const uint32_t nwaves = (gridDim.x >> 3) * 4u;
  const uint32_t end = xcd_start[xcd + 1];
  for (uint32_t i = xcd_start[xcd] + wave; i < end; i += nwaves)
    fused_cell<VecT, VEC, U>(depth, ctx, order, cell_start, perm[i], out, lpr, rpi, dfhw, fhw, s);
}

// ---- backward (fp32 context) ---------------------------------------------------------------------------------------
//   d_depth[p]      = sum_c  g[cell(p), c] * ctx[pixel(p), c]           (0 for points the range mask dropped)
//   d_ctx[pixel, c] = sum over the D frustum points p of the pixel of  depth[p] * g[cell(p), c]
// `cell_of_point[p]` (rank of the BEV cell, or >= ncells when dropped) is the plan's sort key in point order.

__device__ __forceinline__ size_t grad_cell_offset(uint32_t r, const FusedDims& s) {
  const int gb = r % s.B; r /= s.B;
  const int gz = r % s.D; r /= s.D;
  const int gy = r % s.W; r /= s.W;
  const int gx = (int)r;
  return ((((size_t)gb * s.D + gz) * s.H + gx) * s.W + gy) * (size_t)s.C;
}

// cell_of_point[order[j]] = ranks_sorted[j]
__global__ __launch_bounds__(256) void bev_cell_of_point_kernel(const uint32_t* __restrict__ order,
                                                                const uint32_t* __restrict__ ranks_sorted, int n,
                                                                uint32_t* __restrict__ cell_of_point) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n) cell_of_point[order[j]] = ranks_sorted[j];
}

// one row slot (lpr lanes) per frustum point: dot product of the cell's gradient row with the pixel's context row
__global__ __launch_bounds__(256) void bev_pool_fused_bwd_depth_kernel(
    const float* __restrict__ out_grad, const float4* __restrict__ ctx, const uint32_t* __restrict__ cell_of_point,
    uint32_t ncells, int n, float* __restrict__ d_depth, int lpr, int rpi, uint32_t dfhw, uint32_t fhw, FusedDims s) {
  const int lane = threadIdx.x & 63;
  const int slot = lane / lpr;
  const int cv = lane - slot * lpr;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t p = wave * rpi + slot;
  const bool active = slot < rpi && p < (size_t)n;
  float part = 0.f;
  if (active) {
    const uint32_t r = cell_of_point[p];
    if (r < ncells) {
      const float4 g = *(const float4*)(out_grad + grad_cell_offset(r, s) + (size_t)cv * 4);
      const float4 c = ctx[(size_t)pixel_of((uint32_t)p, dfhw, fhw) * lpr + cv];
      part = g.x * c.x + g.y * c.y + g.z * c.z + g.w * c.w;
    }
  }
  // sum the lpr lanes of the slot: segmented butterfly inside the slot's lane range (every lane shuffles)
  for (int o = 1; o < lpr; o <<= 1) {
    const float t = __shfl_down(part, o, 64);
    if (cv + o < lpr) part += t;
  }
  if (active && cv == 0) d_depth[p] = part;
}

// one row slot per context pixel: walk its D depth bins
__global__ __launch_bounds__(256) void bev_pool_fused_bwd_ctx_kernel(
    const float* __restrict__ out_grad, const float* __restrict__ depth, const uint32_t* __restrict__ cell_of_point,
    uint32_t ncells, int npix, int depth_bins, float4* __restrict__ d_ctx, int lpr, int rpi, uint32_t fhw, FusedDims s) {
  const int lane = threadIdx.x & 63;
  const int slot = lane / lpr;
  const int cv = lane - slot * lpr;
  if (slot >= rpi) return;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t pix = wave * rpi + slot;
  if (pix >= (size_t)npix) return;
  const uint32_t cam = (uint32_t)(pix / fhw), inner = (uint32_t)(pix - (size_t)cam * fhw);
  const size_t p0 = (size_t)cam * depth_bins * fhw + inner;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int d = 0; d < depth_bins; ++d) {
    const size_t p = p0 + (size_t)d * fhw;
    const uint32_t r = cell_of_point[p];
    if (r < ncells) {
      const float w = depth[p];
      const float4 g = *(const float4*)(out_grad + grad_cell_offset(r, s) + (size_t)cv * 4);
      acc.x = fmaf(w, g.x, acc.x); acc.y = fmaf(w, g.y, acc.y); acc.z = fmaf(w, g.z, acc.z); acc.w = fmaf(w, g.w, acc.w);
    }
  }
  d_ctx[pix * lpr + cv] = acc;
}

}  // namespace bevamd

using namespace bevamd;

extern "C" {

/* out [b, d, h, w, c] fp32 (every cell written once, no pre-zeroing) from
 *   depth [n] fp32 (n = cams * depth_bins * fh * fw frustum points in the order of the geometry the plan was built from),
 *   ctx   [cams * fh * fw, c] channels-last, fp32 (ctx_is_bf16 = 0) or bf16 bits (1),
 *   order / cell_start: the pooling plan of bevamd_bev_pool_prepare[_from_geom]. */
int bevamd_bev_pool_fused_forward(const float* depth, const void* ctx, int ctx_is_bf16, const uint32_t* order,
                                  const uint32_t* cell_start, float* out, int n, int c, int depth_bins, int fh, int fw,
                                  int b, int d, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(n >= 0 && c > 0 && depth_bins > 0 && fh > 0 && fw > 0 && b > 0 && d > 0 && h > 0 && w > 0,
                 "bev_pool_fused_forward: bad sizes");
  const long long per_cam = (long long)depth_bins * fh * fw;
  BEVAMD_REQUIRE(n % per_cam == 0, "bev_pool_fused_forward: n=%d is not a multiple of depth_bins*fh*fw=%lld", n, per_cam);
  const unsigned long long ncells64 = (unsigned long long)b * d * h * w;
  BEVAMD_REQUIRE(ncells64 < 0xFFFFFFF0ull, "bev_pool_fused_forward: b*d*h*w must be < 2^32 - 16");
  BEVAMD_REQUIRE(out && cell_start && (n == 0 || (depth && ctx && order)), "bev_pool_fused_forward: null buffer");
  const int vec = ctx_is_bf16 ? 8 : 4;
  BEVAMD_REQUIRE(c % vec == 0 && c / vec <= 64 && ((uintptr_t)ctx & 15) == 0 && ((uintptr_t)out & 15) == 0,
                 "bev_pool_fused_forward: c=%d must be a multiple of %d with at most 64 lanes per row, 16-byte aligned buffers",
                 c, vec);
  const uint32_t ncells = (uint32_t)ncells64;
  const int lpr = c / vec, rpi = 64 / lpr;
  FusedDims s{b, d, h, w, c};
  dim3 grid(cdiv(ncells, 4)), block(256);
  if (ctx_is_bf16)
    bev_pool_fused_cells_kernel<FU4, 8, 4><<<grid, block, 0, stream>>>(depth, (const FU4*)ctx, order, cell_start, ncells, out,
                                                                       lpr, rpi, (uint32_t)per_cam, (uint32_t)(fh * fw), s);
  else
    bev_pool_fused_cells_kernel<float4, 4, 4><<<grid, block, 0, stream>>>(depth, (const float4*)ctx, order, cell_start, ncells,
                                                                          out, lpr, rpi, (uint32_t)per_cam, (uint32_t)(fh * fw), s);
  BEVAMD_LAUNCH_CHECK("bev_pool_fused_cells");
  return BEVAMD_OK;
}

/* Camera-sector schedule of the fused pooling for a plan (static per calibration, like the plan): perm [b*d*h*w] = the cells
 * ordered by (frame, camera of the cell's first point, cell), xcd_start [9] = the boundaries of 8 chunks of equal work; see
 * bev_pool_fused_sched_kernel.  ws: bevamd_bev_pool_fused_schedule_workspace_bytes(b*d*h*w). */
size_t bevamd_bev_pool_fused_schedule_workspace_bytes(int ncells) {
  if (ncells <= 0) return 0;
  const size_t a = align_up((size_t)ncells * 4, 256);
  const size_t s1 = radix_sort_workspace_bytes((size_t)ncells), s2 = scan_workspace_bytes((size_t)ncells);
  return 6 * a + 256 + align_up(s1 > s2 ? s1 : s2, 256);
}

int bevamd_bev_pool_fused_schedule(const uint32_t* order, const uint32_t* cell_start, int n, int depth_bins, int fh, int fw,
                                   int b, int d, int h, int w, uint32_t* perm, uint32_t* xcd_start, void* ws, size_t ws_bytes,
                                   void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(n > 0 && depth_bins > 0 && fh > 0 && fw > 0 && b > 0 && d > 0 && h > 0 && w > 0, "bev_pool_fused_schedule: bad sizes");
  const long long per_cam = (long long)depth_bins * fh * fw;
  BEVAMD_REQUIRE(n % (per_cam * b) == 0, "bev_pool_fused_schedule: n=%d is not b * cameras * depth_bins*fh*fw", n);
  const int ncam = (int)(n / (per_cam * b));
  const unsigned long long ncells64 = (unsigned long long)b * d * h * w;
  BEVAMD_REQUIRE(ncells64 < 0x7FFFFFF0ull, "bev_pool_fused_schedule: b*d*h*w too large");
  BEVAMD_REQUIRE((ncells64 / b) * ((unsigned long long)b * ncam) < 0xFFFFFFF0ull, "bev_pool_fused_schedule: sort key overflows 32 bits");
  BEVAMD_REQUIRE(order && cell_start && perm && xcd_start, "bev_pool_fused_schedule: null buffer");
  const uint32_t ncells = (uint32_t)ncells64;
  if (!ws || ws_bytes < bevamd_bev_pool_fused_schedule_workspace_bytes((int)ncells)) {
    set_error("bev_pool_fused_schedule: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  Carver cv(ws, ws_bytes);
  uint32_t* keys_a = cv.take<uint32_t>(ncells);
  uint32_t* vals_a = cv.take<uint32_t>(ncells);
  uint32_t* keys_b = cv.take<uint32_t>(ncells);
  uint32_t* work = cv.take<uint32_t>(ncells);
  uint32_t* pwork = cv.take<uint32_t>(ncells);
  uint32_t* spare = cv.take<uint32_t>(ncells);
  uint32_t* total = cv.take<uint32_t>(1);
  (void)spare;
  void* sws = cv.base + cv.off;
  const size_t sws_bytes = ws_bytes - cv.off;
  const dim3 grid(cdiv(ncells, 256)), block(256);
  bev_fused_schedule_keys_kernel<<<grid, block, 0, stream>>>(order, cell_start, ncells, b, (uint32_t)per_cam, ncam, keys_a, vals_a, work);
  BEVAMD_LAUNCH_CHECK("bev_fused_schedule_keys");
  const uint64_t key_max = (uint64_t)(ncells64 / b) * ((uint64_t)b * ncam);
  int rc = radix_sort_pairs_u32(keys_a, vals_a, keys_b, perm, (size_t)ncells, bits_for(key_max + 1), sws, sws_bytes, stream);
  if (rc) return rc;
  bev_fused_schedule_work_kernel<<<grid, block, 0, stream>>>(perm, work, ncells, pwork);
  BEVAMD_LAUNCH_CHECK("bev_fused_schedule_work");
  rc = exclusive_scan_u32(pwork, pwork, (size_t)ncells, total, sws, sws_bytes, stream);
  if (rc) return rc;
  bev_fused_schedule_cuts_kernel<<<1, 64, 0, stream>>>(pwork, total, ncells, xcd_start);
  BEVAMD_LAUNCH_CHECK("bev_fused_schedule_cuts");
  return BEVAMD_OK;
}

/* bevamd_bev_pool_fused_forward walked through the schedule: same sums in the same per-cell order (bit-identical output). */
int bevamd_bev_pool_fused_forward_scheduled(const float* depth, const void* ctx, int ctx_is_bf16, const uint32_t* order,
                                            const uint32_t* cell_start, const uint32_t* perm, const uint32_t* xcd_start,
                                            float* out, int n, int c, int depth_bins, int fh, int fw, int b, int d, int h, int w,
                                            void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(n > 0 && c > 0 && depth_bins > 0 && fh > 0 && fw > 0 && b > 0 && d > 0 && h > 0 && w > 0,
                 "bev_pool_fused_forward_scheduled: bad sizes");
  const long long per_cam = (long long)depth_bins * fh * fw;
  BEVAMD_REQUIRE(n % per_cam == 0, "bev_pool_fused_forward_scheduled: n=%d is not a multiple of depth_bins*fh*fw=%lld", n, per_cam);
  BEVAMD_REQUIRE((unsigned long long)b * d * h * w < 0x7FFFFFF0ull, "bev_pool_fused_forward_scheduled: b*d*h*w too large");
  BEVAMD_REQUIRE(out && cell_start && depth && ctx && order && perm && xcd_start, "bev_pool_fused_forward_scheduled: null buffer");
  const int vec = ctx_is_bf16 ? 8 : 4;
  BEVAMD_REQUIRE(c % vec == 0 && c / vec <= 64 && ((uintptr_t)ctx & 15) == 0 && ((uintptr_t)out & 15) == 0,
                 "bev_pool_fused_forward_scheduled: c=%d must be a multiple of %d with at most 64 lanes per row, 16-byte aligned buffers",
                 c, vec);
  const int lpr = c / vec, rpi = 64 / lpr;
  FusedDims s{b, d, h, w, c};
  // Workgroups per XCD (measured at 8 frames, us per launch): 32: 1836, 64: 1180, 128: 751, 256: 598, 512: 554 — the kernel
  // lives on waves in flight (every cell is a chain of dependent round trips), so more than fit at once (32 CUs x 8) still helps:
  // the next workgroup starts the moment one drains.  BEVAMD_FUSED_SCHED_WGS overrides (tuning).
  static int wgs_per_xcd = 0;
  if (wgs_per_xcd == 0) {
    const char* e = getenv("BEVAMD_FUSED_SCHED_WGS");
    wgs_per_xcd = e ? atoi(e) : 512;
    if (wgs_per_xcd < 1) wgs_per_xcd = 512;
  }
  const dim3 grid(8 * wgs_per_xcd), block(256);
  if (ctx_is_bf16)
    bev_pool_fused_sched_kernel<FU4, 8, 4><<<grid, block, 0, stream>>>(depth, (const FU4*)ctx, order, cell_start, perm, xcd_start, out,
                                                                       lpr, rpi, (uint32_t)per_cam, (uint32_t)(fh * fw), s);
  else
    bev_pool_fused_sched_kernel<float4, 4, 4><<<grid, block, 0, stream>>>(depth, (const float4*)ctx, order, cell_start, perm, xcd_start,
                                                                          out, lpr, rpi, (uint32_t)per_cam, (uint32_t)(fh * fw), s);
  BEVAMD_LAUNCH_CHECK("bev_pool_fused_sched");
  return BEVAMD_OK;
}

/* cell_of_point [n] u32: the BEV-cell rank of every frustum point in POINT order (>= b*d*h*w for dropped points),
 * derived from the plan (order, ranks_sorted).  Static per plan; needed by the backward only. */
int bevamd_bev_pool_cell_of_point(const uint32_t* order, const uint32_t* ranks_sorted, int n, uint32_t* cell_of_point,
                                  void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(n >= 0, "bev_pool_cell_of_point: n < 0");
  if (n == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(order && ranks_sorted && cell_of_point, "bev_pool_cell_of_point: null buffer");
  bev_cell_of_point_kernel<<<dim3(cdiv(n, 256)), dim3(256), 0, stream>>>(order, ranks_sorted, n, cell_of_point);
  BEVAMD_LAUNCH_CHECK("bev_cell_of_point");
  return BEVAMD_OK;
}

/* Backward of bevamd_bev_pool_fused_forward for fp32 context: out_grad [b,d,h,w,c] -> d_depth [n] and d_ctx [cams*fh*fw, c]
 * (both fully written, no atomics: one row slot per frustum point / per context pixel). */
int bevamd_bev_pool_fused_backward(const float* out_grad, const float* depth, const float* ctx,
                                   const uint32_t* cell_of_point, float* d_depth, float* d_ctx, int n, int c,
                                   int depth_bins, int fh, int fw, int b, int d, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(n >= 0 && c > 0 && depth_bins > 0 && fh > 0 && fw > 0 && b > 0 && d > 0 && h > 0 && w > 0,
                 "bev_pool_fused_backward: bad sizes");
  const long long per_cam = (long long)depth_bins * fh * fw;
  BEVAMD_REQUIRE(n % per_cam == 0, "bev_pool_fused_backward: n=%d is not a multiple of depth_bins*fh*fw=%lld", n, per_cam);
  if (n == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(out_grad && depth && ctx && cell_of_point && d_depth && d_ctx, "bev_pool_fused_backward: null buffer");
  BEVAMD_REQUIRE(c % 4 == 0 && c / 4 <= 64 && (((uintptr_t)ctx | (uintptr_t)out_grad | (uintptr_t)d_ctx) & 15) == 0,
                 "bev_pool_fused_backward: c=%d must be a multiple of 4 (<= 256), 16-byte aligned buffers", c);
  const uint32_t ncells = (uint32_t)((unsigned long long)b * d * h * w);
  const int lpr = c / 4, rpi = 64 / lpr;
  const int npix = (int)(n / depth_bins);
  FusedDims s{b, d, h, w, c};
  bev_pool_fused_bwd_depth_kernel<<<dim3(cdiv(cdiv(n, rpi), 4)), dim3(256), 0, stream>>>(
      out_grad, (const float4*)ctx, cell_of_point, ncells, n, d_depth, lpr, rpi, (uint32_t)per_cam, (uint32_t)(fh * fw), s);
  BEVAMD_LAUNCH_CHECK("bev_pool_fused_bwd_depth");
  bev_pool_fused_bwd_ctx_kernel<<<dim3(cdiv(cdiv(npix, rpi), 4)), dim3(256), 0, stream>>>(
      out_grad, depth, cell_of_point, ncells, npix, depth_bins, (float4*)d_ctx, lpr, rpi, (uint32_t)(fh * fw), s);
  BEVAMD_LAUNCH_CHECK("bev_pool_fused_bwd_ctx");
  return BEVAMD_OK;
}

}  // extern "C"
