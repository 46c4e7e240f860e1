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
      const uint32_t* ord = order + start;
      int r = slot;
      for (; r + (U - 1) * rpi < len; r += U * rpi) {
        uint32_t p[U];
        float d[U];
        VecT a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) p[u] = ord[r + u * rpi];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          d[u] = depth[p[u]];
          a[u] = ctx[(size_t)pixel_of(p[u], dfhw, fhw) * lpr + cv];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) fma_row(acc, d[u], a[u]);
      }
      for (; r < len; r += rpi) {
        const uint32_t p = ord[r];
        const float d = depth[p];
        const VecT a = ctx[(size_t)pixel_of(p, dfhw, fhw) * lpr + cv];
        fma_row(acc, d, a);
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
