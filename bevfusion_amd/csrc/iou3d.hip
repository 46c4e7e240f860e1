// Rotated-box BEV overlap / IoU / NMS for gfx950.
//
// Replaces (reference, /root/reference/mmdet3d/ops/iou3d):
//   src/iou3d_kernel.cu:126-229   box_overlap / iou_bev          (polygon of edge crossings + contained corners,
//                                                                 vertices ordered by atan2 about their centroid, shoelace)
//   src/iou3d_kernel.cu:231-262   boxes_overlap_kernel / boxes_iou_bev_kernel   (16x16 threads per block)
//   src/iou3d_kernel.cu:264-345   nms_kernel / nms_normal_kernel (64-box blocks, one 64-bit suppression word per pair of blocks)
//   src/iou3d.cpp:96-180          nms_gpu / nms_normal_gpu: cudaMalloc + kernel + cudaMemcpy of the whole mask to the
//                                 host + a serial CPU sweep + cudaFree, per call
//
// MI355X formulation
//   * pairwise: one wave per box of A against 64 boxes of B — the A box and everything derived from it (centre, rotated
//     corners, sin/cos) are wave-uniform, the B boxes are one coalesced load per lane, the 64 results one 256-byte store;
//     sin/cos are evaluated once per box and thread, not once per containment test;
//   * NMS mask: one wave per (row block, column block) pair with column >= row (the lower triangle is never read);
//     64 lanes = the 64 rows of the block, column boxes staged in LDS, one u64 word per lane;
//   * NMS sweep ON THE DEVICE: one workgroup walks the row blocks; the greedy pass inside a block works on the 64
//     diagonal words held in LDS, the surviving rows then OR their mask rows into the running `removed` words in parallel
//     (coalesced across words).  No mask copy to the host, no malloc/free; the kept count can stay on the device.
// Arithmetic follows the reference in fp32 (same tests, same EPS / MARGIN, same two-formula intersection); sinf / cosf /
// atan2f are the device's, so results agree with the reference to rounding, not bit for bit (tolerance in the tests).
#include "common.h"

namespace bevamd {
namespace iou3d {

constexpr float EPS = 1e-8f;
struct Pt { float x, y; };

__device__ __forceinline__ float cross3(Pt p1, Pt p2, Pt p0) { return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y); }

// everything that depends on one box only
struct Box {
  float x1, y1, x2, y2;
  float cin, sin_;  // cos(-angle), sin(-angle): rotation used by the containment test
  Pt c[5];          // corners rotated by +angle about the centre, c[4] = c[0]
};

__device__ __forceinline__ Box make_box(const float* __restrict__ b) {
  Box r;
  r.x1 = b[0]; r.y1 = b[1]; r.x2 = b[2]; r.y2 = b[3];
  const float ang = b[4];
  const float cx = (r.x1 + r.x2) / 2, cy = (r.y1 + r.y2) / 2;
  const float co = cosf(ang), si = sinf(ang);
  r.cin = cosf(-ang);
  r.sin_ = sinf(-ang);
  const float px[4] = {r.x1, r.x2, r.x2, r.x1}, py[4] = {r.y1, r.y1, r.y2, r.y2};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    r.c[k].x = (px[k] - cx) * co + (py[k] - cy) * si + cx;
    r.c[k].y = -(px[k] - cx) * si + (py[k] - cy) * co + cy;
  }
  r.c[4] = r.c[0];
  return r;
}

__device__ __forceinline__ bool contains(const Box& b, Pt p) {
  const float MARGIN = 1e-5f;
  const float cx = (b.x1 + b.x2) / 2, cy = (b.y1 + b.y2) / 2;
  const float rx = (p.x - cx) * b.cin + (p.y - cy) * b.sin_ + cx;
  const float ry = -(p.x - cx) * b.sin_ + (p.y - cy) * b.cin + cy;
  return rx > b.x1 - MARGIN && rx < b.x2 + MARGIN && ry > b.y1 - MARGIN && ry < b.y2 + MARGIN;
}

__device__ __forceinline__ bool seg_intersection(Pt p1, Pt p0, Pt q1, Pt q0, Pt& ans) {
  const bool rect = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                    fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
  if (!rect) return false;
  const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return false;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > EPS) {
    ans.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    ans.x = (b0 * c1 - b1 * c0) / D;
    ans.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

__device__ float overlap(const Box& A, const Box& B) {
  Pt poly[16];
  float key[16];
  int cnt = 0;
  float sx = 0.f, sy = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Pt p;
      if (seg_intersection(A.c[i + 1], A.c[i], B.c[j + 1], B.c[j], p)) {
        sx += p.x;
        sy += p.y;
        poly[cnt++] = p;
      }
    }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (contains(A, B.c[k])) {
      sx += B.c[k].x;
      sy += B.c[k].y;
      poly[cnt++] = B.c[k];
    }
    if (contains(B, A.c[k])) {
      sx += A.c[k].x;
      sy += A.c[k].y;
      poly[cnt++] = A.c[k];
    }
  }
  if (cnt < 3) return 0.f;  // the reference's loops produce area 0 here as well
  const float cx = sx / cnt, cy = sy / cnt;
  for (int i = 0; i < cnt; ++i) key[i] = atan2f(poly[i].y - cy, poly[i].x - cx);
  // the reference bubble-sorts with a strict '>' comparison on the same keys: a stable ascending sort
  for (int j = 0; j < cnt - 1; ++j)
    for (int i = 0; i < cnt - j - 1; ++i)
      if (key[i] > key[i + 1]) {
        const Pt tp = poly[i];
        poly[i] = poly[i + 1];
        poly[i + 1] = tp;
        const float tk = key[i];
        key[i] = key[i + 1];
        key[i + 1] = tk;
      }
  float area = 0.f;
  for (int k = 0; k < cnt - 1; ++k) {
    const float ux = poly[k].x - poly[0].x, uy = poly[k].y - poly[0].y;
    const float vx = poly[k + 1].x - poly[0].x, vy = poly[k + 1].y - poly[0].y;
    area += ux * vy - uy * vx;
  }
  return fabsf(area) / 2.0f;
}

__device__ __forceinline__ float iou_rotated(const Box& A, const Box& B) {
  const float sa = (A.x2 - A.x1) * (A.y2 - A.y1), sb = (B.x2 - B.x1) * (B.y2 - B.y1);
  const float so = overlap(A, B);
  return so / fmaxf(sa + sb - so, EPS);
}

__device__ __forceinline__ float iou_normal(const float* a, const float* b) {
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]), top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f), inter = w * h;
  const float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  return inter / fmaxf(sa + sb - inter, EPS);
}

// MODE 0: overlap area, 1: rotated IoU.  One wave per A box, lanes over 64 B boxes.
template <int MODE>
__global__ __launch_bounds__(256) void pairwise_kernel(const float* __restrict__ boxes_a, int num_a,
                                                       const float* __restrict__ boxes_b, int num_b, float* __restrict__ out) {
  const int a_idx = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b_idx = blockIdx.x * 64 + (threadIdx.x & 63);
  if (a_idx >= num_a || b_idx >= num_b) return;
  const Box A = make_box(boxes_a + (size_t)a_idx * 5);
  const Box B = make_box(boxes_b + (size_t)b_idx * 5);
  out[(size_t)a_idx * num_b + b_idx] = MODE == 0 ? overlap(A, B) : iou_rotated(A, B);
}

// suppression words of row block blockIdx.y against column block blockIdx.x (only column >= row is launched/used)
template <bool NORMAL>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thresh, int col_blocks,
                                                      unsigned long long* __restrict__ mask) {
  const int rb = blockIdx.y, cb = blockIdx.x;
  if (cb < rb) return;
  __shared__ float cols[64 * 5];
  const int lane = threadIdx.x;
  const int col_size = min(n - cb * 64, 64), row_size = min(n - rb * 64, 64);
  if (lane < col_size) {
#pragma unroll
    for (int f = 0; f < 5; ++f) cols[lane * 5 + f] = boxes[((size_t)cb * 64 + lane) * 5 + f];
  }
  __syncthreads();
  if (lane >= row_size) return;
  const int i = rb * 64 + lane;
  const float* mine = boxes + (size_t)i * 5;
  unsigned long long t = 0;
  const int start = rb == cb ? lane + 1 : 0;
  if (NORMAL) {
    for (int j = start; j < col_size; ++j)
      if (iou_normal(mine, cols + j * 5) > thresh) t |= 1ull << j;
  } else {
    const Box A = make_box(mine);
    for (int j = start; j < col_size; ++j) {
      const Box B = make_box(cols + j * 5);
      if (iou_rotated(A, B) > thresh) t |= 1ull << j;
    }
  }
  mask[(size_t)i * col_blocks + cb] = t;
}

// greedy sweep over the sorted boxes (iou3d.cpp:115-132) by one workgroup
__global__ __launch_bounds__(256) void nms_sweep_kernel(const unsigned long long* __restrict__ mask, int n, int col_blocks,
                                                        unsigned long long* __restrict__ removed /* [col_blocks], zeroed */,
                                                        long long* __restrict__ keep, int* __restrict__ num_out) {
  __shared__ unsigned long long diag[64];
  __shared__ unsigned long long kept_bits;
  __shared__ int count;
  if (threadIdx.x == 0) count = 0;
  for (int rb = 0; rb < col_blocks; ++rb) {
    const int rows = min(n - rb * 64, 64);
    if (threadIdx.x < 64) diag[threadIdx.x] = threadIdx.x < rows ? mask[((size_t)rb * 64 + threadIdx.x) * col_blocks + rb] : 0ull;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long alive = ~removed[rb];
      if (rows < 64) alive &= (1ull << rows) - 1ull;
      unsigned long long kept = 0;
      int c = count;
      for (int t = 0; t < rows; ++t)
        if ((alive >> t) & 1ull) {
          kept |= 1ull << t;
          alive &= ~diag[t];
          keep[c++] = (long long)rb * 64 + t;
        }
      count = c;
      kept_bits = kept;
    }
    __syncthreads();
    const unsigned long long kept = kept_bits;
    for (int j = rb + 1 + (int)threadIdx.x; j < col_blocks; j += 256) {
      unsigned long long acc = removed[j];
      unsigned long long k = kept;
      while (k) {
        const int t = __ffsll((long long)k) - 1;
        k &= k - 1;
        acc |= mask[((size_t)rb * 64 + t) * col_blocks + j];
      }
      removed[j] = acc;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) *num_out = count;
}

}  // namespace iou3d
}  // namespace bevamd

using namespace bevamd;

extern "C" {

static int pairwise(const float* a, int na, const float* b, int nb, float* out, int mode, hipStream_t stream) {
  BEVAMD_REQUIRE(na >= 0 && nb >= 0, "iou3d: negative box count");
  if (na == 0 || nb == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(a && b && out, "iou3d: null buffer");
  dim3 grid(cdiv(nb, 64), cdiv(na, 4)), block(256);
  if (mode == 0) iou3d::pairwise_kernel<0><<<grid, block, 0, stream>>>(a, na, b, nb, out);
  else iou3d::pairwise_kernel<1><<<grid, block, 0, stream>>>(a, na, b, nb, out);
  BEVAMD_LAUNCH_CHECK("iou3d_pairwise");
  return BEVAMD_OK;
}

int bevamd_iou3d_boxes_overlap_bev(const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* ans_overlap,
                                   void* stream) {
  return pairwise(boxes_a, num_a, boxes_b, num_b, ans_overlap, 0, (hipStream_t)stream);
}

int bevamd_iou3d_boxes_iou_bev(const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* ans_iou,
                               void* stream) {
  return pairwise(boxes_a, num_a, boxes_b, num_b, ans_iou, 1, (hipStream_t)stream);
}

size_t bevamd_iou3d_nms_workspace_bytes(int num_boxes) {
  if (num_boxes <= 0) return 256;
  const size_t cb = ((size_t)num_boxes + 63) / 64;
  return align_up((size_t)num_boxes * cb * 8, 256) + align_up(cb * 8, 256) + 256;
}

int bevamd_iou3d_nms(const float* boxes, int num_boxes, float thresh, int normal, long long* keep, int* num_out_dev,
                     int* num_out_host, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(num_boxes >= 0, "iou3d_nms: negative box count");
  BEVAMD_REQUIRE(num_out_dev != nullptr, "iou3d_nms: num_out_dev is null");
  if (num_boxes == 0) {
    int rc = device_fill_u32((uint32_t*)num_out_dev, 1, 0u, stream);
    if (rc) return rc;
    if (num_out_host) { BEVAMD_HIP_CHECK(hipStreamSynchronize(stream)); *num_out_host = 0; }
    return BEVAMD_OK;
  }
  BEVAMD_REQUIRE(boxes && keep, "iou3d_nms: null buffer");
  if (!ws || ws_bytes < bevamd_iou3d_nms_workspace_bytes(num_boxes)) {
    set_error("iou3d_nms: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  const int cb = (num_boxes + 63) / 64;
  BEVAMD_REQUIRE(cb <= 65535, "iou3d_nms: more than 4 M boxes");
  Carver cv(ws, ws_bytes);
  unsigned long long* mask = cv.take<unsigned long long>((size_t)num_boxes * cb);
  unsigned long long* removed = cv.take<unsigned long long>((size_t)cb);
  int rc = device_fill_u32((uint32_t*)removed, (size_t)cb * 2, 0u, stream);
  if (rc) return rc;
  dim3 grid(cb, cb), block(64);
  if (normal) iou3d::nms_mask_kernel<true><<<grid, block, 0, stream>>>(boxes, num_boxes, thresh, cb, mask);
  else iou3d::nms_mask_kernel<false><<<grid, block, 0, stream>>>(boxes, num_boxes, thresh, cb, mask);
  BEVAMD_LAUNCH_CHECK("iou3d_nms_mask");
  iou3d::nms_sweep_kernel<<<1, 256, 0, stream>>>(mask, num_boxes, cb, removed, keep, num_out_dev);
  BEVAMD_LAUNCH_CHECK("iou3d_nms_sweep");
  if (num_out_host) {
    BEVAMD_HIP_CHECK(hipMemcpyAsync(num_out_host, num_out_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
    BEVAMD_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return BEVAMD_OK;
}

}  // extern "C"
