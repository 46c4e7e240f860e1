// Device-side glue of the camera view transform for gfx950.
//
// depth raster — replaces the Python loops of BaseDepthTransform.forward
//   (reference: /root/reference/mmdet3d/models/vtransforms/base.py:283-329): for every sample the reference un-does the
//   LiDAR augmentation, projects all points into each of the 6 cameras, applies the image augmentation, truncates to a
//   pixel and writes the point's (clamped) depth with boolean indexing — dozens of small launches and one host sync per
//   camera.  Here: one thread per (point, camera); colliding points are resolved as "the LAST point in input order wins"
//   (what the reference's assignment gives on CPU; on GPU its index_put is unordered) with ONE 64-bit atomicMax per hit on
//   (point index + 1) << 32 | depth bits, then a pass over the pixels that unpacks the image — deterministic, no sort.
//   Arithmetic follows the reference op by op in fp32: subtract, the three GEMMs ([3,3]x[3,n], [N,3,3]x[3,n],
//   [N,3,3]x[N,3,n] -> BLAS sgemm) as k-ascending chains of FUSED multiply-adds, true division, truncation toward zero like
//   `.long()`; `dist` is the CLAMPED depth, because the reference's `dist` is a view of the tensor it clamps in place
//   (base.py:300-302).  Bit-exact against tests/golden/vtransform_ref.npz (the reference's own function body on CPU torch).
//
// frustum geometry — BaseTransform.get_geometry (base.py:92-135): frustum (u, v, d) -> lidar frame, one thread per
//   frustum point and camera, same op order.  Its 3x3 products are per-point broadcast bmm's (ATen's naive kernel:
//   acc = 0; acc += a_k * b_k, k ascending, product and sum rounded SEPARATELY) -> `#pragma clang fp contract(off)` here, never
//   contracted to an fma.  Bit-exact against the same fixture at the flagship size (SHA-256 of the 24 MB result).
//   Static per calibration; it exists so that building a pooling plan for a new calibration is two launches (this +
//   bevamd_bev_pool_prepare_from_geom) instead of ~10 broadcasting matmuls.
//
// camera matrices — the 3x3 inverses in front of both (torch.inverse = LAPACK getrf/getrs in the reference, third-party
//   arithmetic that is not even layout-stable on CPU): `mat3_inverse_kernel` evaluates adjugate / determinant in fp64 and
//   rounds once to fp32 (within 1-2 ulp of any fp32 LU), one thread per matrix, so that a new calibration costs no
//   host-side LAPACK call and no host sync.
#include "common.h"

namespace bevamd {

struct Mat3 { float m[9]; };

__device__ __forceinline__ Mat3 load_mat3(const float* __restrict__ p, int row_stride) {
  Mat3 r;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = p[i * row_stride + j];
  return r;
}
// y = M x as a BLAS sgemm evaluates a k = 3 product: one multiply, two fused multiply-adds, k ascending
__device__ __forceinline__ void mat3_apply(const Mat3& M, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = fmaf(M.m[2], z, fmaf(M.m[1], y, M.m[0] * x));
  oy = fmaf(M.m[5], z, fmaf(M.m[4], y, M.m[3] * x));
  oz = fmaf(M.m[8], z, fmaf(M.m[7], y, M.m[6] * x));
}
// y = M x as ATen's naive bmm kernel evaluates it: every product and every sum rounded on its own (0 + p is exact).
// HIP's __fmul_rn / __fadd_rn are plain operators that hipcc's default -ffp-contract=fast-honor-pragmas may still fuse:
// the pragma is what keeps these a v_mul_f32 + v_add_f32 pair (checked in the ISA: no v_fma / v_fmac in lss_geometry_kernel).
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float dot3_rn(float a0, float a1, float a2, float x, float y, float z) {
#pragma clang fp contract(off)
  const float p0 = a0 * x, p1 = a1 * y, p2 = a2 * z;
  const float s = p0 + p1;
  return s + p2;
}
__device__ __forceinline__ void mat3_apply_rn(const Mat3& M, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = dot3_rn(M.m[0], M.m[1], M.m[2], x, y, z);
  oy = dot3_rn(M.m[3], M.m[4], M.m[5], x, y, z);
  oz = dot3_rn(M.m[6], M.m[7], M.m[8], x, y, z);
}

struct RasterArgs {
  const float* points;       // [n, nfeat] xyz first
  const float* aug_inv_rot;  // [3, 3]  inverse(lidar_aug_matrix[:3, :3])
  const float* aug_trans;    // [3]     lidar_aug_matrix[:3, 3]
  const float* lidar2image;  // [ncam, 4, 4]
  const float* img_aug;      // [ncam, 4, 4]
  int n, nfeat, ncam, ih, iw;
};

// pixel (row, col) and depth of point i in camera c; false if it falls outside the image
__device__ __forceinline__ bool project(const RasterArgs& a, int i, int c, int& row, int& col, float& dist) {
  const float* p = a.points + (size_t)i * a.nfeat;
  float x = p[0] - a.aug_trans[0], y = p[1] - a.aug_trans[1], z = p[2] - a.aug_trans[2];
  float u, v, w;
  mat3_apply(load_mat3(a.aug_inv_rot, 3), x, y, z, u, v, w);
  const float* l2i = a.lidar2image + (size_t)c * 16;
  mat3_apply(load_mat3(l2i, 4), u, v, w, x, y, z);
  x += l2i[3]; y += l2i[7]; z += l2i[11];
  z = fminf(fmaxf(z, 1e-5f), 1e5f);
  dist = z;
  x = x / z;
  y = y / z;
  const float* ia = a.img_aug + (size_t)c * 16;
  mat3_apply(load_mat3(ia, 4), x, y, z, u, v, w);
  u += ia[3]; v += ia[7];
  // (row, col) = (v, u); on-image test on the float values, then truncation (base.py:311-317)
  if (!(v < (float)a.ih && v >= 0.f && u < (float)a.iw && u >= 0.f)) return false;
  row = (int)v;
  col = (int)u;
  return true;
}

// Collisions resolve to the LAST point in input order (deterministic; the reference's GPU index_put is unordered).  One pass: a
// pixel holds (point index + 1) << 32 | depth bits and takes the 64-bit maximum, so the winner carries its depth with it and
// nothing is projected twice (rounds 1-2: atomicMax on the index, then a second pass over every (point, camera) pair that
// re-projected it to write the winner's depth: 106 of 230 us at 8 frames); a pass over the PIXELS then unpacks the image
// (0 where nothing landed — the reference's torch.zeros).
__device__ __forceinline__ unsigned long long raster_pack(int i, float dist) {
  return ((unsigned long long)(unsigned)(i + 1) << 32) | (unsigned long long)__float_as_uint(dist);
}

__global__ __launch_bounds__(256) void depth_raster_packed_kernel(RasterArgs a, unsigned long long* __restrict__ packed) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)a.n * a.ncam) return;
  const int i = (int)(t / a.ncam), c = (int)(t - (long long)i * a.ncam);
  int row, col;
  float dist;
  if (project(a, i, c, row, col, dist)) atomicMax(&packed[((size_t)c * a.ih + row) * a.iw + col], raster_pack(i, dist));
}

__global__ __launch_bounds__(256) void depth_raster_unpack_kernel(const unsigned long long* __restrict__ packed, size_t npix,
                                                                  float* __restrict__ depth) {
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256)
    depth[p] = __uint_as_float((unsigned)(packed[p] & 0xFFFFFFFFull));   // empty pixel: 0 | 0 -> 0.0f
}
// the same, leaving the map zero behind it: the next raster over the same (persistent) map needs no fill launch
__global__ __launch_bounds__(256) void depth_raster_unpack_clear_kernel(unsigned long long* __restrict__ packed, size_t npix,
                                                                        float* __restrict__ depth) {
  for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < npix; p += (size_t)gridDim.x * 256) {
    const unsigned long long w = packed[p];
    depth[p] = __uint_as_float((unsigned)(w & 0xFFFFFFFFull));
    if (w) packed[p] = 0ull;
  }
}

// ---- batched raster: up to RASTER_MAX_BATCH samples per launch pair (pointers by value in the kernel argument) ---------------
constexpr int RASTER_MAX_BATCH = 16;
struct RasterBatch {
  const float* points[RASTER_MAX_BATCH];
  int n[RASTER_MAX_BATCH];
  int start[RASTER_MAX_BATCH + 1];   // exclusive prefix of n[] (thread -> sample by a short scan)
  const float* aug_inv_rot;          // [B, 3, 3]
  const float* aug_trans;            // [B, 3] (row b at b * trans_stride)
  const float* lidar2image;          // [B, ncam, 4, 4]
  const float* img_aug;              // [B, ncam, 4, 4]
  int batch, nfeat, ncam, ih, iw, trans_stride;
};

__device__ __forceinline__ bool raster_batch_locate(const RasterBatch& rb, long long t, RasterArgs& a, int& b, int& i, int& c) {
  const long long pt = t / rb.ncam;
  c = (int)(t - pt * rb.ncam);
  if (pt >= rb.start[rb.batch]) return false;
  b = 0;
#pragma unroll 1
  while (b + 1 < rb.batch && pt >= rb.start[b + 1]) ++b;
  i = (int)(pt - rb.start[b]);
  a.points = rb.points[b];
  a.aug_inv_rot = rb.aug_inv_rot + (size_t)b * 9;
  a.aug_trans = rb.aug_trans + (size_t)b * rb.trans_stride;
  a.lidar2image = rb.lidar2image + (size_t)b * rb.ncam * 16;
  a.img_aug = rb.img_aug + (size_t)b * rb.ncam * 16;
  a.n = rb.n[b]; a.nfeat = rb.nfeat; a.ncam = rb.ncam; a.ih = rb.ih; a.iw = rb.iw;
  return true;
}

__global__ __launch_bounds__(256) void depth_raster_batch_packed_kernel(RasterBatch rb, unsigned long long* __restrict__ packed) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  RasterArgs a;
  int b, i, c, row, col;
  float dist;
  if (!raster_batch_locate(rb, t, a, b, i, c)) return;
  if (project(a, i, c, row, col, dist))
    atomicMax(&packed[(((size_t)b * a.ncam + c) * a.ih + row) * a.iw + col], raster_pack(i, dist));
}

// The same with one thread per POINT and blockIdx.y = sample: the thread walks the cameras, so sample and camera are uniform across
// the wave and the ~45 matrix words of a projection come through scalar loads instead of 45 per-lane loads per (point, camera)
// pair (125 us per 8 x 310 k points x 6 cameras were mostly those loads).  Same arithmetic per pair; the 64-bit maximum does
// not depend on the order the pairs arrive in.
__global__ __launch_bounds__(256) void depth_raster_batch_points_kernel(RasterBatch rb, unsigned long long* __restrict__ packed) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rb.n[b]) return;
  RasterArgs a;
  a.points = rb.points[b];
  a.aug_inv_rot = rb.aug_inv_rot + (size_t)b * 9;
  a.aug_trans = rb.aug_trans + (size_t)b * rb.trans_stride;
  a.lidar2image = rb.lidar2image + (size_t)b * rb.ncam * 16;
  a.img_aug = rb.img_aug + (size_t)b * rb.ncam * 16;
  a.n = rb.n[b]; a.nfeat = rb.nfeat; a.ncam = rb.ncam; a.ih = rb.ih; a.iw = rb.iw;
  for (int c = 0; c < rb.ncam; ++c) {
    int row, col;
    float dist;
    if (project(a, i, c, row, col, dist))
      atomicMax(&packed[(((size_t)b * a.ncam + c) * a.ih + row) * a.iw + col], raster_pack(i, dist));
  }
}

struct GeomArgs {
  const float* frustum;        // [D*fH*fW, 3] (u, v, d)
  const float* post_rot_inv;   // [ncam_total, 3, 3]  inverse(img_aug[:3,:3])
  const float* post_trans;     // [ncam_total, 3]
  const float* combine;        // [ncam_total, 3, 3]  camera2lidar_rot @ inverse(intrinsics)
  const float* c2l_trans;      // [ncam_total, 3]
  const float* extra_rot;      // [batch, 3, 3] or null
  const float* extra_trans;    // [batch, 3]    or null
  int npts, ncam_total, cams_per_sample;
};

__global__ __launch_bounds__(256) void lss_geometry_kernel(GeomArgs a, float* __restrict__ geom) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)a.npts * a.ncam_total) return;
  const int cam = (int)(t / a.npts), j = (int)(t - (long long)cam * a.npts);
  const float* f = a.frustum + (size_t)j * 3;
  const float* pt = a.post_trans + (size_t)cam * 3;
  float x = f[0] - pt[0], y = f[1] - pt[1], z = f[2] - pt[2];
  float u, v, w;
  mat3_apply_rn(load_mat3(a.post_rot_inv + (size_t)cam * 9, 3), x, y, z, u, v, w);
  u = mul_rn(u, w);   // (x*z, y*z, z)  base.py:110-116
  v = mul_rn(v, w);
  mat3_apply_rn(load_mat3(a.combine + (size_t)cam * 9, 3), u, v, w, x, y, z);
  const float* ct = a.c2l_trans + (size_t)cam * 3;
  x = add_rn(x, ct[0]); y = add_rn(y, ct[1]); z = add_rn(z, ct[2]);
  const int b = cam / a.cams_per_sample;
  if (a.extra_rot) {
    mat3_apply_rn(load_mat3(a.extra_rot + (size_t)b * 9, 3), x, y, z, u, v, w);
    x = u; y = v; z = w;
  }
  if (a.extra_trans) {
    const float* et = a.extra_trans + (size_t)b * 3;
    x = add_rn(x, et[0]); y = add_rn(y, et[1]); z = add_rn(z, et[2]);
  }
  float* o = geom + (size_t)t * 3;
  o[0] = x; o[1] = y; o[2] = z;
}

// ---- per-camera matrices ------------------------------------------------------------------------------------------------
// out[i] = inverse(m[i]) for `count` 3x3 matrices addressed m + i*mat_stride + r*row_stride + c (so the top-left block of a
// 4x4 is mat_stride 16, row_stride 4): cofactors and determinant in fp64 (exact products of fp32 inputs, <= 3 roundings per
// entry), one rounding to fp32.
__device__ __forceinline__ void mat3_inverse_f64(const float* __restrict__ m, long long row_stride, float* __restrict__ o) {
  double a[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) a[r * 3 + c] = (double)m[r * row_stride + c];
  const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02;
  const double r = 1.0 / det;
  o[0] = (float)(c00 * r);
  o[1] = (float)((a[2] * a[7] - a[1] * a[8]) * r);
  o[2] = (float)((a[1] * a[5] - a[2] * a[4]) * r);
  o[3] = (float)(c01 * r);
  o[4] = (float)((a[0] * a[8] - a[2] * a[6]) * r);
  o[5] = (float)((a[2] * a[3] - a[0] * a[5]) * r);
  o[6] = (float)(c02 * r);
  o[7] = (float)((a[1] * a[6] - a[0] * a[7]) * r);
  o[8] = (float)((a[0] * a[4] - a[1] * a[3]) * r);
}

// + the fourth column of the same rows packed as [count, 3] (rotation inverse and translation of a [B, 4, 4] augmentation matrix
// in one launch: a strided-copy kernel of the framework otherwise)
__global__ void mat3_inverse_col_kernel(const float* __restrict__ m, long long mat_stride, long long row_stride, int count,
                                        float* __restrict__ out, float* __restrict__ col) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  mat3_inverse_f64(m + (size_t)i * mat_stride, row_stride, out + (size_t)i * 9);
  for (int r = 0; r < 3; ++r) col[(size_t)i * 3 + r] = m[(size_t)i * mat_stride + r * row_stride + 3];
}

__global__ void mat3_inverse_kernel(const float* __restrict__ m, long long mat_stride, long long row_stride, int count,
                                    float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) mat3_inverse_f64(m + (size_t)i * mat_stride, row_stride, out + (size_t)i * 9);
}

// post_rot_inv[i] = inverse(post_rots[i]); combine[i] = c2l_rots[i] @ inverse(intrins[i]) (base.py:106, 118; the product
// in the naive-bmm order: acc = 0; acc += a[r][k] * b[k][c], k ascending, separate roundings)
__global__ void lss_camera_matrices_kernel(const float* __restrict__ post_rots, const float* __restrict__ c2l_rots,
                                           const float* __restrict__ intrins, long long mat_stride, long long row_stride,
                                           int count, float* __restrict__ post_rot_inv, float* __restrict__ combine) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  mat3_inverse_f64(post_rots + (size_t)i * mat_stride, row_stride, post_rot_inv + (size_t)i * 9);
  float ik[9];
  mat3_inverse_f64(intrins + (size_t)i * mat_stride, row_stride, ik);
  const float* R = c2l_rots + (size_t)i * mat_stride;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      combine[(size_t)i * 9 + r * 3 + c] =
          dot3_rn(R[r * row_stride + 0], R[r * row_stride + 1], R[r * row_stride + 2], ik[c], ik[3 + c], ik[6 + c]);
}

}  // namespace bevamd

using namespace bevamd;

extern "C" {

/* workspace of bevamd_depth_raster: one u64 per output pixel, (winning point index + 1) << 32 | depth bits, 0 = no hit */
size_t bevamd_depth_raster_workspace_bytes(int ncam, int ih, int iw) {
  if (ncam <= 0 || ih <= 0 || iw <= 0) return 0;
  return align_up((size_t)ncam * ih * iw * sizeof(unsigned long long), 256);
}

int bevamd_depth_raster(const float* points, int num_points, int num_features, const float* lidar_aug_inv_rot,
                        const float* lidar_aug_trans, const float* lidar2image, const float* img_aug, int ncam, int ih,
                        int iw, float* depth, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(num_points >= 0 && num_features >= 3 && ncam > 0 && ih > 0 && iw > 0, "depth_raster: bad sizes");
  BEVAMD_REQUIRE(depth != nullptr, "depth_raster: depth is null");
  const size_t npix = (size_t)ncam * ih * iw;
  int rc = BEVAMD_OK;
  if (num_points == 0) return device_fill_u32((uint32_t*)depth, npix, 0u, stream);  // reference: torch.zeros
  BEVAMD_REQUIRE(points && lidar_aug_inv_rot && lidar_aug_trans && lidar2image && img_aug, "depth_raster: null input");
  if (!ws || ws_bytes < bevamd_depth_raster_workspace_bytes(ncam, ih, iw)) {
    set_error("depth_raster: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  unsigned long long* packed = (unsigned long long*)ws;
  rc = device_fill_u32((uint32_t*)packed, npix * 2, 0u, stream);
  if (rc) return rc;
  RasterArgs a{points, lidar_aug_inv_rot, lidar_aug_trans, lidar2image, img_aug, num_points, num_features, ncam, ih, iw};
  const long long total = (long long)num_points * ncam;
  dim3 grid(cdiv(total, 256)), block(256);
  depth_raster_packed_kernel<<<grid, block, 0, stream>>>(a, packed);
  BEVAMD_LAUNCH_CHECK("depth_raster_packed");
  const size_t ub = (npix + 255) / 256;
  depth_raster_unpack_kernel<<<dim3((unsigned)(ub < 8192 ? ub : 8192)), block, 0, stream>>>(packed, npix, depth);
  BEVAMD_LAUNCH_CHECK("depth_raster_unpack");
  return BEVAMD_OK;
}

/* bevamd_depth_raster for a whole batch in one launch pair (2 fills + 2 kernels per <= 16 samples instead of 4 launches per
 * sample): points[b] DEVICE pointers (host array), num_points[b] host ints; lidar_aug_inv_rot [B,3,3], lidar_aug_trans rows at
 * lidar_aug_trans + b * trans_stride floats (trans_stride 3 for a packed [B,3], 16 for column 3 of a [B,4,4] starting at
 * element 3 with row stride 4 is NOT contiguous -> pass a packed copy), lidar2image / img_aug [B,ncam,4,4]; depth [B,ncam,1,ih,iw];
 * ws: batch * bevamd_depth_raster_workspace_bytes(ncam, ih, iw).  Same arithmetic and collision rule per sample. */
static int depth_raster_batch(const float* const* points, const int* num_points, int batch, int num_features,
                              const float* lidar_aug_inv_rot, const float* lidar_aug_trans, int trans_stride,
                              const float* lidar2image, const float* img_aug, int ncam, int ih, int iw, float* depth, void* ws,
                              size_t ws_bytes, void* stream_, bool ws_is_zero);

int bevamd_depth_raster_batch(const float* const* points, const int* num_points, int batch, int num_features,
                              const float* lidar_aug_inv_rot, const float* lidar_aug_trans, int trans_stride,
                              const float* lidar2image, const float* img_aug, int ncam, int ih, int iw, float* depth, void* ws,
                              size_t ws_bytes, void* stream_) {
  return depth_raster_batch(points, num_points, batch, num_features, lidar_aug_inv_rot, lidar_aug_trans, trans_stride, lidar2image,
                            img_aug, ncam, ih, iw, depth, ws, ws_bytes, stream_, false);
}

/* bevamd_depth_raster_batch over a PERSISTENT map: ws must be all zero when the call starts and is all zero again when its kernels
 * have run (the unpack pass clears what it reads) — two launches instead of three, for a caller that keeps one zero-initialised
 * workspace per (device, size) and serialises its rasters on one stream. */
int bevamd_depth_raster_batch_zero_ws(const float* const* points, const int* num_points, int batch, int num_features,
                                      const float* lidar_aug_inv_rot, const float* lidar_aug_trans, int trans_stride,
                                      const float* lidar2image, const float* img_aug, int ncam, int ih, int iw, float* depth,
                                      void* ws, size_t ws_bytes, void* stream_) {
  return depth_raster_batch(points, num_points, batch, num_features, lidar_aug_inv_rot, lidar_aug_trans, trans_stride, lidar2image,
                            img_aug, ncam, ih, iw, depth, ws, ws_bytes, stream_, true);
}

static int depth_raster_batch(const float* const* points, const int* num_points, int batch, int num_features,
                              const float* lidar_aug_inv_rot, const float* lidar_aug_trans, int trans_stride,
                              const float* lidar2image, const float* img_aug, int ncam, int ih, int iw, float* depth, void* ws,
                              size_t ws_bytes, void* stream_, bool ws_is_zero) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(batch > 0 && num_features >= 3 && ncam > 0 && ih > 0 && iw > 0 && trans_stride >= 3, "depth_raster_batch: bad sizes");
  BEVAMD_REQUIRE(points && num_points && depth && lidar_aug_inv_rot && lidar_aug_trans && lidar2image && img_aug,
                 "depth_raster_batch: null buffer");
  const size_t per = (size_t)ncam * ih * iw;
  if (!ws || ws_bytes < (size_t)batch * bevamd_depth_raster_workspace_bytes(ncam, ih, iw)) {
    set_error("depth_raster_batch: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  unsigned long long* packed = (unsigned long long*)ws;
  if (!ws_is_zero) {
    int rc = device_fill_u32((uint32_t*)packed, per * batch * 2, 0u, stream);
    if (rc) return rc;
  }
  for (int b0 = 0; b0 < batch; b0 += RASTER_MAX_BATCH) {
    RasterBatch rb;
    rb.batch = batch - b0 < RASTER_MAX_BATCH ? batch - b0 : RASTER_MAX_BATCH;
    rb.start[0] = 0;
    for (int j = 0; j < rb.batch; ++j) {
      BEVAMD_REQUIRE(num_points[b0 + j] >= 0 && (num_points[b0 + j] == 0 || points[b0 + j]), "depth_raster_batch: bad sample %d", b0 + j);
      rb.points[j] = points[b0 + j];
      rb.n[j] = num_points[b0 + j];
      rb.start[j + 1] = rb.start[j] + rb.n[j];
    }
    for (int j = rb.batch; j < RASTER_MAX_BATCH; ++j) { rb.points[j] = nullptr; rb.n[j] = 0; rb.start[j + 1] = rb.start[rb.batch]; }
    rb.aug_inv_rot = lidar_aug_inv_rot + (size_t)b0 * 9;
    rb.aug_trans = lidar_aug_trans + (size_t)b0 * trans_stride;
    rb.lidar2image = lidar2image + (size_t)b0 * ncam * 16;
    rb.img_aug = img_aug + (size_t)b0 * ncam * 16;
    rb.nfeat = num_features; rb.ncam = ncam; rb.ih = ih; rb.iw = iw; rb.trans_stride = trans_stride;
    const long long total = (long long)rb.start[rb.batch] * ncam;
    if (total == 0) continue;
    static const bool by_pairs = getenv("BEVAMD_RASTER_BY_PAIRS") && getenv("BEVAMD_RASTER_BY_PAIRS")[0] == '1';   // tuning: rounds 2-3
    if (by_pairs) {
      dim3 grid(cdiv(total, 256)), block(256);
      depth_raster_batch_packed_kernel<<<grid, block, 0, stream>>>(rb, packed + (size_t)b0 * per);
    } else {
      int nmax = 0;
      for (int j = 0; j < rb.batch; ++j) nmax = rb.n[j] > nmax ? rb.n[j] : nmax;
      dim3 grid(cdiv(nmax, 256), rb.batch), block(256);
      depth_raster_batch_points_kernel<<<grid, block, 0, stream>>>(rb, packed + (size_t)b0 * per);
    }
    BEVAMD_LAUNCH_CHECK("depth_raster_batch_packed");
  }
  const size_t npix = per * batch, ub = (npix + 255) / 256;
  // every pixel written: no zero fill of `depth`
  if (ws_is_zero) depth_raster_unpack_clear_kernel<<<dim3((unsigned)(ub < 8192 ? ub : 8192)), dim3(256), 0, stream>>>(packed, npix, depth);
  else depth_raster_unpack_kernel<<<dim3((unsigned)(ub < 8192 ? ub : 8192)), dim3(256), 0, stream>>>(packed, npix, depth);
  BEVAMD_LAUNCH_CHECK("depth_raster_unpack");
  return BEVAMD_OK;
}

int bevamd_lss_geometry(const float* frustum, int frustum_points, const float* post_rot_inv, const float* post_trans,
                        const float* combine, const float* camera2lidar_trans, const float* extra_rot,
                        const float* extra_trans, int batch_size, int cams_per_sample, float* geom, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(frustum_points > 0 && batch_size > 0 && cams_per_sample > 0, "lss_geometry: bad sizes");
  BEVAMD_REQUIRE(frustum && post_rot_inv && post_trans && combine && camera2lidar_trans && geom, "lss_geometry: null buffer");
  GeomArgs a{frustum, post_rot_inv, post_trans, combine, camera2lidar_trans, extra_rot, extra_trans, frustum_points,
             batch_size * cams_per_sample, cams_per_sample};
  const long long total = (long long)frustum_points * a.ncam_total;
  lss_geometry_kernel<<<dim3(cdiv(total, 256)), dim3(256), 0, stream>>>(a, geom);
  BEVAMD_LAUNCH_CHECK("lss_geometry");
  return BEVAMD_OK;
}

/* out[i] = inverse of the 3x3 at m + i*mat_stride + r*row_stride + c, i < count (fp64 adjugate / determinant, one rounding).
 * Replaces torch.inverse at base.py:106, 118, 292 on the device path; no workspace, no host sync. */
int bevamd_mat3_inverse(const float* m, long long mat_stride, long long row_stride, int count, float* out, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(count >= 0 && row_stride >= 3 && mat_stride >= 0, "mat3_inverse: bad sizes");
  if (count == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(m && out, "mat3_inverse: null buffer");
  mat3_inverse_kernel<<<dim3(cdiv(count, 64)), dim3(64), 0, stream>>>(m, mat_stride, row_stride, count, out);
  BEVAMD_LAUNCH_CHECK("mat3_inverse");
  return BEVAMD_OK;
}

/* bevamd_mat3_inverse + col[i][r] = m[i*mat_stride + r*row_stride + 3] (row_stride >= 4): the inverse rotation and the translation
 * column of [count, 4, 4] matrices (base.py:289-292) from one launch. */
int bevamd_mat3_inverse_with_column(const float* m, long long mat_stride, long long row_stride, int count, float* out, float* col,
                                    void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(count >= 0 && row_stride >= 4 && mat_stride >= 0, "mat3_inverse_with_column: bad sizes");
  if (count == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(m && out && col, "mat3_inverse_with_column: null buffer");
  mat3_inverse_col_kernel<<<dim3(cdiv(count, 64)), dim3(64), 0, stream>>>(m, mat_stride, row_stride, count, out, col);
  BEVAMD_LAUNCH_CHECK("mat3_inverse_col");
  return BEVAMD_OK;
}

/* The two per-camera matrices bevamd_lss_geometry consumes, from the raw calibration (all three inputs share one
 * (mat_stride, row_stride) addressing, e.g. 16 / 4 for the top-left blocks of [B, N, 4, 4] tensors):
 * post_rot_inv[i] = inverse(post_rots[i]), combine[i] = camera2lidar_rots[i] @ inverse(intrins[i]), i < ncam_total. */
int bevamd_lss_camera_matrices(const float* post_rots, const float* camera2lidar_rots, const float* intrins,
                               long long mat_stride, long long row_stride, int ncam_total, float* post_rot_inv,
                               float* combine, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(ncam_total > 0 && row_stride >= 3 && mat_stride >= 0, "lss_camera_matrices: bad sizes");
  BEVAMD_REQUIRE(post_rots && camera2lidar_rots && intrins && post_rot_inv && combine, "lss_camera_matrices: null buffer");
  lss_camera_matrices_kernel<<<dim3(cdiv(ncam_total, 64)), dim3(64), 0, stream>>>(post_rots, camera2lidar_rots, intrins, mat_stride,
                                                                           row_stride, ncam_total, post_rot_inv, combine);
  BEVAMD_LAUNCH_CHECK("lss_camera_matrices");
  return BEVAMD_OK;
}

}  // extern "C"
