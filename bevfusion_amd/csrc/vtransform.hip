// Device-side glue of the camera view transform for gfx950.
//
// depth raster — replaces the Python loops of BaseDepthTransform.forward
//   (reference: /root/reference/mmdet3d/models/vtransforms/base.py:283-329): for every sample the reference un-does the
//   LiDAR augmentation, projects all points into each of the 6 cameras, applies the image augmentation, truncates to a
//   pixel and writes the point's (clamped) depth with boolean indexing — dozens of small launches and one host sync per
//   camera.  Here: one thread per (point, camera); colliding points are resolved as "the LAST point in input order wins"
//   (what the reference's assignment gives on CPU; on GPU its index_put is unordered) with an atomicMax on the point
//   index followed by a second pass that lets only the winner write — deterministic, no sort.
//   Arithmetic follows the reference op by op in fp32 (subtract, 3x3 products as k-ordered fma chains, true division,
//   truncation toward zero like `.long()`); `dist` is the CLAMPED depth, because the reference's `dist` is a view of
//   the tensor it clamps in place (base.py:300-302).
//
// frustum geometry — BaseTransform.get_geometry (base.py:92-135): frustum (u, v, d) -> lidar frame, one thread per
//   frustum point and camera, same op order.  Static per calibration; it exists so that building a pooling plan for a new
//   calibration is two launches (this + bevamd_bev_pool_prepare_from_geom) instead of ~10 broadcasting matmuls.
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
// y = M x as torch's matmul evaluates a k = 3 product: one multiply, two fused multiply-adds, k ascending
__device__ __forceinline__ void mat3_apply(const Mat3& M, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = fmaf(M.m[2], z, fmaf(M.m[1], y, M.m[0] * x));
  oy = fmaf(M.m[5], z, fmaf(M.m[4], y, M.m[3] * x));
  oz = fmaf(M.m[8], z, fmaf(M.m[7], y, M.m[6] * x));
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

__global__ __launch_bounds__(256) void depth_raster_winner_kernel(RasterArgs a, int* __restrict__ winner) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)a.n * a.ncam) return;
  const int i = (int)(t / a.ncam), c = (int)(t - (long long)i * a.ncam);
  int row, col;
  float dist;
  if (project(a, i, c, row, col, dist)) atomicMax(&winner[((size_t)c * a.ih + row) * a.iw + col], i);
}

__global__ __launch_bounds__(256) void depth_raster_write_kernel(RasterArgs a, const int* __restrict__ winner,
                                                                 float* __restrict__ depth) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)a.n * a.ncam) return;
  const int i = (int)(t / a.ncam), c = (int)(t - (long long)i * a.ncam);
  int row, col;
  float dist;
  if (project(a, i, c, row, col, dist)) {
    const size_t pix = ((size_t)c * a.ih + row) * a.iw + col;
    if (winner[pix] == i) depth[pix] = dist;
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
  mat3_apply(load_mat3(a.post_rot_inv + (size_t)cam * 9, 3), x, y, z, u, v, w);
  u = u * w;   // (x*z, y*z, z)  base.py:110-116
  v = v * w;
  mat3_apply(load_mat3(a.combine + (size_t)cam * 9, 3), u, v, w, x, y, z);
  const float* ct = a.c2l_trans + (size_t)cam * 3;
  x += ct[0]; y += ct[1]; z += ct[2];
  const int b = cam / a.cams_per_sample;
  if (a.extra_rot) {
    mat3_apply(load_mat3(a.extra_rot + (size_t)b * 9, 3), x, y, z, u, v, w);
    x = u; y = v; z = w;
  }
  if (a.extra_trans) {
    const float* et = a.extra_trans + (size_t)b * 3;
    x += et[0]; y += et[1]; z += et[2];
  }
  float* o = geom + (size_t)t * 3;
  o[0] = x; o[1] = y; o[2] = z;
}

}  // namespace bevamd

using namespace bevamd;

extern "C" {

/* winner workspace of bevamd_depth_raster: one int32 per output pixel */
size_t bevamd_depth_raster_workspace_bytes(int ncam, int ih, int iw) {
  if (ncam <= 0 || ih <= 0 || iw <= 0) return 0;
  return align_up((size_t)ncam * ih * iw * sizeof(int), 256);
}

int bevamd_depth_raster(const float* points, int num_points, int num_features, const float* lidar_aug_inv_rot,
                        const float* lidar_aug_trans, const float* lidar2image, const float* img_aug, int ncam, int ih,
                        int iw, float* depth, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(num_points >= 0 && num_features >= 3 && ncam > 0 && ih > 0 && iw > 0, "depth_raster: bad sizes");
  BEVAMD_REQUIRE(depth != nullptr, "depth_raster: depth is null");
  const size_t npix = (size_t)ncam * ih * iw;
  int rc = device_fill_u32((uint32_t*)depth, npix, 0u, stream);  // reference: torch.zeros
  if (rc) return rc;
  if (num_points == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(points && lidar_aug_inv_rot && lidar_aug_trans && lidar2image && img_aug, "depth_raster: null input");
  if (!ws || ws_bytes < bevamd_depth_raster_workspace_bytes(ncam, ih, iw)) {
    set_error("depth_raster: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  int* winner = (int*)ws;
  rc = device_fill_u32((uint32_t*)winner, npix, 0xFFFFFFFFu, stream);  // -1
  if (rc) return rc;
  RasterArgs a{points, lidar_aug_inv_rot, lidar_aug_trans, lidar2image, img_aug, num_points, num_features, ncam, ih, iw};
  const long long total = (long long)num_points * ncam;
  dim3 grid(cdiv(total, 256)), block(256);
  depth_raster_winner_kernel<<<grid, block, 0, stream>>>(a, winner);
  BEVAMD_LAUNCH_CHECK("depth_raster_winner");
  depth_raster_write_kernel<<<grid, block, 0, stream>>>(a, winner, depth);
  BEVAMD_LAUNCH_CHECK("depth_raster_write");
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

}  // extern "C"
