/* ORACLE — test infrastructure only (see oracle/__init__.py).
 *
 * View-transform glue of the camera branch, restated op by op in fp32 from the reference's Python
 * (/root/reference/mmdet3d/models/vtransforms/base.py), with the rounding behaviour of the torch CPU kernels those lines
 * dispatch to — established by running the reference's own function bodies in the build container
 * (tests/golden/make_vtransform_golden.py -> tests/golden/vtransform_ref.npz) and pinned bit for bit by
 * tests/test_oracle_vtransform.py:
 *
 *   get_geometry (base.py:92-135): every `[.., 3, 3].matmul([.., 3, 1])` is a broadcast bmm with 9 MACs per batch entry;
 *     ATen runs those through its naive `baddbmm_cpu_kernel` (contraction * rows * cols < 400): acc = 0; acc += a_k * b_k
 *     for k ascending, product and sum rounded SEPARATELY (no FMA in that translation unit).
 *   depth raster (base.py:283-329): `[3,3].matmul([3,n])` and `[N,3,3].matmul([.., 3, n])` are real GEMMs (MKL sgemm /
 *     sgemm_batch): k ascending chains of FUSED multiply-adds, fma(a2,b2, fma(a1,b1, a0*b0)).
 *   `torch.inverse` (LAPACK getrf/getrs inside MKL, not vendored under /root/reference) is NOT restated: callers pass the
 *     inverse matrices in (the golden fixture records the ones the reference computed).
 *
 * This file is compiled with -ffp-contract=off: `a * b + c` below is two roundings, fmaf() is one.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* y = M x, separate roundings, k ascending, accumulator starts at +0 (0 + p is exact) */
static void mat3_muladd(const float* M, float x, float y, float z, float* o) {
  for (int i = 0; i < 3; ++i) {
    float acc = 0.0f;
    acc = acc + M[i * 3 + 0] * x;
    acc = acc + M[i * 3 + 1] * y;
    acc = acc + M[i * 3 + 2] * z;
    o[i] = acc;
  }
}
/* y = M x as a k-ascending fma chain (row stride rs: 3 for a packed 3x3, 4 for the top-left block of a 4x4) */
static void mat3_fma(const float* M, int rs, float x, float y, float z, float* o) {
  for (int i = 0; i < 3; ++i) o[i] = fmaf(M[i * rs + 2], z, fmaf(M[i * rs + 1], y, M[i * rs + 0] * x));
}

/* base.py:92-135.  frustum [npts,3]; per camera (cam = b*N + n): inv_post_rots [3,3], post_trans [3], combine [3,3],
 * c2l_trans [3]; per sample: extra_rots [3,3] / extra_trans [3] (either may be NULL) -> geom [B*N, npts, 3]. */
void oracle_lss_geometry(const float* frustum, int64_t npts, const float* inv_post_rots, const float* post_trans,
                         const float* combine, const float* c2l_trans, const float* extra_rots, const float* extra_trans,
                         int64_t B, int64_t N, float* geom) {
  for (int64_t cam = 0; cam < B * N; ++cam) {
    const int64_t b = cam / N;
    const float* pt = post_trans + cam * 3;
    const float* ct = c2l_trans + cam * 3;
    for (int64_t j = 0; j < npts; ++j) {
      const float* f = frustum + j * 3;
      float p[3], q[3];
      mat3_muladd(inv_post_rots + cam * 9, f[0] - pt[0], f[1] - pt[1], f[2] - pt[2], p);   /* :104-109 */
      p[0] = p[0] * p[2];                                                                   /* :111-117 */
      p[1] = p[1] * p[2];
      mat3_muladd(combine + cam * 9, p[0], p[1], p[2], q);                                  /* :118-119 */
      q[0] += ct[0]; q[1] += ct[1]; q[2] += ct[2];                                          /* :120 */
      if (extra_rots) {                                                                     /* :122-128 */
        mat3_muladd(extra_rots + b * 9, q[0], q[1], q[2], p);
        q[0] = p[0]; q[1] = p[1]; q[2] = p[2];
      }
      if (extra_trans) {                                                                    /* :129-133 */
        q[0] += extra_trans[b * 3 + 0]; q[1] += extra_trans[b * 3 + 1]; q[2] += extra_trans[b * 3 + 2];
      }
      float* o = geom + (cam * npts + j) * 3;
      o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
    }
  }
}

/* One sample of base.py:283-329 (scalar depth, no extra features).  points [n, nfeat]; inv_aug_rot [3,3] =
 * inverse(lidar_aug_matrix[:3,:3]); aug_trans [3]; lidar2image / img_aug [ncam,4,4].
 * depth [ncam, ih, iw] (zero-filled here), winner [ncam, ih, iw] = index of the point that was written last (-1 = none):
 * sequential assignment in input order, i.e. the LAST point projecting into a pixel stays. */
void oracle_depth_raster(const float* points, int64_t n, int64_t nfeat, const float* inv_aug_rot, const float* aug_trans,
                         const float* lidar2image, const float* img_aug, int64_t ncam, int64_t ih, int64_t iw,
                         float* depth, int32_t* winner) {
  memset(depth, 0, sizeof(float) * (size_t)(ncam * ih * iw));
  for (int64_t i = 0; i < ncam * ih * iw; ++i) winner[i] = -1;
  for (int64_t c = 0; c < ncam; ++c) {
    const float* l2i = lidar2image + c * 16;
    const float* ia = img_aug + c * 16;
    for (int64_t i = 0; i < n; ++i) {
      const float* p = points + i * nfeat;
      float a[3], q[3], r[3];
      mat3_fma(inv_aug_rot, 3, p[0] - aug_trans[0], p[1] - aug_trans[1], p[2] - aug_trans[2], a);   /* :291-294 */
      mat3_fma(l2i, 4, a[0], a[1], a[2], q);                                                          /* :296 */
      q[0] += l2i[3]; q[1] += l2i[7]; q[2] += l2i[11];                                                /* :297 */
      float z = q[2];                                                                                  /* :299-301 */
      z = z < 1e-5f ? 1e-5f : (z > 1e5f ? 1e5f : z);
      const float dist = z;  /* `dist` is a view of the clamped row */
      q[0] = q[0] / z;                                                                                 /* :302 */
      q[1] = q[1] / z;
      mat3_fma(ia, 4, q[0], q[1], z, r);                                                               /* :305 */
      const float u = r[0] + ia[3], v = r[1] + ia[7];                                                  /* :306 */
      if (!(v < (float)ih && v >= 0.0f && u < (float)iw && u >= 0.0f)) continue;                       /* :311-316 (row = v, col = u) */
      const int64_t row = (int64_t)v, col = (int64_t)u;                                                /* .long(): truncation */
      depth[(c * ih + row) * iw + col] = dist;
      winner[(c * ih + row) * iw + col] = (int32_t)i;
    }
  }
}
