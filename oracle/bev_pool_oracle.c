/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, gcc) of the reference's bev_pool algorithm.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 *
 * Parity pin: the reference has NO CPU implementation and no tests/golden vectors for
 * bev_pool (bev_pool_cpu.cpp is the CUDA host wrapper).  The pin is
 * tests/golden/bev_pool_ref_*.npz, produced on an MI355X by the reference's own kernel
 * (oracle/_ref, hipified from /root/reference at build time) — see oracle/README.md.
 * Until those fixtures exist this oracle is "parity unpinned".
 *
 * Each function cites the reference lines it restates (paths under /root/reference).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* mmdet3d/ops/bev_pool/bev_pool.py:86-91
 *   ranks = coords[:,0]*(W*D*B) + coords[:,1]*(D*B) + coords[:,2]*B + coords[:,3]
 * coords: [n,4] int64 (x, y, z, b). */
void oracle_bev_pool_ranks(const int64_t* coords, int64_t n, int64_t B, int64_t D, int64_t H, int64_t W,
                           int64_t* ranks) {
  (void)H;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t* c = coords + 4 * i;
    ranks[i] = c[0] * (W * D * B) + c[1] * (D * B) + c[2] * B + c[3];
  }
}

/* mmdet3d/models/vtransforms/base.py:149-169
 *   geom = ((geom - (bx - dx/2)) / dx).long()      fp32 subtract, fp32 divide, truncate toward zero
 *   append batch index; kept = 0 <= idx < nx on all three axes
 * geom: [n,3] fp32 (batch-major, n = B * points_per_batch); origin = bx - dx/2 (fp32, computed by
 * the caller with fp32 tensor ops as the reference does).  coords out: [n,4] int64 (x,y,z,b);
 * kept out: [n] uint8.  */
void oracle_bev_cell_index(const float* geom, int64_t n, int64_t points_per_batch, const float* origin,
                           const float* dx, const int64_t* nx, int64_t* coords, uint8_t* kept) {
  for (int64_t i = 0; i < n; ++i) {
    int ok = 1;
    for (int a = 0; a < 3; ++a) {
      volatile float diff = geom[3 * i + a] - origin[a]; /* volatile: forbid fusing sub+div */
      volatile float q = diff / dx[a];
      int64_t idx = (int64_t)q; /* C cast == torch .long(): truncation toward zero */
      coords[4 * i + a] = idx;
      if (idx < 0 || idx >= nx[a]) ok = 0;
    }
    coords[4 * i + 3] = i / points_per_batch;
    kept[i] = (uint8_t)ok;
  }
}

/* mmdet3d/ops/bev_pool/bev_pool.py:39-46 (QuickCumsumCuda.forward)
 *   kept[0] = 1; kept[i] = ranks[i] != ranks[i-1]; starts = where(kept);
 *   lengths[k] = starts[k+1]-starts[k]; lengths[last] = n - starts[last]
 * ranks must be sorted.  Returns n_intervals. */
int64_t oracle_bev_pool_intervals(const int64_t* ranks_sorted, int64_t n, int32_t* starts, int32_t* lengths) {
  int64_t k = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (i == 0 || ranks_sorted[i] != ranks_sorted[i - 1]) starts[k++] = (int32_t)i;
  }
  for (int64_t j = 0; j < k; ++j) lengths[j] = (int32_t)((j + 1 < k ? starts[j + 1] : n) - starts[j]);
  return k;
}

/* mmdet3d/ops/bev_pool/src/bev_pool_cuda.cu:20-42 (bev_pool_kernel) with the zero-filled output of
 * bev_pool_cpu.cpp:40.  out[b,z,x,y,c] = sum_{i in interval} x[i,c]; geom rows are (x,y,z,b).
 * float64 accumulation: the accuracy oracle ("within 1e-4 for BEV feature sums"). */
void oracle_bev_pool_forward_f64(const float* x, const int32_t* geom, const int32_t* starts,
                                 const int32_t* lengths, int64_t n_intervals, int64_t c, int64_t b, int64_t d,
                                 int64_t h, int64_t w, double* out) {
  memset(out, 0, (size_t)(b * d * h * w * c) * sizeof(double));
  for (int64_t k = 0; k < n_intervals; ++k) {
    const int32_t* g = geom + 4 * (int64_t)starts[k];
    double* o = out + (int64_t)g[3] * d * h * w * c + (int64_t)g[2] * h * w * c + (int64_t)g[0] * w * c +
                (int64_t)g[1] * c;
    for (int64_t cc = 0; cc < c; ++cc) {
      double psum = 0.0;
      for (int64_t i = 0; i < lengths[k]; ++i) psum += (double)x[((int64_t)starts[k] + i) * c + cc];
      o[cc] = psum;
    }
  }
}

/* Same, fp32 sequential adds in row order — bit-for-bit what bev_pool_cuda.cu:37-40 computes for a
 * given row order (the reference's own order is not reproducible: argsort ties, bev_pool.py:92). */
void oracle_bev_pool_forward_f32(const float* x, const int32_t* geom, const int32_t* starts,
                                 const int32_t* lengths, int64_t n_intervals, int64_t c, int64_t b, int64_t d,
                                 int64_t h, int64_t w, float* out) {
  memset(out, 0, (size_t)(b * d * h * w * c) * sizeof(float));
  for (int64_t k = 0; k < n_intervals; ++k) {
    const int32_t* g = geom + 4 * (int64_t)starts[k];
    float* o = out + (int64_t)g[3] * d * h * w * c + (int64_t)g[2] * h * w * c + (int64_t)g[0] * w * c +
               (int64_t)g[1] * c;
    for (int64_t cc = 0; cc < c; ++cc) {
      volatile float psum = 0.0f;
      for (int64_t i = 0; i < lengths[k]; ++i) psum = psum + x[((int64_t)starts[k] + i) * c + cc];
      o[cc] = psum;
    }
  }
}

/* mmdet3d/ops/bev_pool/src/bev_pool_cuda.cu:61-84 (bev_pool_grad_kernel) with the zero-filled
 * x_grad of bev_pool_cpu.cpp:78: x_grad[i,c] = out_grad[cell(interval of i), c]. */
void oracle_bev_pool_backward(const float* out_grad, const int32_t* geom, const int32_t* starts,
                              const int32_t* lengths, int64_t n_intervals, int64_t n, int64_t c, int64_t b,
                              int64_t d, int64_t h, int64_t w, float* x_grad) {
  (void)b;
  memset(x_grad, 0, (size_t)(n * c) * sizeof(float));
  for (int64_t k = 0; k < n_intervals; ++k) {
    const int32_t* g = geom + 4 * (int64_t)starts[k];
    const float* go = out_grad + (int64_t)g[3] * d * h * w * c + (int64_t)g[2] * h * w * c +
                      (int64_t)g[0] * w * c + (int64_t)g[1] * c;
    for (int64_t i = 0; i < lengths[k]; ++i)
      memcpy(x_grad + ((int64_t)starts[k] + i) * c, go, (size_t)c * sizeof(float));
  }
}
