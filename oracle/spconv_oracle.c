/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, gcc) of the reference's spconv v1.x rulebook and sparse convolution.
 * Follows /root/reference/mmdet3d/ops/spconv:
 *   include/spconv/geometry.h:24-85    getValidOutPos       (offset numbering)
 *   include/spconv/geometry.h:144-194  getIndicePairsConv   (strided conv rulebook, CPU order)
 *   include/spconv/geometry.h:86-141,196-245 getValidOutPosTranspose / getIndicePairsDeConv (transposed conv)
 *   src/maxpool_cpu.cc:22-66           SparseMaxPool{Forward,Backward}Functor
 *   include/spconv/geometry.h:247-297  getIndicePairsSubM   (submanifold rulebook)
 *   include/spconv/spconv_ops.h:130 + indice.cu.h:112-145   (CUDA output order: ascending linear index)
 *   include/spconv/spconv_ops.h:260-361 indiceConv          (accumulation: centre GEMM, then offsets)
 *   include/spconv/spconv_ops.h:363-456 indiceConvBackward
 *
 * Parity pin: tests/golden/spconv_ref_*.npz hold outputs of the reference's own CPU functors
 * (oracle/_ref/sparse_conv_ext compiled from /root/reference; tests/golden/make_spconv_golden.py).
 * The per-offset GEMM of the reference is torch::mm_out (cuBLAS/hipBLASLt, un-vendored): no bit-exact
 * pin exists for the floating-point part; tolerance is the bar there.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ND 3

/* geometry.h:24-85.  out: rows of (o0,o1,o2,offset); returns number of VALID rows. */
static int valid_out_pos(const int32_t* in_pos, const int32_t* ksize, const int32_t* stride, const int32_t* pad,
                         const int32_t* dil, const int32_t* out_shape, int32_t* out) {
  int32_t lowers[ND], uppers[ND], counter[ND], counter_size[ND];
  int32_t num_points = 1, point_counter = 0;
  for (int i = 0; i < ND; ++i) {
    lowers[i] = (in_pos[i] - (ksize[i] - 1) * dil[i] - 1 + stride[i] + pad[i]) / stride[i];
    uppers[i] = (in_pos[i] + pad[i]) / stride[i];
  }
  for (int i = 0; i < ND; ++i) {
    counter_size[i] = (uppers[i] - lowers[i]) / dil[i] + 1;
    num_points *= counter_size[i];
    counter[i] = 0;
  }
  for (int i = 0; i < num_points; ++i) {
    int valid = 1;
    int32_t m = 1, offset = 0;
    for (int j = ND - 1; j >= 0; --j) {
      int32_t val = uppers[j] - counter[j] * dil[j];
      out[point_counter * (ND + 1) + j] = val;
      if (val < 0 || val > out_shape[j] - 1) valid = 0;
      offset += m * (in_pos[j] - val * stride[j] + pad[j]) / dil[j];
      m *= ksize[j];
    }
    out[point_counter * (ND + 1) + ND] = offset;
    if (valid) ++point_counter;
    counter[ND - 1] += 1;
    for (int c = ND - 1; c >= 0; --c) {
      if (counter[c] == counter_size[c] && c > 0) {
        counter[c - 1] += 1;
        counter[c] = 0;
      }
    }
  }
  return point_counter;
}

static int64_t row_idx(const int32_t* p, const int32_t* shape) {
  int64_t r = 0;
  for (int i = 0; i < ND; ++i) r = r * shape[i] + p[i];
  return r;
}

/* open-addressing map int64 -> int32 (stands in for the dense gridsOut array of the reference) */
typedef struct { int64_t* keys; int32_t* vals; uint64_t cap; } gmap;
static void gmap_init(gmap* m, uint64_t n) {
  uint64_t cap = 16; while (cap < 2 * n + 16) cap <<= 1;
  m->cap = cap; m->keys = (int64_t*)malloc(cap * 8); m->vals = (int32_t*)malloc(cap * 4);
  for (uint64_t i = 0; i < cap; ++i) m->keys[i] = -1;
}
static void gmap_free(gmap* m) { free(m->keys); free(m->vals); }
static int32_t* gmap_find(gmap* m, int64_t key, int create) {
  uint64_t h = ((uint64_t)key * 0x9E3779B97F4A7C15ull) & (m->cap - 1);
  while (m->keys[h] != -1 && m->keys[h] != key) h = (h + 1) & (m->cap - 1);
  if (m->keys[h] == -1) {
    if (!create) return NULL;
    m->keys[h] = key; m->vals[h] = -1;
  }
  return &m->vals[h];
}

/* geometry.h:247-297 (getIndicePairsSubM).  indices [n,4] (b,x,y,z); pairs [K,2,n] pre-filled with -1;
 * indice_num [K] zeroed.  subm uses stride 1 and padding ksize/2 (spconv_ops.h:78-81). */
void oracle_subm_indice_pairs(const int32_t* indices, int64_t n, const int32_t* shape, const int32_t* ksize,
                              const int32_t* dil, int32_t* pairs, int32_t* indice_num) {
  int32_t stride[ND] = {1, 1, 1}, pad[ND];
  int64_t vol = 1, K = 1;
  for (int i = 0; i < ND; ++i) { pad[i] = ksize[i] / 2; vol *= shape[i]; K *= ksize[i]; }
  gmap g; gmap_init(&g, (uint64_t)n);
  for (int64_t j = 0; j < n; ++j) *gmap_find(&g, row_idx(indices + 4 * j + 1, shape) + vol * indices[4 * j], 1) = (int32_t)j;
  int32_t* vp = (int32_t*)malloc((size_t)K * (ND + 1) * sizeof(int32_t));
  for (int64_t j = 0; j < n; ++j) {
    int nv = valid_out_pos(indices + 4 * j + 1, ksize, stride, pad, dil, shape, vp);
    for (int i = 0; i < nv; ++i) {
      const int32_t* p = vp + i * (ND + 1);
      int32_t off = p[ND];
      int32_t* v = gmap_find(&g, row_idx(p, shape) + vol * indices[4 * j], 0);
      if (v && *v > -1) {
        pairs[((int64_t)off * 2 + 0) * n + indice_num[off]] = (int32_t)j;
        pairs[((int64_t)off * 2 + 1) * n + indice_num[off]] = *v;
        indice_num[off]++;
      }
    }
  }
  free(vp); gmap_free(&g);
}

/* geometry.h:144-194 (getIndicePairsConv).  out_indices [n*K,4]; returns number of active outputs.
 * Output rows are numbered in order of first appearance — the CPU order. */
int64_t oracle_conv_indice_pairs(const int32_t* indices, int64_t n, const int32_t* ksize, const int32_t* stride,
                                 const int32_t* pad, const int32_t* dil, const int32_t* out_shape,
                                 int32_t* out_indices, int32_t* pairs, int32_t* indice_num) {
  int64_t vol = 1, K = 1;
  for (int i = 0; i < ND; ++i) { vol *= out_shape[i]; K *= ksize[i]; }
  gmap g; gmap_init(&g, (uint64_t)(n * K));
  int32_t* vp = (int32_t*)malloc((size_t)K * (ND + 1) * sizeof(int32_t));
  int64_t num_act = 0;
  for (int64_t j = 0; j < n; ++j) {
    int32_t b = indices[4 * j];
    int nv = valid_out_pos(indices + 4 * j + 1, ksize, stride, pad, dil, out_shape, vp);
    for (int i = 0; i < nv; ++i) {
      const int32_t* p = vp + i * (ND + 1);
      int32_t off = p[ND];
      int32_t* v = gmap_find(&g, row_idx(p, out_shape) + vol * b, 1);
      if (*v == -1) {
        for (int k = 0; k < ND; ++k) out_indices[4 * num_act + 1 + k] = p[k];
        out_indices[4 * num_act] = b;
        *v = (int32_t)num_act++;
      }
      pairs[((int64_t)off * 2 + 0) * n + indice_num[off]] = (int32_t)j;
      pairs[((int64_t)off * 2 + 1) * n + indice_num[off]] = *v;
      indice_num[off]++;
    }
  }
  free(vp); gmap_free(&g);
  return num_act;
}

/* geometry.h:86-141 (getValidOutPosTranspose): out = in*stride - pad + k*dil, offset = the tap index k. */
static int valid_out_pos_transpose(const int32_t* in_pos, const int32_t* ksize, const int32_t* stride, const int32_t* pad,
                                   const int32_t* dil, const int32_t* out_shape, int32_t* out) {
  int32_t lowers[ND], uppers[ND], counter[ND], counter_size[ND];
  int32_t num_points = 1, point_counter = 0;
  for (int i = 0; i < ND; ++i) {
    lowers[i] = in_pos[i] * stride[i] - pad[i];
    uppers[i] = lowers[i] + (ksize[i] - 1) * dil[i];
  }
  for (int i = 0; i < ND; ++i) {
    counter_size[i] = (uppers[i] - lowers[i]) / dil[i] + 1;
    num_points *= counter_size[i];
    counter[i] = 0;
  }
  for (int i = 0; i < num_points; ++i) {
    int valid = 1;
    int32_t m = 1, offset = 0;
    for (int j = ND - 1; j >= 0; --j) {
      int32_t val = uppers[j] - counter[j] * dil[j];
      out[point_counter * (ND + 1) + j] = val;
      if (val < 0 || val > out_shape[j] - 1) valid = 0;
      offset += m * (val - lowers[j]) / dil[j];
      m *= ksize[j];
    }
    out[point_counter * (ND + 1) + ND] = offset;
    if (valid) ++point_counter;
    counter[ND - 1] += 1;
    for (int c = ND - 1; c >= 0; --c) {
      if (counter[c] == counter_size[c] && c > 0) {
        counter[c - 1] += 1;
        counter[c] = 0;
      }
    }
  }
  return point_counter;
}

/* geometry.h:196-245 (getIndicePairsDeConv).  Same contract as oracle_conv_indice_pairs. */
int64_t oracle_deconv_indice_pairs(const int32_t* indices, int64_t n, const int32_t* ksize, const int32_t* stride,
                                   const int32_t* pad, const int32_t* dil, const int32_t* out_shape,
                                   int32_t* out_indices, int32_t* pairs, int32_t* indice_num) {
  int64_t vol = 1, K = 1;
  for (int i = 0; i < ND; ++i) { vol *= out_shape[i]; K *= ksize[i]; }
  gmap g; gmap_init(&g, (uint64_t)(n * K));
  int32_t* vp = (int32_t*)malloc((size_t)K * (ND + 1) * sizeof(int32_t));
  int64_t num_act = 0;
  for (int64_t j = 0; j < n; ++j) {
    int32_t b = indices[4 * j];
    int nv = valid_out_pos_transpose(indices + 4 * j + 1, ksize, stride, pad, dil, out_shape, vp);
    for (int i = 0; i < nv; ++i) {
      const int32_t* p = vp + i * (ND + 1);
      int32_t off = p[ND];
      int32_t* v = gmap_find(&g, row_idx(p, out_shape) + vol * b, 1);
      if (*v == -1) {
        for (int k = 0; k < ND; ++k) out_indices[4 * num_act + 1 + k] = p[k];
        out_indices[4 * num_act] = b;
        *v = (int32_t)num_act++;
      }
      pairs[((int64_t)off * 2 + 0) * n + indice_num[off]] = (int32_t)j;
      pairs[((int64_t)off * 2 + 1) * n + indice_num[off]] = *v;
      indice_num[off]++;
    }
  }
  free(vp); gmap_free(&g);
  return num_act;
}

/* maxpool_cpu.cc:22-40 + pool_ops.h:33: out starts at 0; offsets in ascending order; out takes strictly larger inputs. */
void oracle_indice_maxpool_f32(const float* features, const int32_t* pairs, const int32_t* indice_num, int64_t K,
                               int64_t L, int64_t n_out, int64_t C, float* out) {
  memset(out, 0, (size_t)(n_out * C) * sizeof(float));
  for (int64_t k = 0; k < K; ++k) {
    const int32_t* pin = pairs + (k * 2 + 0) * L;
    const int32_t* pout = pairs + (k * 2 + 1) * L;
    for (int64_t t = 0; t < indice_num[k]; ++t) {
      const float* f = features + (int64_t)pin[t] * C;
      float* o = out + (int64_t)pout[t] * C;
      for (int64_t c = 0; c < C; ++c)
        if (o[c] < f[c]) o[c] = f[c];
    }
  }
}

/* maxpool_cpu.cc:43-66 + pool_ops.h:69: din starts at 0; din[i] += dout[o] where out[o] == in[i]. */
void oracle_indice_maxpool_backward_f32(const float* features, const float* out_features, const float* out_grad,
                                        const int32_t* pairs, const int32_t* indice_num, int64_t K, int64_t L,
                                        int64_t n_in, int64_t C, float* in_grad) {
  memset(in_grad, 0, (size_t)(n_in * C) * sizeof(float));
  for (int64_t k = 0; k < K; ++k) {
    const int32_t* pin = pairs + (k * 2 + 0) * L;
    const int32_t* pout = pairs + (k * 2 + 1) * L;
    for (int64_t t = 0; t < indice_num[k]; ++t) {
      const int64_t i = (int64_t)pin[t] * C, o = (int64_t)pout[t] * C;
      for (int64_t c = 0; c < C; ++c)
        if (out_features[o + c] == features[i + c]) in_grad[i + c] += out_grad[o + c];
    }
  }
}

/* spconv_ops.h:260-361 (indiceConv), double accumulation (accuracy oracle).
 * features [n_in, cin]; filters [K, cin, cout] (= [kx,ky,kz,cin,cout] flattened, conv.py:100);
 * pairs [K,2,L]; out [n_out, cout].  inverse swaps the roles of the two pair columns.
 * The SubM centre shortcut (out += features @ W[argmax]) is the same sum as running the centre offset
 * through its identity pairs, so all offsets are treated uniformly here. */
void oracle_indice_conv_f64(const float* features, const float* filters, const int32_t* pairs,
                            const int32_t* indice_num, int64_t K, int64_t L, int64_t n_out, int64_t cin,
                            int64_t cout, int32_t inverse, double* out) {
  memset(out, 0, (size_t)(n_out * cout) * sizeof(double));
  for (int64_t k = 0; k < K; ++k) {
    const int32_t* pin = pairs + (k * 2 + (inverse ? 1 : 0)) * L;
    const int32_t* pout = pairs + (k * 2 + (inverse ? 0 : 1)) * L;
    const float* w = filters + k * cin * cout;
    for (int64_t t = 0; t < indice_num[k]; ++t) {
      const float* f = features + (int64_t)pin[t] * cin;
      double* o = out + (int64_t)pout[t] * cout;
      for (int64_t ci = 0; ci < cin; ++ci) {
        double a = f[ci];
        const float* wr = w + ci * cout;
        for (int64_t co = 0; co < cout; ++co) o[co] += a * (double)wr[co];
      }
    }
  }
}

/* spconv_ops.h:363-456 (indiceConvBackward), double accumulation.
 *   in_grad[i]  += out_grad[o] @ W[k]^T          for every pair (i,o) of offset k
 *   w_grad[k]   += features[i]^T @ out_grad[o]                                    */
void oracle_indice_conv_backward_f64(const float* features, const float* filters, const float* out_grad,
                                     const int32_t* pairs, const int32_t* indice_num, int64_t K, int64_t L,
                                     int64_t n_in, int64_t cin, int64_t cout, int32_t inverse, double* in_grad,
                                     double* w_grad) {
  memset(in_grad, 0, (size_t)(n_in * cin) * sizeof(double));
  memset(w_grad, 0, (size_t)(K * cin * cout) * sizeof(double));
  for (int64_t k = 0; k < K; ++k) {
    const int32_t* pin = pairs + (k * 2 + (inverse ? 1 : 0)) * L;
    const int32_t* pout = pairs + (k * 2 + (inverse ? 0 : 1)) * L;
    const float* w = filters + k * cin * cout;
    double* gw = w_grad + k * cin * cout;
    for (int64_t t = 0; t < indice_num[k]; ++t) {
      const float* f = features + (int64_t)pin[t] * cin;
      const float* go = out_grad + (int64_t)pout[t] * cout;
      double* gi = in_grad + (int64_t)pin[t] * cin;
      for (int64_t ci = 0; ci < cin; ++ci) {
        double acc = 0.0;
        for (int64_t co = 0; co < cout; ++co) {
          acc += (double)go[co] * (double)w[ci * cout + co];
          gw[ci * cout + co] += (double)f[ci] * (double)go[co];
        }
        gi[ci] += acc;
      }
    }
  }
}
