// Training-mode BatchNorm1d over sparse feature rows [N, C], with the ReLU and the residual add of the sparse blocks folded in
// (gfx950).  Reference behaviour kept: ops/sparse_block.py:88-107 (conv - bn - relu - conv - bn - (+identity) - relu),
// models/backbones/sparse_encoder.py:39 (BN1d, eps 1e-3, momentum 0.01) on torch.nn.BatchNorm1d semantics: batch statistics
// (biased variance) normalise, running_mean / running_var (unbiased) move by `momentum`.
//
// torch runs four kernels per BatchNorm (collect_statistics, transform_input, backward_reduce, backward_elemt: 99 + 35 + 115 +
// 39 us per layer at 4 frames, profiles/r03_train_step_amp_kernel_trace_stats.txt) plus separate ReLU / add / threshold_backward
// launches: ~6 of the 31.8 ms of the --amp training step.  Here: two launches forward, two backward, each a streaming pass:
//   bn_stats      per-channel sum / sum of squares of a row slab per workgroup (fp32), the LAST workgroup to finish adds the
//                 slab partials up in a fixed order (fp64) -> mean, 1/sqrt(var + eps), running statistics.  Deterministic.
//   bn_apply      y = ((x - mean) * invstd * w + b -> storage type) [+ residual -> storage type] [ReLU]
//   bn_bwd_reduce dz = ReLU'(y) * dy;  sum dz, sum dz * xhat per channel (same last-workgroup scheme) = d bias, d weight
//   bn_bwd_apply  dx = w * invstd * (dz - mean(dz) - xhat * mean(dz * xhat));  d residual = dz
// Rows are 16-bit (fp16 / bf16: 8 channels per lane) or fp32 (4 per lane); statistics and parameters are fp32.
#include "common.h"

namespace bevamd {
namespace bn {

constexpr int THREADS = 256;
constexpr int MAX_C = 256;

template <int DT> struct Row;          // DT 0: fp32 (4 channels per 16-byte vector), 1: fp16, 2: bf16 (8 per vector)
template <> struct Row<0> {
  static constexpr int VEC = 4;
  typedef float4 V;
  __device__ static void unpack(const V& v, float (&f)[4]) { f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w; }
  __device__ static V pack(const float (&f)[4]) { return make_float4(f[0], f[1], f[2], f[3]); }
  __device__ static float round(float x) { return x; }
};
struct alignas(16) H8 { uint32_t w[4]; };
template <> struct Row<1> {
  static constexpr int VEC = 8;
  typedef H8 V;
  __device__ static void unpack(const V& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = (float)__builtin_bit_cast(_Float16, (unsigned short)(v.w[i] & 0xFFFFu));
      f[2 * i + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(v.w[i] >> 16));
    }
  }
  __device__ static V pack(const float (&f)[8]) {
    V v;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      v.w[i] = (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f[2 * i]) |
               ((uint32_t)__builtin_bit_cast(unsigned short, (_Float16)f[2 * i + 1]) << 16);
    return v;
  }
  __device__ static float round(float x) { return (float)(_Float16)x; }
};
template <> struct Row<2> {
  static constexpr int VEC = 8;
  typedef H8 V;
  __device__ static void unpack(const V& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(v.w[i] << 16);
      f[2 * i + 1] = __uint_as_float(v.w[i] & 0xFFFF0000u);
    }
  }
  __device__ static unsigned short bf(float x) {   // round to nearest even, NaN kept
    const uint32_t u = __float_as_uint(x);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
  }
  __device__ static V pack(const float (&f)[8]) {
    V v;
#pragma unroll
    for (int i = 0; i < 4; ++i) v.w[i] = (uint32_t)bf(f[2 * i]) | ((uint32_t)bf(f[2 * i + 1]) << 16);
    return v;
  }
  __device__ static float round(float x) { return __uint_as_float((uint32_t)bf(x) << 16); }
};

// rows [lo, hi) of workgroup b when n rows are dealt out as gridDim.x contiguous slabs
__device__ __forceinline__ void slab_of(long long n, long long& lo, long long& hi) {
  const long long per = (n + gridDim.x - 1) / gridDim.x;
  lo = (long long)blockIdx.x * per;
  hi = lo + per < n ? lo + per : n;
}

// Per-channel totals of two quantities over all workgroups, in two levels so that no workgroup walks more than GROUP partials
// alone: every workgroup leaves its slab's sums in part[g][2][C] (fp32); the LAST of each group of GROUP consecutive workgroups to
// arrive (a ticket per group) adds the group's partials up in slab order in fp64 and leaves them in gpart[group][2][C]; the last
// GROUP leader to finish (one more ticket) adds the group sums up in group order and hands (A, B) per channel to `fin`.  Fixed
// orders throughout: bit-reproducible.  Tickets are zero before the launch and zero again after it.
constexpr int GROUP = 32;
constexpr int MAX_SLABS = 1024;
constexpr int MAX_GROUPS = MAX_SLABS / GROUP;
constexpr int BN_DEFAULT_SLABS = 256, BN_DEFAULT_UNROLL = 4;   // tools/time_bn.py, 20 layers at 4 frames: 192 slabs 0.42 / 1.21 ms (stats / backward), 256: 0.42 / 1.17, 512: 0.47 / 1.32
// Round 6: one ticket per 128-byte line.  The tickets of all groups used to sit in consecutive words of ONE line, so the 512 arrivals
// of a launch were 512 atomics on one L2 line, served one after another (~50 ns each): 25 us of a 55-us kernel, and the reason why
// MORE slabs made the reductions slower (1024 slabs: 81 us, 256: 36 us for the same 54 MB; tools/time_bn.py).
constexpr int TICKET_STRIDE = 32;   // words

// sum over `n` partial rows (stride floats/doubles apart) per channel, by one workgroup: thread -> (channel, share), four rows in flight
template <typename T, typename Out>
__device__ __forceinline__ void block_column_sums(const T* __restrict__ src, int n, size_t stride, int C, double (*red)[THREADS], Out out) {
  const int Cp = C <= THREADS ? C : THREADS;
  const int nshare = THREADS / Cp;
  for (int c0 = 0; c0 < C; c0 += Cp) {
    const int c = c0 + (int)threadIdx.x % Cp, s = (int)threadIdx.x / Cp;
    double a = 0.0, b = 0.0;
    if (c < C && s < nshare)
      for (int g = s; g < n; g += 4 * nshare) {   // four partials requested together, added in order
        T pa[4], pb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int gu = g + u * nshare < n ? g + u * nshare : g;
          pa[u] = __hip_atomic_load(src + (size_t)gu * stride + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          pb[u] = __hip_atomic_load(src + (size_t)gu * stride + C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (g + u * nshare < n) { a += (double)pa[u]; b += (double)pb[u]; }
      }
    red[0][threadIdx.x] = a;
    red[1][threadIdx.x] = b;
    __syncthreads();
    if ((int)threadIdx.x < Cp && c < C) {
      double ta = 0.0, tb = 0.0;
      for (int s2 = 0; s2 < nshare; ++s2) { ta += red[0][s2 * Cp + threadIdx.x]; tb += red[1][s2 * Cp + threadIdx.x]; }
      out(c, ta, tb);
    }
    __syncthreads();
  }
}

template <typename Fin>
__device__ __forceinline__ void finish_totals(const float* sa, const float* sb, int C, float* __restrict__ part,
                                              double* __restrict__ gpart, unsigned* __restrict__ tickets, Fin fin) {
  // sa / sb: this workgroup's sums in LDS, [C] each;  tickets[0 .. ngroups) per group, tickets[MAX_GROUPS] for the leaders
  __shared__ bool last;
  __shared__ double red[2][THREADS];
  const int G = (int)gridDim.x, grp = (int)blockIdx.x / GROUP, ngroups = (G + GROUP - 1) / GROUP;
  const int gsize = (grp + 1) * GROUP <= G ? GROUP : G - grp * GROUP;
  float* mine = part + (size_t)blockIdx.x * 2 * C;
  for (int c = threadIdx.x; c < C; c += THREADS) {
    __hip_atomic_store(mine + c, sa[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(mine + C + c, sb[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // no __threadfence() here (round 6): the barrier orders the partial stores of all threads before thread 0's ticket, and the ticket
  // is an acq_rel read-modify-write at agent scope — its release half publishes everything that happens-before it.  The fence was
  // executed by every wave of every workgroup (an L2 write-back each on a multi-XCD part) and made the kernels slower the more
  // slabs they were given: 81 us at 1024 slabs, 55 at 512, 36 at 256 for the same 54 MB (tools/time_bn.py).
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(tickets + grp * TICKET_STRIDE, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = t == (unsigned)gsize - 1u;
    if (last) __hip_atomic_store(tickets + grp * TICKET_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  double* gm = gpart + (size_t)grp * 2 * C;
  block_column_sums<float>(part + (size_t)grp * GROUP * 2 * C, gsize, (size_t)2 * C, C, red, [&](int c, double a, double b) {
    __hip_atomic_store(gm + c, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(gm + C + c, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  });
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(tickets + MAX_GROUPS * TICKET_STRIDE, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    last = t == (unsigned)ngroups - 1u;
    if (last) __hip_atomic_store(tickets + MAX_GROUPS * TICKET_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  block_column_sums<double>(gpart, ngroups, (size_t)2 * C, C, red, fin);
}

// this workgroup's per-channel sums of (p, q) over its rows -> LDS sa / sb [C]; `rowfn(row, cv, p[VEC], q[VEC])` supplies the terms
template <int VEC, int U, typename RowFn>
__device__ __forceinline__ void slab_sums(long long n, int C, float* sa, float* sb, RowFn rowfn) {
  extern __shared__ float dyn[];   // [rows in flight][2][C]
  const int lpr = C / VEC, rif = THREADS / lpr;
  const int r_in = (int)threadIdx.x / lpr, cv = (int)threadIdx.x - r_in * lpr;
  long long lo, hi;
  slab_of(n, lo, hi);
  float p[VEC], q[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) p[j] = q[j] = 0.f;
  // four rows per trip, their loads issued together (the callback loads all four before it touches the sums): with one load in
  // flight per lane the reductions ran at 0.8 TB/s (first cut, profiles/r04_train_step_amp_kernel_trace_stats_a.txt)
  if (r_in < rif)
    for (long long r = lo + r_in; r < hi; r += (long long)U * rif) rowfn(r, (long long)rif, hi, cv, p, q);
  if (r_in < rif) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      dyn[((size_t)r_in * 2 + 0) * C + cv * VEC + j] = p[j];
      dyn[((size_t)r_in * 2 + 1) * C + cv * VEC + j] = q[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += THREADS) {   // column sums in row order: deterministic (rif is a power of two >= 4)
    float t = 0.f;
    for (int r = 0; r < rif; r += 4) {
      const float v0 = dyn[(size_t)r * 2 * C + c], v1 = dyn[(size_t)(r + 1) * 2 * C + c], v2 = dyn[(size_t)(r + 2) * 2 * C + c],
                  v3 = dyn[(size_t)(r + 3) * 2 * C + c];
      t += v0; t += v1; t += v2; t += v3;
    }
    (c < C ? sa : sb)[c < C ? c : c - C] = t;
  }
  __syncthreads();
}

template <int DT, int U>
__global__ __launch_bounds__(THREADS) void bn_stats_kernel(const void* __restrict__ x_, long long n, int C, long long stride,
                                                           float eps, float momentum, float* __restrict__ mean,
                                                           float* __restrict__ invstd, float* __restrict__ running_mean,
                                                           float* __restrict__ running_var, float* __restrict__ part,
                                                           double* __restrict__ gpart, unsigned* __restrict__ ticket) {
  typedef Row<DT> R;
  typedef typename R::V V;
  __shared__ float sa[MAX_C], sb[MAX_C];
  const char* x = (const char*)x_;
  const size_t esz = DT == 0 ? 4 : 2;
  // SHIFTED sums: p = sum (x - x0), q = sum (x - x0)^2 with x0 = row 0 of the tensor — the same pivot in every
  // workgroup, a value of the channel's own distribution — so that var = q/n - (p/n)^2 does not cancel when |mean| >> std
  // (E[x^2] - mean^2 on raw fp32 sums loses ~1e-7 * mean^2 / var of relative accuracy; torch's batch_norm runs Welford).
  slab_sums<R::VEC, U>(n, C, sa, sb, [&](long long r, long long step, long long hi, int cv, float (&p)[R::VEC], float (&q)[R::VEC]) {
    V v[U];
    float x0[R::VEC];
    R::unpack(*(const V*)(x + (size_t)cv * R::VEC * esz), x0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long ru = r + u * step < hi ? r + u * step : r;
      v[u] = *(const V*)(x + ((size_t)ru * stride + (size_t)cv * R::VEC) * esz);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r + u * step < hi) {
        float f[R::VEC];
        R::unpack(v[u], f);
#pragma unroll
        for (int j = 0; j < R::VEC; ++j) { const float d = f[j] - x0[j]; p[j] += d; q[j] = fmaf(d, d, q[j]); }
      }
    }
  });
  finish_totals(sa, sb, C, part, gpart, ticket, [&](int c, double s, double ss) {
    float x0v[R::VEC];
    R::unpack(*(const V*)(x + (size_t)(c / R::VEC) * R::VEC * esz), x0v);
    const double ms = s / (double)n;                         // mean of the shifted values
    const double m = (double)x0v[c % R::VEC] + ms;
    double var = ss / (double)n - ms * ms;
    var = var > 0.0 ? var : 0.0;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * m);
    if (running_var) {
      const double unbiased = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
      running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
    }
  });
}

// Elementwise kernels: lanes-per-row (C / VEC) divides THREADS, so a thread keeps ONE channel group for all its rows — its
// per-channel constants load once — and walks rows `rows_per_trip` apart, four trips' loads issued together.
template <int DT>
__global__ __launch_bounds__(THREADS) void bn_apply_kernel(const void* __restrict__ x_, long long n, int C, long long stride,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ weight, const float* __restrict__ bias,
                                                           const void* __restrict__ res_, long long res_stride, int relu,
                                                           void* __restrict__ y_, long long y_stride) {
  typedef Row<DT> R;
  typedef typename R::V V;
  const size_t esz = DT == 0 ? 4 : 2;
  const int lpr = C / R::VEC;
  const int c0 = ((int)threadIdx.x % lpr) * R::VEC;
  const long long step = (long long)gridDim.x * (THREADS / lpr);                      // rows per trip of the whole grid
  float sc[R::VEC], sh[R::VEC];
#pragma unroll
  for (int j = 0; j < R::VEC; ++j) {
    const float w = weight ? weight[c0 + j] : 1.f, b = bias ? bias[c0 + j] : 0.f;
    sc[j] = invstd[c0 + j] * w;
    sh[j] = b;
  }
  float mu[R::VEC];
#pragma unroll
  for (int j = 0; j < R::VEC; ++j) mu[j] = mean[c0 + j];
  for (long long r0 = (long long)blockIdx.x * (THREADS / lpr) + (int)threadIdx.x / lpr; r0 < n; r0 += 4 * step) {
    V vx[4], vr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long r = r0 + u * step < n ? r0 + u * step : r0;
      vx[u] = *(const V*)((const char*)x_ + ((size_t)r * stride + c0) * esz);
      if (res_) vr[u] = *(const V*)((const char*)res_ + ((size_t)r * res_stride + c0) * esz);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r0 + u * step >= n) break;
      float f[R::VEC], g[R::VEC];
      R::unpack(vx[u], f);
      if (res_) R::unpack(vr[u], g);
#pragma unroll
      for (int j = 0; j < R::VEC; ++j) {
        // torch's transform_input: (x - mean) * (invstd * w) + b in fp32, rounded to the storage type; the add and the ReLU of the
        // block are separate 16-bit tensors in the unfused pipeline, so each rounds once
        float v = R::round((f[j] - mu[j]) * sc[j] + sh[j]);
        if (res_) v = R::round(v + g[j]);
        if (relu) v = v > 0.f ? v : (v != v ? v : 0.f);
        f[j] = v;
      }
      *(V*)((char*)y_ + ((size_t)(r0 + u * step) * y_stride + c0) * esz) = R::pack(f);
    }
  }
}

template <int DT, int U>
__global__ __launch_bounds__(THREADS) void bn_bwd_reduce_kernel(const void* __restrict__ dy_, long long dy_stride,
                                                                const void* __restrict__ y_, long long y_stride,
                                                                const void* __restrict__ x_, long long stride, long long n, int C,
                                                                int relu, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd, float* __restrict__ sum_dz,
                                                                float* __restrict__ sum_dz_xhat, float* __restrict__ part,
                                                                double* __restrict__ gpart, unsigned* __restrict__ ticket) {
  typedef Row<DT> R;
  typedef typename R::V V;
  __shared__ float sa[MAX_C], sb[MAX_C];
  const size_t esz = DT == 0 ? 4 : 2;
  slab_sums<R::VEC, U>(n, C, sa, sb, [&](long long r, long long step, long long hi, int cv, float (&p)[R::VEC], float (&q)[R::VEC]) {
    const int c0 = cv * R::VEC;
    float mu[R::VEC], is[R::VEC];
#pragma unroll
    for (int j = 0; j < R::VEC; ++j) { mu[j] = mean[c0 + j]; is[j] = invstd[c0 + j]; }
    V vd[U], vx[U], vy[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long ru = r + u * step < hi ? r + u * step : r;
      vd[u] = *(const V*)((const char*)dy_ + ((size_t)ru * dy_stride + c0) * esz);
      vx[u] = *(const V*)((const char*)x_ + ((size_t)ru * stride + c0) * esz);
      if (relu) vy[u] = *(const V*)((const char*)y_ + ((size_t)ru * y_stride + c0) * esz);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (r + u * step < hi) {
        float d[R::VEC], f[R::VEC], o[R::VEC];
        R::unpack(vd[u], d);
        R::unpack(vx[u], f);
        if (relu) R::unpack(vy[u], o);
#pragma unroll
        for (int j = 0; j < R::VEC; ++j) {
          const float dz = relu && o[j] <= 0.f ? 0.f : d[j];   // torch's threshold_backward: zero where y <= 0
          p[j] += dz;
          q[j] = fmaf(dz, (f[j] - mu[j]) * is[j], q[j]);
        }
      }
    }
  });
  finish_totals(sa, sb, C, part, gpart, ticket, [&](int c, double a, double b) {
    sum_dz[c] = (float)a;
    sum_dz_xhat[c] = (float)b;
  });
}

template <int DT>
__global__ __launch_bounds__(THREADS) void bn_bwd_apply_kernel(const void* __restrict__ dy_, long long dy_stride,
                                                               const void* __restrict__ y_, long long y_stride,
                                                               const void* __restrict__ x_, long long stride, long long n, int C,
                                                               int relu, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const float* __restrict__ weight,
                                                               const float* __restrict__ sum_dz,
                                                               const float* __restrict__ sum_dz_xhat, void* __restrict__ dx_,
                                                               long long dx_stride, void* __restrict__ dres_,
                                                               long long dres_stride) {
  typedef Row<DT> R;
  typedef typename R::V V;
  const size_t esz = DT == 0 ? 4 : 2;
  const int lpr = C / R::VEC;
  const int c0 = ((int)threadIdx.x % lpr) * R::VEC;
  const long long step = (long long)gridDim.x * (THREADS / lpr);
  const float inv_n = 1.f / (float)n;
  float mu[R::VEC], is[R::VEC], k0[R::VEC], k1[R::VEC], k2[R::VEC];
#pragma unroll
  for (int j = 0; j < R::VEC; ++j) {
    mu[j] = mean[c0 + j];
    is[j] = invstd[c0 + j];
    k0[j] = (weight ? weight[c0 + j] : 1.f) * is[j];     // w * invstd
    k1[j] = sum_dz[c0 + j] * inv_n;                       // mean of dz
    k2[j] = sum_dz_xhat[c0 + j] * inv_n;                  // mean of dz * xhat
  }
  for (long long r0 = (long long)blockIdx.x * (THREADS / lpr) + (int)threadIdx.x / lpr; r0 < n; r0 += 4 * step) {
    V vd[4], vx[4], vy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long r = r0 + u * step < n ? r0 + u * step : r0;
      vd[u] = *(const V*)((const char*)dy_ + ((size_t)r * dy_stride + c0) * esz);
      vx[u] = *(const V*)((const char*)x_ + ((size_t)r * stride + c0) * esz);
      if (relu) vy[u] = *(const V*)((const char*)y_ + ((size_t)r * y_stride + c0) * esz);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r0 + u * step >= n) break;
      float d[R::VEC], f[R::VEC], o[R::VEC];
      R::unpack(vd[u], d);
      R::unpack(vx[u], f);
      if (relu) R::unpack(vy[u], o);
#pragma unroll
      for (int j = 0; j < R::VEC; ++j) {
        const float dz = relu && o[j] <= 0.f ? 0.f : d[j];   // torch's threshold_backward: zero where y <= 0
        d[j] = dz;
        const float xhat = (f[j] - mu[j]) * is[j];
        f[j] = k0[j] * (dz - k1[j] - xhat * k2[j]);
      }
      const size_t r = (size_t)(r0 + u * step);
      *(V*)((char*)dx_ + (r * dx_stride + c0) * esz) = R::pack(f);
      if (dres_) *(V*)((char*)dres_ + (r * dres_stride + c0) * esz) = R::pack(d);
    }
  }
}

static int check_shape(const char* who, long long n, int c, int dtype) {
  const int vec = dtype == 0 ? 4 : 8;
  if (n < 0 || c <= 0 || c > MAX_C || dtype < 0 || dtype > 2 || c % vec != 0 || THREADS % (c / vec) != 0) {
    set_error("%s: unsupported shape (n >= 0, 0 < c <= %d, c a multiple of %d with c / %d dividing %d): n=%lld c=%d dtype=%d", who,
              MAX_C, vec, vec, THREADS, n, c, dtype);
    return BEVAMD_ERR_INVALID_ARG;
  }
  return BEVAMD_OK;
}

// tuning knobs of the two reducing kernels (round 6): BEVAMD_BN_SLABS = most workgroups (default 256, at most MAX_SLABS),
// BEVAMD_BN_UNROLL = rows a lane keeps in flight per trip (4 | 8)
static int bn_max_slabs() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("BEVAMD_BN_SLABS"); v = e ? atoi(e) : BN_DEFAULT_SLABS; v = v < 1 ? 1 : v > MAX_SLABS ? MAX_SLABS : v; }
  return v;
}
static int bn_unroll() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("BEVAMD_BN_UNROLL"); v = e ? atoi(e) : BN_DEFAULT_UNROLL; v = v == 8 ? 8 : 4; }
  return v;
}
static unsigned reduce_grid(long long n, int c, int dtype) {
  const int vec = dtype == 0 ? 4 : 8, rif = THREADS / (c / vec);
  // >= 16 rows per lane (four trips of four), at most 512 slabs: a slab's fixed cost (LDS column sums, partial stores, fence,
  // ticket: ~5 us) is paid per workgroup — 1024 slabs measured 62 / 95 us per layer (stats / backward reduce) against 50 / 68 at 512
  long long g = n / ((long long)rif * 16);
  return (unsigned)(g < 1 ? 1 : g > bn_max_slabs() ? bn_max_slabs() : g);
}

static unsigned apply_grid(long long n, int c, int dtype) {
  const int vec = dtype == 0 ? 4 : 8;
  const int rpw = THREADS / (c / vec);                               // rows per workgroup and trip
  long long g = (n + (long long)rpw * 4 - 1) / ((long long)rpw * 4);   // one batch of four rows per lane up to 4096 workgroups
  return (unsigned)(g < 1 ? 1 : g > 4096 ? 4096 : g);
}

}  // namespace bn
}  // namespace bevamd

using namespace bevamd;
using namespace bevamd::bn;

extern "C" {

/* Scratch of the two reducing kernels: slab partials (fp32) + one ticket word (must be ZERO before the first launch that uses
 * it; every launch leaves it zero again).  bytes: bevamd_sparse_bn_workspace_bytes(c). */
static size_t ws_part_bytes(int c) { return align_up((size_t)MAX_SLABS * 2 * c * sizeof(float), 256); }
static size_t ws_gpart_bytes(int c) { return align_up((size_t)MAX_GROUPS * 2 * c * sizeof(double), 256); }
size_t bevamd_sparse_bn_workspace_bytes(int c) {
  return c > 0 ? ws_part_bytes(c) + ws_gpart_bytes(c) + align_up((size_t)(MAX_GROUPS + 1) * TICKET_STRIDE * sizeof(unsigned), 256) : 0;
}

/* Training-mode statistics of x [n, c] (row pitch `stride` elements; dtype 0 fp32 | 1 fp16 | 2 bf16): mean [c], invstd [c] =
 * 1 / sqrt(biased var + eps) (fp32), and — when given — running_mean / running_var updated in place with `momentum` (unbiased
 * variance), as torch.nn.BatchNorm1d does in train().  ws: see bevamd_sparse_bn_workspace_bytes. */
int bevamd_sparse_bn_stats(const void* x, int dtype, long long n, int c, long long stride, float eps, float momentum, float* mean,
                           float* invstd, float* running_mean, float* running_var, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_shape("sparse_bn_stats", n, c, dtype);
  if (rc) return rc;
  BEVAMD_REQUIRE(n > 0 && x && mean && invstd && stride >= c, "sparse_bn_stats: null buffer / no rows / bad pitch");
  if (!ws || ws_bytes < bevamd_sparse_bn_workspace_bytes(c)) {
    set_error("sparse_bn_stats: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  float* part = (float*)ws;
  double* gpart = (double*)((char*)ws + ws_part_bytes(c));
  unsigned* ticket = (unsigned*)((char*)ws + ws_part_bytes(c) + ws_gpart_bytes(c));   // MAX_GROUPS + 1 words, zero between launches
  const unsigned g = reduce_grid(n, c, dtype);
  const int vec = dtype == 0 ? 4 : 8;
  const size_t lds = (size_t)(THREADS / (c / vec)) * 2 * c * sizeof(float);
#define BEVAMD_GO(DT)                                                                                                                          \
  do {                                                                                                                                         \
    if (bn_unroll() == 8) bn_stats_kernel<DT, 8><<<dim3(g), dim3(THREADS), lds, stream>>>(x, n, c, stride, eps, momentum, mean, invstd, running_mean, running_var, part, gpart, ticket); \
    else bn_stats_kernel<DT, 4><<<dim3(g), dim3(THREADS), lds, stream>>>(x, n, c, stride, eps, momentum, mean, invstd, running_mean, running_var, part, gpart, ticket); \
  } while (0)
  if (dtype == 0) BEVAMD_GO(0); else if (dtype == 1) BEVAMD_GO(1); else BEVAMD_GO(2);
#undef BEVAMD_GO
  BEVAMD_LAUNCH_CHECK("bn_stats");
  return BEVAMD_OK;
}

/* y = BatchNorm(x) with the given statistics [+ residual] [ReLU], rounded to the storage type after each of the three steps (the
 * unfused pipeline's tensors).  weight / bias / residual optional (NULL). */
int bevamd_sparse_bn_apply(const void* x, int dtype, long long n, int c, long long stride, const float* mean, const float* invstd,
                           const float* weight, const float* bias, const void* residual, long long res_stride, int relu, void* y,
                           long long y_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_shape("sparse_bn_apply", n, c, dtype);
  if (rc) return rc;
  if (n == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(x && y && mean && invstd && stride >= c && y_stride >= c && (!residual || res_stride >= c),
                 "sparse_bn_apply: null buffer / bad pitch");
  const unsigned g = apply_grid(n, c, dtype);
#define BEVAMD_GO(DT) bn_apply_kernel<DT><<<dim3(g), dim3(THREADS), 0, stream>>>(x, n, c, stride, mean, invstd, weight, bias, residual, res_stride, relu, y, y_stride)
  if (dtype == 0) BEVAMD_GO(0); else if (dtype == 1) BEVAMD_GO(1); else BEVAMD_GO(2);
#undef BEVAMD_GO
  BEVAMD_LAUNCH_CHECK("bn_apply");
  return BEVAMD_OK;
}

/* Backward of bevamd_sparse_bn_apply: dz = dy where the forward output y was positive (relu != 0; y may be NULL otherwise);
 * sum_dz [c] = d bias, sum_dz_xhat [c] = d weight (fp32, deterministic); dx [n, c] in the storage type; d_residual (optional) = dz. */
int bevamd_sparse_bn_backward(const void* dy, long long dy_stride, const void* y, long long y_stride, const void* x, long long stride,
                              int dtype, long long n, int c, int relu, const float* mean, const float* invstd, const float* weight,
                              float* sum_dz, float* sum_dz_xhat, void* dx, long long dx_stride, void* d_residual,
                              long long dres_stride, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_shape("sparse_bn_backward", n, c, dtype);
  if (rc) return rc;
  BEVAMD_REQUIRE(n > 0 && dy && x && dx && mean && invstd && sum_dz && sum_dz_xhat && (!relu || y), "sparse_bn_backward: null buffer / no rows");
  BEVAMD_REQUIRE(stride >= c && dy_stride >= c && dx_stride >= c && (!relu || y_stride >= c) && (!d_residual || dres_stride >= c),
                 "sparse_bn_backward: bad pitch");
  if (!ws || ws_bytes < bevamd_sparse_bn_workspace_bytes(c)) {
    set_error("sparse_bn_backward: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  float* part = (float*)ws;
  double* gpart = (double*)((char*)ws + ws_part_bytes(c));
  unsigned* ticket = (unsigned*)((char*)ws + ws_part_bytes(c) + ws_gpart_bytes(c));
  const unsigned g = reduce_grid(n, c, dtype), ga = apply_grid(n, c, dtype);
  const int vec = dtype == 0 ? 4 : 8;
  const size_t lds = (size_t)(THREADS / (c / vec)) * 2 * c * sizeof(float);
#define BEVAMD_GO(DT)                                                                                                              \
  do {                                                                                                                             \
    if (bn_unroll() == 8)                                                                                                          \
      bn_bwd_reduce_kernel<DT, 8><<<dim3(g), dim3(THREADS), lds, stream>>>(dy, dy_stride, y, y_stride, x, stride, n, c, relu, mean,  \
                                                                         invstd, sum_dz, sum_dz_xhat, part, gpart, ticket);        \
    else                                                                                                                           \
      bn_bwd_reduce_kernel<DT, 4><<<dim3(g), dim3(THREADS), lds, stream>>>(dy, dy_stride, y, y_stride, x, stride, n, c, relu, mean,  \
                                                                         invstd, sum_dz, sum_dz_xhat, part, gpart, ticket);        \
    bn_bwd_apply_kernel<DT><<<dim3(ga), dim3(THREADS), 0, stream>>>(dy, dy_stride, y, y_stride, x, stride, n, c, relu, mean, invstd, \
                                                                    weight, sum_dz, sum_dz_xhat, dx, dx_stride, d_residual,        \
                                                                    dres_stride);                                                  \
  } while (0)
  if (dtype == 0) BEVAMD_GO(0); else if (dtype == 1) BEVAMD_GO(1); else BEVAMD_GO(2);
#undef BEVAMD_GO
  BEVAMD_LAUNCH_CHECK("bn_backward");
  return BEVAMD_OK;
}

}  // extern "C"
