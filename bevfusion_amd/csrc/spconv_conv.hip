// Sparse 3D convolution for gfx950: output-stationary implicit GEMM on MFMA.
//
// Replaces (reference, /root/reference/mmdet3d/ops/spconv):
//   include/spconv/spconv_ops.h:260-361   indiceConv<T>: per kernel offset gather -> torch::mm_out ->
//                                         scatter-add (up to 3 launches x 27 offsets, a D2H sync, and
//                                         output rows read-modify-written once per offset)
//   include/spconv/reordering.cu.h:21-157 gather / scatterAdd kernels
//   include/spconv/spconv_ops.h:363-456   indiceConvBackward<T>
//
// One launch per convolution.  A wavefront owns 16*MT output rows and all output channels; for each
// kernel offset k it gathers the rows nbr[k][o] straight into MFMA A-fragments (16-byte loads, no
// staging buffer), multiplies by W[k] from a pre-transposed filter image and accumulates in fp32
// registers across ALL offsets; the finished tile is written once, with bias / BatchNorm scale+shift /
// residual / ReLU applied in the epilogue.  Offsets with no neighbour in the whole tile are skipped
// with one ballot.  No atomics, no scatter, fixed summation order (offset-major, then channel) ->
// bit-reproducible.
//
//   16-bit (fp16 / bf16): v_mfma_f32_16x16x32  — lane l holds A[row l&15][8 ch of group l>>4]
//   fp32                : v_mfma_f32_16x16x4   — exact fp32 FMA chain (no TF32-like path on gfx950);
//                         one float4 load feeds 4 MFMAs (k-slot permutation shared by A and B)
//
// MFMA is used for nothing else on this path (the op is gather/L2-bound).
#include "common.h"

namespace bevamd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum DType { DT_F32 = 0, DT_F16 = 1, DT_BF16 = 2 };

struct alignas(16) Raw16 { uint32_t w[4]; };

template <int DT> struct Elem;
template <> struct Elem<DT_F32> {
  typedef float T;
  static constexpr int VEC = 4;   // elements per 16-byte fragment load
  static constexpr int CH = 16;   // input channels consumed per chunk (4 lane groups x VEC)
  __device__ static float to_f32(float v) { return v; }
  __device__ static float from_f32(float v) { return v; }
};
template <> struct Elem<DT_F16> {
  typedef _Float16 T;
  static constexpr int VEC = 8;
  static constexpr int CH = 32;
  __device__ static float to_f32(_Float16 v) { return (float)v; }
  __device__ static _Float16 from_f32(float v) { return (_Float16)v; }
};
template <> struct Elem<DT_BF16> {
  typedef uint16_t T;  // raw bf16 bits
  static constexpr int VEC = 8;
  static constexpr int CH = 32;
  __device__ static float to_f32(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
  __device__ static uint16_t from_f32(float v) {  // round to nearest even
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  }
};

template <int DT>
__device__ __forceinline__ void mfma_step(const Raw16& a, const Raw16& b, f32x4& acc) {
  if constexpr (DT == DT_F16) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*(const f16x8*)&a, *(const f16x8*)&b, acc, 0, 0, 0);
  } else if constexpr (DT == DT_BF16) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)&a, *(const bf16x8*)&b, acc, 0, 0, 0);
  } else {
    const float* af = (const float*)&a;
    const float* bf = (const float*)&b;
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t], bf[t], acc, 0, 0, 0);
  }
}

// 16 bytes (VEC elements) of a feature row starting at channel c; zero past `cin` / for row < 0.
template <int DT>
__device__ __forceinline__ Raw16 load_a(const typename Elem<DT>::T* __restrict__ feat, int row, int cin, int c,
                                        bool vec_ok) {
  typedef typename Elem<DT>::T T;
  constexpr int VEC = Elem<DT>::VEC;
  Raw16 r;
  r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0u;
  if (row < 0 || c >= cin) return r;
  const T* p = feat + (size_t)row * cin + c;
  if (vec_ok) {
    r = *(const Raw16*)p;
  } else {
    T* e = (T*)&r;
#pragma unroll
    for (int j = 0; j < VEC; ++j)
      if (c + j < cin) e[j] = p[j];
  }
  return r;
}

struct Epilogue {
  const void* bias;       // [cout], same dtype as features (conv.py:215-216), or null
  const float* scale;     // [cout] fp32, or null   } folded BatchNorm: y = y * scale + shift
  const float* shift;     // [cout] fp32, or null   }
  const void* residual;   // [m, cout], same dtype, or null
  int relu;
};

// wt: prepared filters [cout_pad][K * cin_pad] (see prepare kernel), zero padded.
template <int DT, int NT, int MT>
__global__ __launch_bounds__(256) void spconv_fwd_kernel(
    const typename Elem<DT>::T* __restrict__ feat, const typename Elem<DT>::T* __restrict__ wt,
    const int* __restrict__ nbr, int nbr_stride, int m_cap, const int* __restrict__ m_dev, int K, int cin,
    int cin_pad, int cout, typename Elem<DT>::T* __restrict__ out, Epilogue ep) {
  typedef typename Elem<DT>::T T;
  constexpr int VEC = Elem<DT>::VEC;
  constexpr int CH = Elem<DT>::CH;
  const int m = m_dev ? *m_dev : m_cap;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int row0 = wave * 16 * MT;
  if (row0 >= m) return;
  const int lane = threadIdx.x & 63;
  const int r = lane & 15;  // A row / B column inside a 16x16 tile
  const int g = lane >> 4;  // k-group
  const bool vec_ok = (cin % VEC) == 0;
  const size_t wrow = (size_t)K * cin_pad;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k = 0; k < K; ++k) {
    int nb[MT];
    bool any = false;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int row = row0 + mt * 16 + r;
      nb[mt] = row < m ? nbr[(size_t)k * nbr_stride + row] : -1;
      any |= nb[mt] >= 0;
    }
    if (!__any(any)) continue;  // no output row of this tile has a neighbour at offset k
    const T* wk = wt + (size_t)k * cin_pad;
    for (int c0 = 0; c0 < cin_pad; c0 += CH) {
      const int c = c0 + g * VEC;
      Raw16 a[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) a[mt] = load_a<DT>(feat, nb[mt], cin, c, vec_ok);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const Raw16 b = *(const Raw16*)(wk + (size_t)(nt * 16 + r) * wrow + c);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) mfma_step<DT>(a[mt], b, acc[mt][nt]);
      }
    }
  }

  // C/D layout of the 16x16 MFMA: column = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col = nt * 16 + r;
      if (col >= cout) continue;
      float bias = ep.bias ? Elem<DT>::to_f32(((const T*)ep.bias)[col]) : 0.f;
      float sc = ep.scale ? ep.scale[col] : 1.f;
      float sh = ep.shift ? ep.shift[col] : 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = row0 + mt * 16 + g * 4 + j;
        if (row >= m) continue;
        float v = acc[mt][nt][j];
        if (ep.bias) v = Elem<DT>::to_f32(Elem<DT>::from_f32(v)) + bias;  // reference adds bias after the cast
        if (ep.scale || ep.shift) v = v * sc + sh;
        if (ep.residual) v += Elem<DT>::to_f32(((const T*)ep.residual)[(size_t)row * cout + col]);
        if (ep.relu) v = v > 0.f ? v : 0.f;
        out[(size_t)row * cout + col] = Elem<DT>::from_f32(v);
      }
    }
  }
}

// filters [K][cin][cout] (reference layout [kx,ky,kz,cin,cout], conv.py:100)
//   -> wt [cout_pad][K][cin_pad], zero padded; transpose_io swaps the roles of cin/cout
//      (wt'[ci][k][co] = W[k][ci][co]) for the input-gradient pass.
template <int DT>
__global__ __launch_bounds__(256) void spconv_prepare_filters_kernel(const typename Elem<DT>::T* __restrict__ w, int K,
                                                                     int cin, int cout, int rows_pad, int cols_pad,
                                                                     int transpose_io,
                                                                     typename Elem<DT>::T* __restrict__ wt) {
  // rows = output channels of this pass, cols = input channels of this pass
  const int rows = transpose_io ? cin : cout;
  const int cols = transpose_io ? cout : cin;
  size_t total = (size_t)rows_pad * K * cols_pad;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    int c = (int)(i % cols_pad);
    size_t t = i / cols_pad;
    int k = (int)(t % K);
    int rr = (int)(t / K);
    typename Elem<DT>::T v = Elem<DT>::from_f32(0.f);
    if (rr < rows && c < cols) {
      int ci = transpose_io ? rr : c, co = transpose_io ? c : rr;
      v = w[((size_t)k * cin + ci) * cout + co];
    }
    wt[i] = v;
  }
}

// ---------------------------------------------------------------------------
// filter gradient (spconv_ops.h:363-456, the `filtersGrad[i] = buf^T * gradbuf` GEMM of every offset):
//   gW[k][ci][co] = sum over output rows o with i = nbr[k][o] >= 0 of feat[i][ci] * gout[o][co]
// — the rulebook-gathered GEMM of the backward pass, on MFMA (v_mfma_f32_16x16x4_f32: exact fp32 products and sums for every
// feature dtype; 16-bit inputs are widened while they are staged), deterministic, without atomics:
//   pass 1  grid (row slab s, kernel offset k, 64-wide output-channel group z): a workgroup stages 32 rows at a time —
//           feat[nbr[k][o]] (zeros where the offset has no neighbour) and gout[o] — into LDS as fp32 and accumulates
//           X^T * gY for its slab: wave w owns output-channel tile w of the group and every input-channel tile
//           (A = X^T: lane (ci, r) <- fa[r][ci]; B = gY: lane (r, co) <- fb[r][co]; one ds_read_b32 each, row pitch padded by
//           16 floats so that the two 32-lane halves of a read hit disjoint banks); the slab's partial goes to
//           part[s][k][ci][co] (fp32) — every entry written exactly once, no memset;
//   pass 2  gW[k][ci][co] = sum_s part[s][k][ci][co], s ascending (fixed order -> bit-reproducible), cast to the filter dtype.
// Any cin, cout <= 128 (padded to multiples of 16 inside the kernel).
// ---------------------------------------------------------------------------
constexpr int WG_ROWS = 32;      // rows staged per iteration
constexpr int WG_COG = 64;       // output channels per workgroup (4 waves x one 16-wide tile)
constexpr int WG_MAX_SLABS = 32;

template <int DT, int CIT>   // CIT = input-channel tiles (cin_pad / 16): 1, 2, 4 or 8
__global__ __launch_bounds__(256) void spconv_wgrad_mfma_kernel(const typename Elem<DT>::T* __restrict__ feat,
                                                                const typename Elem<DT>::T* __restrict__ gout,
                                                                const int* __restrict__ nbr, int nbr_stride, int m, int K,
                                                                int cin, int cout, int cout_pad, int rows_per_slab,
                                                                float* __restrict__ part) {
  constexpr int CINP = CIT * 16;
  constexpr int LDA = CINP + 16, LDB = WG_COG + 16;
  __shared__ float fa[WG_ROWS * LDA];
  __shared__ float fb[WG_ROWS * LDB];
  __shared__ int rowmap[WG_ROWS];
  const int s = blockIdx.x, k = blockIdx.y, co0 = blockIdx.z * WG_COG;
  const int rbeg = s * rows_per_slab;
  const int rend = rbeg + rows_per_slab < m ? rbeg + rows_per_slab : m;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  f32x4 acc[CIT];
#pragma unroll
  for (int t = 0; t < CIT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool wave_live = co0 + w * 16 < cout_pad;   // a narrow layer leaves the upper waves of the group without a tile
  for (int r0 = rbeg; r0 < rend; r0 += WG_ROWS) {
    __syncthreads();   // previous chunk fully consumed
    if (threadIdx.x < WG_ROWS) {
      const int o = r0 + (int)threadIdx.x;
      rowmap[threadIdx.x] = o < rend ? nbr[(size_t)k * nbr_stride + o] : -1;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < WG_ROWS * CINP; t += 256) {
      const int rr = t / CINP, c = t - rr * CINP;
      const int i = rowmap[rr];
      fa[rr * LDA + c] = (i >= 0 && c < cin) ? Elem<DT>::to_f32(feat[(size_t)i * cin + c]) : 0.f;
    }
    for (int t = threadIdx.x; t < WG_ROWS * WG_COG; t += 256) {
      const int rr = t / WG_COG, c = t - rr * WG_COG;
      const int o = r0 + rr, co = co0 + c;
      fb[rr * LDB + c] = (rowmap[rr] >= 0 && co < cout) ? Elem<DT>::to_f32(gout[(size_t)o * cout + co]) : 0.f;
    }
    __syncthreads();
    if (wave_live) {
#pragma unroll
      for (int q = 0; q < WG_ROWS / 4; ++q) {
        const int r = q * 4 + l4;
        const float b = fb[r * LDB + w * 16 + l15];
#pragma unroll
        for (int t = 0; t < CIT; ++t) {
          const float a = fa[r * LDA + t * 16 + l15];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
      }
    }
  }
  if (!wave_live) return;
  // D[i = ci][j = co]: lane holds column co = l15, rows ci = l4 * 4 + e of tile t
  const int co = co0 + w * 16 + l15;
  float* dst = part + ((size_t)s * K + k) * CINP * cout_pad;
#pragma unroll
  for (int t = 0; t < CIT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[(size_t)(t * 16 + l4 * 4 + e) * cout_pad + co] = acc[t][e];
}

// ---------------------------------------------------------------------------
// filter gradient, second formulation (round 3): ROW-TILE stationary.  The kernel above puts the kernel offset k in the grid, so
// every offset re-reads the whole out_grad matrix and walks its own column of the neighbour table: 27 passes over M rows each,
// one dependent (table -> gather) round trip and three workgroup barriers per 32 rows — 1.4-3.1 ms per layer at 4 frames, 10 % of
// the fp32 MFMA rate.  Here a workgroup owns a slab of rows and a (32 input) x (32 output) channel block and walks its rows ONCE:
//   * the out_grad tile (32 rows x 32 channels) is staged in LDS once per tile (double buffered: ONE barrier per tile);
//   * wave w owns the kernel offsets k = w, w + 4, ...: it keeps their accumulators (7 x 4 MFMA tiles) in registers for the whole
//     slab, fetches nbr[k][tile rows] for all its offsets up front, gathers the 32 neighbour rows of an offset into its PRIVATE
//     LDS tile while the MFMAs of the previous offset run (no workgroup barrier), and skips an offset no row of the tile reaches
//     (level 1: 4.6 of 27 taps exist);
//   * v_mfma_f32_16x16x4_f32 as before (exact fp32 for every feature dtype), A = X^T and B = gY read with ds_read_b32 from rows
//     48 floats apart (consecutive rows 16 banks apart: the two 32-lane halves of a read hit disjoint banks).
// Slab partials part[s][k][ci][co] and the fixed-order reduce are unchanged: deterministic, no atomics.
// ---------------------------------------------------------------------------
// Round 5, X3 flavour (fp32 rows only): the same walk, the products on the bf16 matrix cores by three-way operand splitting
// (see spconv_tile_f32x3.hip: x = hi + mid + lo exactly, six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block instead of eight
// v_mfma_f32_16x16x4_f32 at the fp32 vector rate: 96 against 256 MFMA cycles).  One MFMA reduces all 32 rows of the tile: lane
// (i = l15, g = l4) supplies rows r = e * 4 + g (e = 0..7) of its channel — consecutive g are consecutive rows, 16 banks apart
// with the 48-float row pitch, so the eight column reads are conflict-free; out_grad's fragments are split ONCE per tile and reused
// by the wave's offsets.
typedef unsigned int wg_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wg_split3(const float (&v)[8], wg_u32x4& hi, wg_u32x4& mid, wg_u32x4& lo) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = __float_as_uint(v[e]) & 0xFFFF0000u;
    const float r1 = v[e] - __uint_as_float(h[e]);
    m[e] = __float_as_uint(r1) & 0xFFFF0000u;
    l[e] = __float_as_uint(r1 - __uint_as_float(m[e]));
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    hi[p] = __builtin_amdgcn_perm(h[2 * p + 1], h[2 * p], 0x07060302u);
    mid[p] = __builtin_amdgcn_perm(m[2 * p + 1], m[2 * p], 0x07060302u);
    lo[p] = __builtin_amdgcn_perm(l[2 * p + 1], l[2 * p], 0x07060302u);
  }
}
__device__ __forceinline__ f32x4 wg_mfma_bf16(const wg_u32x4& a, const wg_u32x4& b, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

constexpr int WG2_R = 32;          // rows per tile
constexpr int WG2_NW = 4;          // waves per workgroup; wave w owns kernel offsets w, w + 4, ...
constexpr int WG2_TARGET_WGS = 768;
constexpr int WG2_MAX_SLABS = 256;

template <int DT, int CIT, int COT, int KPW, bool X3 = false>
__global__ __launch_bounds__(256, 2) void spconv_wgrad2_kernel(const typename Elem<DT>::T* __restrict__ feat,
                                                            const typename Elem<DT>::T* __restrict__ gout,
                                                            const int* __restrict__ nbr, int nbr_stride, int m, int K, int cin,
                                                            int cout, int cinp_tot, int coutp_tot, int rows_per_slab,
                                                            float* __restrict__ part) {
  typedef typename Elem<DT>::T T;
  constexpr int R = WG2_R, CIB = CIT * 16, COB = COT * 16;
  constexpr int LDA = CIB + 16, LDB = COB + 16;
  constexpr int A4 = CIB / 4, B4 = COB / 4;          // 4-element pieces per row
  constexpr int NLD = R * A4 / 64;                   // gathered pieces per lane and offset
  __shared__ __attribute__((aligned(16))) float fb[2][R * LDB];
  __shared__ __attribute__((aligned(16))) float fa[WG2_NW][R * LDA];
  const int s = blockIdx.x, ci0 = blockIdx.y * CIB, co0 = blockIdx.z * COB;
  const int rbeg = s * rows_per_slab;
  const int rend = rbeg + rows_per_slab < m ? rbeg + rows_per_slab : m;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const bool vec_a = (cin & 3) == 0, vec_b = (cout & 3) == 0;
  f32x4 acc[KPW][CIT][COT];
#pragma unroll
  for (int kk = 0; kk < KPW; ++kk)
#pragma unroll
    for (int a = 0; a < CIT; ++a)
#pragma unroll
      for (int b = 0; b < COT; ++b) acc[kk][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // 4 consecutive elements starting at channel c of row `row` (zeros past `width`), widened to fp32
  auto load4 = [&](const T* base, int row, int width, int c, bool vec) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < 0 || c >= width) return v;
    const T* p = base + (size_t)row * width + c;
    if (vec) {
      if constexpr (DT == DT_F32) {
        v = *(const float4*)p;
      } else {
        const uint2 raw = *(const uint2*)p;
        const T* e = (const T*)&raw;
        v = make_float4(Elem<DT>::to_f32(e[0]), Elem<DT>::to_f32(e[1]), Elem<DT>::to_f32(e[2]), Elem<DT>::to_f32(e[3]));
      }
    } else {
      float* f = (float*)&v;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (c + j < width) f[j] = Elem<DT>::to_f32(p[j]);
    }
    return v;
  };

  int it = 0;
  for (int r0 = rbeg; r0 < rend; r0 += R, ++it) {
    const int buf = it & 1;
    if (tid < R * B4) {   // out_grad tile -> LDS
      const int row = tid / B4, c4 = tid % B4;
      const int o = r0 + row;
      *(float4*)&fb[buf][row * LDB + c4 * 4] = load4(gout, o < rend ? o : -1, cout, co0 + c4 * 4, vec_b);
    }
    int idx[KPW];   // neighbour rows of this wave's offsets for the tile's rows (lane & 31 = row)
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
      const int k = w + kk * WG2_NW, o = r0 + (lane & 31);
      idx[kk] = (k < K && o < rend) ? nbr[(size_t)k * nbr_stride + o] : -1;
    }
    __syncthreads();   // the tile's out_grad is in place; (double buffer: nobody still reads the buffer the NEXT tile overwrites)
    wg_u32x4 bh[COT], bm[COT], bl[COT];   // X3: out_grad fragments of the tile (rows e*4 + l4 of channel t*16 + l15), split once
    if constexpr (X3) {
#pragma unroll
      for (int t = 0; t < COT; ++t) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fb[buf][(e * 4 + l4) * LDB + t * 16 + l15];
        wg_split3(v, bh[t], bm[t], bl[t]);
      }
    }
    float4 nx[NLD];
    auto gather = [&](int kk) {
#pragma unroll
      for (int p = 0; p < NLD; ++p) {
        const int e = p * 64 + lane, row = e / A4, c4 = e % A4;
        const int src = __shfl(idx[kk], row, 64);
        nx[p] = load4(feat, src, cin, ci0 + c4 * 4, vec_a);
      }
    };
    gather(0);
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
      const bool live = __ballot(idx[kk] >= 0) != 0ull;   // wave-uniform: some row of the tile has a neighbour at this offset
      if (live) {
#pragma unroll
        for (int p = 0; p < NLD; ++p) {
          const int e = p * 64 + lane, row = e / A4, c4 = e % A4;
          *(float4*)&fa[w][row * LDA + c4 * 4] = nx[p];
        }
      }
      if (kk + 1 < KPW) gather(kk + 1);   // in flight under the MFMAs below
      if (live) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's tile is written (wave-private: no barrier)
        __builtin_amdgcn_wave_barrier();
        if constexpr (X3) {
#pragma unroll
          for (int ta = 0; ta < CIT; ++ta) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fa[w][(e * 4 + l4) * LDA + ta * 16 + l15];
            wg_u32x4 ah, am, al;
            wg_split3(v, ah, am, al);
#pragma unroll
            for (int tb = 0; tb < COT; ++tb) {
              f32x4 c = acc[kk][ta][tb];
              c = wg_mfma_bf16(al, bh[tb], c);   // smallest terms first
              c = wg_mfma_bf16(am, bm[tb], c);
              c = wg_mfma_bf16(ah, bl[tb], c);
              c = wg_mfma_bf16(am, bh[tb], c);
              c = wg_mfma_bf16(ah, bm[tb], c);
              c = wg_mfma_bf16(ah, bh[tb], c);
              acc[kk][ta][tb] = c;
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < R / 4; ++q) {
            const int r = q * 4 + l4;
            float a[CIT], b[COT];
#pragma unroll
            for (int t = 0; t < CIT; ++t) a[t] = fa[w][r * LDA + t * 16 + l15];
#pragma unroll
            for (int t = 0; t < COT; ++t) b[t] = fb[buf][r * LDB + t * 16 + l15];
#pragma unroll
            for (int ta = 0; ta < CIT; ++ta)
#pragma unroll
              for (int tb = 0; tb < COT; ++tb) acc[kk][ta][tb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[tb], acc[kk][ta][tb], 0, 0, 0);
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next offset overwrites the tile
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  // D[i = ci][j = co]: lane holds column co = l15, rows ci = l4 * 4 + e of a tile
#pragma unroll
  for (int kk = 0; kk < KPW; ++kk) {
    const int k = w + kk * WG2_NW;
    if (k >= K) continue;
    float* dst = part + ((size_t)s * K + k) * cinp_tot * coutp_tot;
#pragma unroll
    for (int ta = 0; ta < CIT; ++ta)
#pragma unroll
      for (int tb = 0; tb < COT; ++tb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          dst[(size_t)(ci0 + ta * 16 + l4 * 4 + e) * coutp_tot + co0 + tb * 16 + l15] = acc[kk][ta][tb][e];
  }
}

// The same row-tile-stationary filter gradient for 16-BIT features (fp16 / bf16: the reference's default training arithmetic,
// configs/default.yaml fp16 -> autocast: half operands, fp32 accumulation) on v_mfma_f32_16x16x32_{f16,bf16}: one MFMA covers all
// 32 rows of a tile (the fp32 flavour needs eight v_mfma_f32_16x16x4_f32 at half the rate each — it is bound by the fp32 MFMA
// rate at 64 channels and above).  The reduction index of this GEMM is the ROW, but rows are what lie apart in memory: the
// tiles stay row-major in LDS in their storage type (8-byte pieces, linear writes) and the operands are read with
// ds_read_b64_tr_b16, the transposing LDS read — lane (c = l & 15, g = l >> 4) supplies the address of row 8g + (c >> 2), columns
// 4 (c & 3) .. +3, and receives rows 8g .. 8g + 3 of column c (probed on the hardware: tools/ubench/tr16_probe.hip): two reads give
// the 8 consecutive reduction elements of an A (X^T) or B (gY) fragment.  Products of 16-bit inputs are exact in fp32.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int DT>
__device__ __forceinline__ f32x4 mfma16(const s16x8& a, const s16x8& b, f32x4 acc) {
  if constexpr (DT == DT_F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

template <int DT, int CIT, int COT, int KPW, int NW = WG2_NW>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void spconv_wgrad16_kernel(const typename Elem<DT>::T* __restrict__ feat,
                                                                const typename Elem<DT>::T* __restrict__ gout,
                                                                const int* __restrict__ nbr, int nbr_stride, int m, int K,
                                                                int cin, int cout, int cinp_tot, int coutp_tot,
                                                                int rows_per_slab, float* __restrict__ part) {
  static_assert(DT == DT_F16 || DT == DT_BF16, "16-bit features");
  typedef uint16_t H;   // raw 16-bit storage
  constexpr int R = WG2_R, CIB = CIT * 16, COB = COT * 16;
  constexpr int A8 = CIB / 8, B8 = COB / 8;          // 16-byte (8-element) pieces per row
  constexpr int NLD = (R * A8 + 63) / 64;            // gathered pieces per lane and offset (R * A8 = 64 | 128)
  static_assert(R * A8 % 64 == 0, "whole wave instructions");
  __shared__ __attribute__((aligned(16))) H fb[2][R * COB];
  __shared__ __attribute__((aligned(16))) H fa[NW][R * CIB];
  const H* featb = (const H*)feat;
  const H* goutb = (const H*)gout;
  const int s = blockIdx.x, ci0 = blockIdx.y * CIB, co0 = blockIdx.z * COB;
  const int rbeg = s * rows_per_slab;
  const int rend = rbeg + rows_per_slab < m ? rbeg + rows_per_slab : m;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const bool vec_a = (cin & 7) == 0, vec_b = (cout & 7) == 0;
  f32x4 acc[KPW][CIT][COT];
#pragma unroll
  for (int kk = 0; kk < KPW; ++kk)
#pragma unroll
    for (int a = 0; a < CIT; ++a)
#pragma unroll
      for (int b = 0; b < COT; ++b) acc[kk][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // 8 consecutive 16-bit elements starting at channel c of row `row` (zeros past `width` / for row < 0), as stored
  auto load8 = [&](const H* base, int row, int width, int c, bool vec) -> uint4 {
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (row < 0 || c >= width) return v;
    const H* p = base + (size_t)row * width + c;
    if (vec) {
      v = *(const uint4*)p;
    } else {
      H* e = (H*)&v;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (c + j < width) e[j] = p[j];
    }
    return v;
  };
  // fragment of a 16-column tile starting at column c0 of a row-major [R][PITCH] tile: rows 8*l4 .. 8*l4 + 7 of column c0 + l15
  auto frag = [&](const H* tile, int pitch, int c0) -> s16x8 {
    const H* p = tile + (l4 * 8 + (l15 >> 2)) * pitch + c0 + (l15 & 3) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * pitch));
    return s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  // Everything a tile needs from memory is requested ONE TILE AHEAD: the neighbour rows of this wave's offsets, the gathered
  // pieces of all its KPW offsets (registers nx), the out_grad piece of this thread.  A piece register is refilled for the next
  // tile as soon as it has been written to LDS, so the gathers of tile t+1 are in flight under the MFMAs of tile t and a wave
  // exposes one memory round trip per slab, not one per (tile, offset).
  auto load_idx = [&](int r0, int (&ix)[KPW]) {
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
      const int k = w + kk * NW, o = r0 + (lane & 31);
      ix[kk] = (k < K && o < rend) ? nbr[(size_t)k * nbr_stride + o] : -1;
    }
  };
  auto gather = [&](int ixk, uint4 (&dst)[NLD]) {
#pragma unroll
    for (int p = 0; p < NLD; ++p) {
      const int e = p * 64 + lane, row = e / A8, c8 = e % A8;
      const int src = __shfl(ixk, row, 64);
      dst[p] = load8(featb, src, cin, ci0 + c8 * 8, vec_a);
    }
  };
  auto load_gy = [&](int r0) -> uint4 {
    if (tid >= R * B8) return make_uint4(0u, 0u, 0u, 0u);
    const int row = tid / B8, c8 = tid % B8, o = r0 + row;
    return load8(goutb, o < rend ? o : -1, cout, co0 + c8 * 8, vec_b);
  };

  int idx[KPW], idx_next[KPW];
  uint4 nx[KPW][NLD];
  load_idx(rbeg, idx);
  uint4 gy = load_gy(rbeg);
#pragma unroll
  for (int kk = 0; kk < KPW; ++kk) gather(idx[kk], nx[kk]);
  int it = 0;
  for (int r0 = rbeg; r0 < rend; r0 += R, ++it) {
    const int buf = it & 1;
    load_idx(r0 + R, idx_next);                         // rows past the slab give -1: the refills below become no-ops
    if (tid < R * B8) *(uint4*)&fb[buf][(tid / B8) * COB + (tid % B8) * 8] = gy;
    gy = load_gy(r0 + R);
    __syncthreads();   // the tile's out_grad is in place (double buffer: nobody still reads the buffer the NEXT tile overwrites)
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
      const bool live = __ballot(idx[kk] >= 0) != 0ull;   // wave-uniform
      if (live) {
#pragma unroll
        for (int p = 0; p < NLD; ++p) {
          const int e = p * 64 + lane, row = e / A8, c8 = e % A8;
          *(uint4*)&fa[w][row * CIB + c8 * 8] = nx[kk][p];
        }
      }
      gather(idx_next[kk], nx[kk]);   // tile t+1, in flight under everything below
      if (live) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's tile is written (wave-private: no barrier)
        __builtin_amdgcn_wave_barrier();
        s16x8 a[CIT], b[COT];
#pragma unroll
        for (int t = 0; t < CIT; ++t) a[t] = frag(fa[w], CIB, t * 16);
#pragma unroll
        for (int t = 0; t < COT; ++t) b[t] = frag(fb[buf], COB, t * 16);
#pragma unroll
        for (int ta = 0; ta < CIT; ++ta)
#pragma unroll
          for (int tb = 0; tb < COT; ++tb) acc[kk][ta][tb] = mfma16<DT>(a[ta], b[tb], acc[kk][ta][tb]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads are done before the next offset overwrites the tile
        __builtin_amdgcn_wave_barrier();
      }
    }
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) idx[kk] = idx_next[kk];
  }
#pragma unroll
  for (int kk = 0; kk < KPW; ++kk) {
    const int k = w + kk * NW;
    if (k >= K) continue;
    float* dst = part + ((size_t)s * K + k) * cinp_tot * coutp_tot;
#pragma unroll
    for (int ta = 0; ta < CIT; ++ta)
#pragma unroll
      for (int tb = 0; tb < COT; ++tb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          dst[(size_t)(ci0 + ta * 16 + l4 * 4 + e) * coutp_tot + co0 + tb * 16 + l15] = acc[kk][ta][tb][e];
  }
}

template <int DT>
__global__ __launch_bounds__(256) void spconv_wgrad_reduce_kernel(const float* __restrict__ part, int nslabs, int K, int cin,
                                                                  int cout, int cin_pad, int cout_pad,
                                                                  typename Elem<DT>::T* __restrict__ gw) {
  const size_t total = (size_t)K * cin * cout;
  const size_t slab = (size_t)K * cin_pad * cout_pad;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int co = (int)(i % cout);
    const size_t t = i / cout;
    const int ci = (int)(t % cin), k = (int)(t / cin);
    const size_t off = ((size_t)k * cin_pad + ci) * cout_pad + co;
    float v = 0.f;
    for (int s = 0; s < nslabs; ++s) v += part[s * slab + off];   // fixed order
    gw[i] = Elem<DT>::from_f32(v);
  }
}

static int round_up(int x, int a) { return (x + a - 1) / a * a; }
// output-channel tiles per wave are instantiated for 1, 2, 4, 8 (x16 channels); the prepared filter
// image is padded to that many rows so that every B-fragment load stays in bounds.
static int rows_padded(int rows) {
  int nt = (rows + 15) / 16;
  int sel = nt <= 1 ? 1 : nt <= 2 ? 2 : nt <= 4 ? 4 : nt <= 8 ? 8 : nt;
  return sel * 16;
}

static int elem_size(int dt) { return dt == DT_F32 ? 4 : 2; }
static int chunk_of(int dt) { return dt == DT_F32 ? 16 : 32; }

template <int DT>
static int launch_fwd(const void* feat, const void* wt, const int* nbr, int nbr_stride, int m_cap, const int* m_dev,
                      int K, int cin, int cout, void* out, const Epilogue& ep, hipStream_t stream) {
  typedef typename Elem<DT>::T T;
  const int cin_pad = round_up(cin, Elem<DT>::CH);
  const int ntiles = (cout + 15) / 16;
  // rows per wave: amortise the filter reads over more rows when there are plenty of rows
  const int mt = m_cap >= 65536 ? 2 : 1;
  const int rows_per_block = 4 * 16 * mt;
  dim3 grid(cdiv(m_cap, rows_per_block)), block(256);
#define BEVAMD_SPCONV(NT, MT)                                                                                    \
  spconv_fwd_kernel<DT, NT, MT><<<grid, block, 0, stream>>>((const T*)feat, (const T*)wt, nbr, nbr_stride, m_cap, \
                                                           m_dev, K, cin, cin_pad, cout, (T*)out, ep)
#define BEVAMD_SPCONV_MT(NT) \
  do { if (mt == 2) BEVAMD_SPCONV(NT, 2); else BEVAMD_SPCONV(NT, 1); } while (0)
  if (ntiles <= 1) BEVAMD_SPCONV_MT(1);
  else if (ntiles <= 2) BEVAMD_SPCONV_MT(2);
  else if (ntiles <= 4) BEVAMD_SPCONV_MT(4);
  else if (ntiles <= 8) BEVAMD_SPCONV_MT(8);
  else {
    set_error("spconv_conv_forward: cout=%d > 128 is not supported", cout);
    return BEVAMD_ERR_UNSUPPORTED;
  }
#undef BEVAMD_SPCONV_MT
#undef BEVAMD_SPCONV
  BEVAMD_LAUNCH_CHECK("spconv_fwd");
  return BEVAMD_OK;
}

}  // namespace bevamd

#include "spconv_wgrad_slab.h"

using namespace bevamd;

extern "C" {

/* number of ELEMENTS of the prepared filter image for a conv with K offsets, cin -> cout */
size_t bevamd_spconv_prepared_filter_elems(int dtype, int kernel_volume, int cin, int cout, int transpose_io) {
  int rows = transpose_io ? cin : cout, cols = transpose_io ? cout : cin;
  return (size_t)rows_padded(rows) * kernel_volume * round_up(cols, chunk_of(dtype));
}

int bevamd_spconv_prepare_filters(const void* filters, int dtype, int kernel_volume, int cin, int cout,
                                  int transpose_io, void* prepared, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(dtype >= 0 && dtype <= 2, "spconv_prepare_filters: bad dtype %d", dtype);
  BEVAMD_REQUIRE(kernel_volume > 0 && cin > 0 && cout > 0, "spconv_prepare_filters: bad sizes");
  BEVAMD_REQUIRE(filters && prepared, "spconv_prepare_filters: null buffer");
  int rows = transpose_io ? cin : cout, cols = transpose_io ? cout : cin;
  int rows_pad = rows_padded(rows), cols_pad = round_up(cols, chunk_of(dtype));
  size_t total = (size_t)rows_pad * kernel_volume * cols_pad;
  dim3 grid((unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096)), block(256);
  switch (dtype) {
    case DT_F32: spconv_prepare_filters_kernel<DT_F32><<<grid, block, 0, stream>>>((const float*)filters, kernel_volume, cin, cout, rows_pad, cols_pad, transpose_io, (float*)prepared); break;
    case DT_F16: spconv_prepare_filters_kernel<DT_F16><<<grid, block, 0, stream>>>((const _Float16*)filters, kernel_volume, cin, cout, rows_pad, cols_pad, transpose_io, (_Float16*)prepared); break;
    default:     spconv_prepare_filters_kernel<DT_BF16><<<grid, block, 0, stream>>>((const uint16_t*)filters, kernel_volume, cin, cout, rows_pad, cols_pad, transpose_io, (uint16_t*)prepared); break;
  }
  BEVAMD_LAUNCH_CHECK("spconv_prepare_filters");
  return BEVAMD_OK;
}

/* out[o,:] = epilogue( sum_k features[nbr[k][o], :] @ W[k] ).  `prepared` comes from
 * bevamd_spconv_prepare_filters(transpose_io=0).  num_out rows (or *num_out_dev if non-null, with
 * num_out as the launch bound).  Epilogue operands may be NULL. */
int bevamd_spconv_conv_forward(const void* features, int dtype, const void* prepared, const int* nbr, int nbr_stride,
                               int num_out, const int* num_out_dev, int kernel_volume, int cin, int cout, void* out,
                               const void* bias, const float* bn_scale, const float* bn_shift, const void* residual,
                               int relu, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(dtype >= 0 && dtype <= 2, "spconv_conv_forward: bad dtype %d", dtype);
  BEVAMD_REQUIRE(kernel_volume > 0 && cin > 0 && cout > 0 && num_out >= 0, "spconv_conv_forward: bad sizes");
  if (num_out == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(features && prepared && nbr && out, "spconv_conv_forward: null buffer");
  BEVAMD_REQUIRE(nbr_stride >= num_out, "spconv_conv_forward: nbr_stride %d < num_out %d", nbr_stride, num_out);
  Epilogue ep{bias, bn_scale, bn_shift, residual, relu};
  switch (dtype) {
    case DT_F32: return launch_fwd<DT_F32>(features, prepared, nbr, nbr_stride, num_out, num_out_dev, kernel_volume, cin, cout, out, ep, stream);
    case DT_F16: return launch_fwd<DT_F16>(features, prepared, nbr, nbr_stride, num_out, num_out_dev, kernel_volume, cin, cout, out, ep, stream);
    default:     return launch_fwd<DT_BF16>(features, prepared, nbr, nbr_stride, num_out, num_out_dev, kernel_volume, cin, cout, out, ep, stream);
  }
}

/* filter_grad [K, cin, cout] (same dtype as features) = sum over the rulebook of features^T @ out_grad
 * (sparse_conv_ext.indice_conv_backward_*'s filter half, spconv_ops.h:363-456).  nbr is the FORWARD table (nbr[k][o] = input
 * row).  MFMA (exact fp32), no atomics, bit-reproducible.  ws: WG_MAX_SLABS * K * cin_pad * cout_pad fp32 slab partials. */
// geometry of the row-tile-stationary filter gradient for (K, cin, cout): channel blocks, padded widths, most slabs it will use
struct Wgrad2Plan { int cit, cot, nci, nco, cinp, coutp, kpw, max_slabs, nw; };
// `wide` (16-bit features, cin >= 32, cout >= 64, K > 8): a workgroup of EIGHT waves owns 64 output channels (COT = 4) and each wave
// four kernel offsets.  Every (input block, output block) workgroup gathers its 32-channel slice of a neighbour row per offset,
// so the gathered bytes of a layer are rows x offsets x row bytes x (output blocks): what bounds the kernel (64 -> 64 at 4
// frames: 2.7 GB of 64-byte gathers from L2 in 640 us).  Twice the output channels per workgroup = half the gathers, twice the
// MFMAs per gathered tile.
static bool wgrad2_plan(int K, int cin, int cout, Wgrad2Plan& p, bool wide = false) {
  if (K > 7 * WG2_NW || cin > 128 || cout > 128) return false;
  wide = wide && cin >= 32 && cout >= 64 && K > 8 && K <= 32;
  p.nw = wide ? 8 : WG2_NW;
  p.cit = cin <= 16 ? 1 : 2;
  p.cot = wide ? 4 : cout <= 16 ? 1 : 2;
  p.nci = (cin + p.cit * 16 - 1) / (p.cit * 16);
  p.nco = (cout + p.cot * 16 - 1) / (p.cot * 16);
  p.cinp = p.nci * p.cit * 16;
  p.coutp = p.nco * p.cot * 16;
  const int need = (K + p.nw - 1) / p.nw;
  p.kpw = wide ? 4 : need <= 1 ? 1 : need <= 2 ? 2 : 7;
  int sl = (wide ? WG2_TARGET_WGS / 2 : WG2_TARGET_WGS) / (p.nci * p.nco);
  p.max_slabs = sl < 1 ? 1 : sl > WG2_MAX_SLABS ? WG2_MAX_SLABS : sl;
  return true;
}
// BEVAMD_SPCONV_F32X3: 0 never, 1 whenever served, anything else (default) auto = from 4 096 rows on (ops.py reads the same variable)
static int f32x3_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("BEVAMD_SPCONV_F32X3");
    mode = !e ? 2 : (e[0] == '0' && !e[1]) ? 0 : (e[0] == '1' && !e[1]) ? 1 : 2;
  }
  return mode;
}
static bool wgrad_wide_enabled() {
  static int on = -1;
  if (on < 0) { const char* e = getenv("BEVAMD_SPCONV_WGRAD_WIDE"); on = e ? atoi(e) != 0 : 1; }
  return on != 0;
}

size_t bevamd_spconv_wgrad_workspace_bytes(int kernel_volume, int cin, int cout) {
  if (kernel_volume <= 0 || cin <= 0 || cout <= 0 || cin > 128 || cout > 128) return 0;
  Wgrad2Plan p, pw;
  if (wgrad2_plan(kernel_volume, cin, cout, p)) {
    size_t b = (size_t)p.max_slabs * kernel_volume * p.cinp * p.coutp * sizeof(float);
    if (wgrad2_plan(kernel_volume, cin, cout, pw, true)) {   // the 16-bit wide plan may use more slabs: size for either
      const size_t bw = (size_t)pw.max_slabs * kernel_volume * pw.cinp * pw.coutp * sizeof(float);
      b = bw > b ? bw : b;
    }
    return align_up(b, 256);
  }
  const int cit = (cin + 15) / 16, cinp = (cit <= 1 ? 1 : cit <= 2 ? 2 : cit <= 4 ? 4 : 8) * 16;
  const int coutp = (cout + 15) / 16 * 16;
  return align_up((size_t)WG_MAX_SLABS * kernel_volume * cinp * coutp * sizeof(float), 256);
}

int bevamd_spconv_conv_wgrad(const void* features, const void* out_grad, int dtype, const int* nbr, int nbr_stride,
                             int num_out, int kernel_volume, int cin, int cout, void* filter_grad, void* ws,
                             size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(dtype >= 0 && dtype <= 2, "spconv_conv_wgrad: bad dtype %d", dtype);
  BEVAMD_REQUIRE(kernel_volume > 0 && cin > 0 && cout > 0 && num_out >= 0, "spconv_conv_wgrad: bad sizes");
  BEVAMD_REQUIRE(cin <= 128 && cout <= 128, "spconv_conv_wgrad: %d -> %d channels (<= 128 supported)", cin, cout);
  BEVAMD_REQUIRE(filter_grad != nullptr, "spconv_conv_wgrad: filter_grad is null");
  const size_t nw = (size_t)kernel_volume * cin * cout;
  if (num_out == 0) {
    BEVAMD_HIP_CHECK(hipMemsetAsync(filter_grad, 0, nw * elem_size(dtype), stream));
    return BEVAMD_OK;
  }
  BEVAMD_REQUIRE(features && out_grad && nbr, "spconv_conv_wgrad: null input");
  if (!ws || ws_bytes < bevamd_spconv_wgrad_workspace_bytes(kernel_volume, cin, cout)) {
    set_error("spconv_conv_wgrad: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  Wgrad2Plan wp;
  if (wgrad2_plan(kernel_volume, cin, cout, wp, dtype != DT_F32 && wgrad_wide_enabled())) {
    // row-tile-stationary kernel: slabs of whole 32-row tiles, ~768 workgroups over (slab, ci block, co block)
    int nslabs = (num_out + 4 * WG2_R - 1) / (4 * WG2_R);   // at least 4 tiles per slab
    if (nslabs > wp.max_slabs) nslabs = wp.max_slabs;
    if (nslabs < 1) nslabs = 1;
    int rows_per_slab = ((num_out + nslabs - 1) / nslabs + WG2_R - 1) / WG2_R * WG2_R;
    nslabs = (num_out + rows_per_slab - 1) / rows_per_slab;
    float* part = (float*)ws;
    dim3 grid(nslabs, wp.nci, wp.nco), block(wp.nw * 64);
    // fp32 rows from 4 096 on: the products on the bf16 matrix cores by three-way operand splitting (X3; BEVAMD_SPCONV_F32X3 = 0 | 1 | auto
    // as for the forward / input gradient: one switch for the fp32 training path)
    const bool x3 = dtype == DT_F32 && f32x3_mode() != 0 && (f32x3_mode() == 1 || num_out >= 4096);
#define BEVAMD_WG2(DT, T, CIT, COT, KPW)                                                                                              \
  do {                                                                                                                               \
    if (x3) /* this macro is only expanded for DT_F32 */                                                                             \
      spconv_wgrad2_kernel<DT, CIT, COT, KPW, true><<<grid, block, 0, stream>>>((const T*)features, (const T*)out_grad, nbr, nbr_stride, num_out, \
                                                                                        kernel_volume, cin, cout, wp.cinp, wp.coutp, rows_per_slab, part); \
    else                                                                                                                             \
      spconv_wgrad2_kernel<DT, CIT, COT, KPW><<<grid, block, 0, stream>>>((const T*)features, (const T*)out_grad, nbr, nbr_stride, num_out, \
                                                                          kernel_volume, cin, cout, wp.cinp, wp.coutp, rows_per_slab, part); \
  } while (0)
#define BEVAMD_WG2_K(DT, T, CIT, COT)                                                          \
  do {                                                                                        \
    if (wp.kpw == 1) BEVAMD_WG2(DT, T, CIT, COT, 1); else if (wp.kpw == 2) BEVAMD_WG2(DT, T, CIT, COT, 2); \
    else BEVAMD_WG2(DT, T, CIT, COT, 7);                                                      \
  } while (0)
#define BEVAMD_WG2_C(DT, T)                                                                    \
  do {                                                                                        \
    if (wp.cit == 1 && wp.cot == 1) BEVAMD_WG2_K(DT, T, 1, 1);                                \
    else if (wp.cit == 1) BEVAMD_WG2_K(DT, T, 1, 2);                                          \
    else if (wp.cot == 1) BEVAMD_WG2_K(DT, T, 2, 1);                                          \
    else BEVAMD_WG2_K(DT, T, 2, 2);                                                           \
  } while (0)
#define BEVAMD_WG16(DT, T, CIT, COT, KPW) \
  spconv_wgrad16_kernel<DT, CIT, COT, KPW><<<grid, block, 0, stream>>>((const T*)features, (const T*)out_grad, nbr, nbr_stride, num_out, \
                                                                       kernel_volume, cin, cout, wp.cinp, wp.coutp, rows_per_slab, part)
#define BEVAMD_WG16_K(DT, T, CIT, COT)                                                         \
  do {                                                                                        \
    if (wp.kpw == 1) BEVAMD_WG16(DT, T, CIT, COT, 1); else if (wp.kpw == 2) BEVAMD_WG16(DT, T, CIT, COT, 2); \
    else BEVAMD_WG16(DT, T, CIT, COT, 7);                                                     \
  } while (0)
#define BEVAMD_WG16_C(DT, T)                                                                   \
  do {                                                                                        \
    if (wp.nw == 8)                                                                           \
      spconv_wgrad16_kernel<DT, 2, 4, 4, 8><<<grid, block, 0, stream>>>((const T*)features, (const T*)out_grad, nbr, nbr_stride, num_out, \
                                                                        kernel_volume, cin, cout, wp.cinp, wp.coutp, rows_per_slab, part); \
    else if (wp.cit == 1 && wp.cot == 1) BEVAMD_WG16_K(DT, T, 1, 1);                          \
    else if (wp.cit == 1) BEVAMD_WG16_K(DT, T, 1, 2);                                         \
    else if (wp.cot == 1) BEVAMD_WG16_K(DT, T, 2, 1);                                         \
    else BEVAMD_WG16_K(DT, T, 2, 2);                                                          \
  } while (0)
    if (dtype == DT_F32) BEVAMD_WG2_C(DT_F32, float);
    else if (dtype == DT_F16) BEVAMD_WG16_C(DT_F16, _Float16);
    else BEVAMD_WG16_C(DT_BF16, uint16_t);
#undef BEVAMD_WG16_C
#undef BEVAMD_WG16_K
#undef BEVAMD_WG16
#undef BEVAMD_WG2_C
#undef BEVAMD_WG2_K
#undef BEVAMD_WG2
    BEVAMD_LAUNCH_CHECK("spconv_wgrad2");
    dim3 rgrid2((unsigned)((nw + 255) / 256 < 2048 ? (nw + 255) / 256 : 2048));
    if (dtype == DT_F32) spconv_wgrad_reduce_kernel<DT_F32><<<rgrid2, dim3(256), 0, stream>>>(part, nslabs, kernel_volume, cin, cout, wp.cinp, wp.coutp, (float*)filter_grad);
    else if (dtype == DT_F16) spconv_wgrad_reduce_kernel<DT_F16><<<rgrid2, dim3(256), 0, stream>>>(part, nslabs, kernel_volume, cin, cout, wp.cinp, wp.coutp, (_Float16*)filter_grad);
    else spconv_wgrad_reduce_kernel<DT_BF16><<<rgrid2, dim3(256), 0, stream>>>(part, nslabs, kernel_volume, cin, cout, wp.cinp, wp.coutp, (uint16_t*)filter_grad);
    BEVAMD_LAUNCH_CHECK("spconv_wgrad_reduce");
    return BEVAMD_OK;
  }
  const int cit0 = (cin + 15) / 16, cit = cit0 <= 1 ? 1 : cit0 <= 2 ? 2 : cit0 <= 4 ? 4 : 8;
  const int cinp = cit * 16, coutp = (cout + 15) / 16 * 16;
  // row slabs: enough workgroups to fill 256 CUs (x K offsets x channel groups), at most WG_MAX_SLABS partials to reduce
  int nslabs = (num_out + 2047) / 2048;
  if (nslabs < 1) nslabs = 1;
  if (nslabs > WG_MAX_SLABS) nslabs = WG_MAX_SLABS;
  int rows_per_slab = (num_out + nslabs - 1) / nslabs;
  rows_per_slab = (rows_per_slab + WG_ROWS - 1) / WG_ROWS * WG_ROWS;
  nslabs = (num_out + rows_per_slab - 1) / rows_per_slab;
  float* part = (float*)ws;
  dim3 grid(nslabs, kernel_volume, (coutp + WG_COG - 1) / WG_COG), block(256);
#define BEVAMD_WGRAD(DT, T, CIT) \
  spconv_wgrad_mfma_kernel<DT, CIT><<<grid, block, 0, stream>>>((const T*)features, (const T*)out_grad, nbr, nbr_stride, num_out, \
                                                                kernel_volume, cin, cout, coutp, rows_per_slab, part)
#define BEVAMD_WGRAD_DT(DT, T)                                                          \
  do {                                                                                  \
    if (cit == 1) BEVAMD_WGRAD(DT, T, 1); else if (cit == 2) BEVAMD_WGRAD(DT, T, 2);    \
    else if (cit == 4) BEVAMD_WGRAD(DT, T, 4); else BEVAMD_WGRAD(DT, T, 8);             \
  } while (0)
  if (dtype == DT_F32) BEVAMD_WGRAD_DT(DT_F32, float);
  else if (dtype == DT_F16) BEVAMD_WGRAD_DT(DT_F16, _Float16);
  else BEVAMD_WGRAD_DT(DT_BF16, uint16_t);
#undef BEVAMD_WGRAD_DT
#undef BEVAMD_WGRAD
  BEVAMD_LAUNCH_CHECK("spconv_wgrad_mfma");
  dim3 rgrid((unsigned)((nw + 255) / 256 < 2048 ? (nw + 255) / 256 : 2048));
  if (dtype == DT_F32) spconv_wgrad_reduce_kernel<DT_F32><<<rgrid, block, 0, stream>>>(part, nslabs, kernel_volume, cin, cout, cinp, coutp, (float*)filter_grad);
  else if (dtype == DT_F16) spconv_wgrad_reduce_kernel<DT_F16><<<rgrid, block, 0, stream>>>(part, nslabs, kernel_volume, cin, cout, cinp, coutp, (_Float16*)filter_grad);
  else spconv_wgrad_reduce_kernel<DT_BF16><<<rgrid, block, 0, stream>>>(part, nslabs, kernel_volume, cin, cout, cinp, coutp, (uint16_t*)filter_grad);
  BEVAMD_LAUNCH_CHECK("spconv_wgrad_reduce");
  return BEVAMD_OK;
}

/* The filter gradient of a 3x3x3 SUBMANIFOLD convolution over rows in ascending linear index, from slab metadata (hdr / slots of
 * bevamd_spconv_slab_build* with block_rows = 128, raw slots) instead of the neighbour table: the neighbour rows of a block are
 * staged in LDS once and gathered by the transposing LDS reads (csrc/spconv_wgrad_slab.h).  Replaces the filter half of
 * sparse_conv_ext.indice_conv_backward_half (spconv_ops.h:363-456).  16-bit features, cin == cout in {16, 32, 64, 128}; feature and
 * out_grad pitches in elements (multiples of 8, rows 16-byte aligned).  Deterministic (fixed-order slab partials). */
static unsigned long long* g_wgs_prof = nullptr;
void bevamd_spconv_wgrad_slab_set_profile_buffer(void* buf) { g_wgs_prof = (unsigned long long*)buf; }  /* -DBEVAMD_WGS_PROF builds */

int bevamd_spconv_wgrad_slab_supported(int dtype, int cin, int cout) {
  wgslab::Shape s;
  return (dtype == DT_F16 || dtype == DT_BF16) && wgslab::shape_for(cin, cout, s) ? 1 : 0;
}

/* the block_rows code (rows | slot format << 16) of the metadata the filter gradient of a cin -> cin layer reads; 0 = unsupported */
int bevamd_spconv_wgrad_slab_block_rows(int cin) {
  wgslab::Shape s;
  if (!wgslab::shape_for(cin, cin, s)) return 0;
  return wgslab::BM | ((s.cit == 2 ? slab::FMT_WG64 : slab::FMT_WG32) << slab::FMT_SHIFT);
}

size_t bevamd_spconv_wgrad_slab_workspace_bytes(int cin, int cout) {
  wgslab::Shape s;
  if (!wgslab::shape_for(cin, cout, s)) return 0;
  return align_up((size_t)wgslab::slabs_for(1 << 30, s) * 27 * cin * cout * sizeof(float), 256);
}

int bevamd_spconv_conv_wgrad_slab(const void* features, int feat_stride, int num_in, const void* out_grad, int og_stride,
                                  int dtype, const void* hdr, const void* slots, int block_rows, int num_out, int cin, int cout,
                                  void* filter_grad, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  wgslab::Shape sh;
  BEVAMD_REQUIRE(dtype == DT_F16 || dtype == DT_BF16, "spconv_conv_wgrad_slab: dtype %d is not 16-bit", dtype);
  BEVAMD_REQUIRE(wgslab::shape_for(cin, cout, sh), "spconv_conv_wgrad_slab: %d -> %d channels (cin in 16 | 32 | 64 | 128, cout == cin, or 2 cin up to 128)", cin, cout);
  BEVAMD_REQUIRE(block_rows == bevamd_spconv_wgrad_slab_block_rows(cin),
                 "spconv_conv_wgrad_slab: metadata code %d, %d channels want %d (bevamd_spconv_wgrad_slab_block_rows)", block_rows, cin,
                 bevamd_spconv_wgrad_slab_block_rows(cin));
  BEVAMD_REQUIRE(num_out >= 0 && num_in >= 0 && filter_grad, "spconv_conv_wgrad_slab: bad sizes / null filter_grad");
  const size_t nw = (size_t)27 * cin * cout;
  if (num_out == 0) {
    BEVAMD_HIP_CHECK(hipMemsetAsync(filter_grad, 0, nw * 2, stream));
    return BEVAMD_OK;
  }
  BEVAMD_REQUIRE(features && out_grad && hdr && slots, "spconv_conv_wgrad_slab: null buffer");
  BEVAMD_REQUIRE(feat_stride >= cin && feat_stride % 8 == 0 && og_stride >= cout && og_stride % 8 == 0 &&
                     (((uintptr_t)features | (uintptr_t)out_grad | (uintptr_t)slots) & 15) == 0,
                 "spconv_conv_wgrad_slab: pitches %d / %d must be multiples of 8 covering the channels, buffers 16-byte aligned", feat_stride, og_stride);
  BEVAMD_REQUIRE((unsigned long long)num_in * feat_stride * 2ull < 0x100000000ull && (unsigned long long)num_out * og_stride * 2ull < 0x100000000ull,
                 "spconv_conv_wgrad_slab: tensors must be smaller than 4 GiB (buffer descriptors)");
  if (!ws || ws_bytes < bevamd_spconv_wgrad_slab_workspace_bytes(cin, cout)) {
    set_error("spconv_conv_wgrad_slab: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  wgslab::Args a;
  a.feat = features; a.gout = out_grad; a.hdr = (const int2*)hdr; a.slots = (const uint16_t*)slots; a.part = (float*)ws;
  a.feat_stride = feat_stride; a.gout_stride = og_stride; a.n_in = num_in; a.m = num_out;
  a.nblk = (num_out + wgslab::BM - 1) / wgslab::BM;
  int nslabs = wgslab::slabs_for(a.nblk, sh);
  a.blocks_per_slab = (a.nblk + nslabs - 1) / nslabs;
  a.nslabs = (a.nblk + a.blocks_per_slab - 1) / a.blocks_per_slab;
  a.ncb = sh.nci * sh.nco; a.nco = sh.nco;
  a.cinp_tot = cin; a.coutp_tot = cout;
  const unsigned long long sb = (unsigned long long)a.nblk * wgslab::SLOT_BYTES;
  BEVAMD_REQUIRE(sb < 0x100000000ull, "spconv_conv_wgrad_slab: slot table of 4 GiB or more");
  a.slot_bytes = (unsigned)sb;
  a.prof = g_wgs_prof;
  const int rc = dtype == DT_F16 ? wgslab::launch<true>(a, sh, stream) : wgslab::launch<false>(a, sh, stream);
  if (rc != BEVAMD_OK) return rc;
  const int n = (int)nw;   // 27 * cin * cout: a multiple of 64
  if (dtype == DT_F16) wgslab::wgrad_slab_reduce_kernel<true><<<dim3((n / 4 + 15) / 16), dim3(256), 0, stream>>>(a.part, a.nslabs, n, (uint16_t*)filter_grad);
  else wgslab::wgrad_slab_reduce_kernel<false><<<dim3((n / 4 + 15) / 16), dim3(256), 0, stream>>>(a.part, a.nslabs, n, (uint16_t*)filter_grad);
  BEVAMD_LAUNCH_CHECK("wgrad_slab_reduce");
  return BEVAMD_OK;
}

}  // extern "C"
