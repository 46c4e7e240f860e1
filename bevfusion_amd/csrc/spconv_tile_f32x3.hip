// fp32 sparse convolution on the 16-bit matrix cores by OPERAND SPLITTING (round 5).
//
// Reference: spconv_ops.h:260-361 indiceConv<float> / :363-456 indiceConvBackward<float> (the fp32 training path: per-offset
// gather -> torch::mm_out -> scatter-add).  The exact-chain fp32 kernel of spconv_conv.hip reaches 42 TFLOP/s on real pairs —
// v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate on gfx950 — and is 31 of the 61 ms of an fp32 training step (forward +
// input gradient).  Here every fp32 value is cut into three bf16 pieces by TRUNCATION,
//     x = hi + mid + lo,   hi = x & 0xFFFF0000,  mid = (x - hi) & 0xFFFF0000,  lo = x - hi - mid
// — exact: 24 significand bits = 8 + 8 + 8, the subtractions are exact in fp32, bf16 has fp32's exponent range (an fp16 split
// would lose the small gradients of the backward pass) — and a product is the six largest of the nine piece products,
//     x * w  ~=  hi*hi + hi*mid + mid*hi + hi*lo + mid*mid + lo*hi        (dropped: mid*lo + lo*mid + lo*lo <= 3 * 2^-24 |x w|)
// six v_mfma_f32_16x16x32_bf16 with fp32 accumulation at 16 x the fp32 MFMA rate.  Error against the exact product: a few
// 2^-24, i.e. the same order as ONE fp32 rounding; tests hold it to the fp32 bar of the exact-chain kernel (2e-5 relative to
// 1 + max|ref| against float64) and to 1e-5 against that kernel.
//
// Shape of the kernel (the gather kernels of spconv_tile.h, specialised): output-stationary rows, the tile's neighbour table in
// wave-private LDS, rows gathered with raw buffer loads in load layout (four adjacent lanes = one row's 128 contiguous bytes of a
// 32-channel chunk), moved to MFMA layout with ds_bpermute, split in registers; the filter arrives pre-split as three
// fragment-ordered bf16 images (make_filter_image3) and is loaded straight into registers, one group of <= 4 output tiles ahead;
// no LDS staging of the filter, no workgroup barrier in the loop.  Flat reduction axis (offset, channel) in 32-wide chunks, two
// chunks per loop trip on two register sets.  Summation order is fixed: bit-reproducible.
#include "spconv_tile.h"

namespace bevamd {
namespace tile3 {

using namespace tile;

struct Args3 {
  const float* feat;      // [n_in, feat_stride] fp32
  const void* wimg;       // [chunk][3][nt][64 lanes][8] bf16: parts hi, mid, lo
  const int* nbr;         // [K, nbr_stride]
  const int* m_dev;
  float* out;             // [m, out_stride] fp32
  const float* bias;
  const float* scale;
  const float* shift;
  const float* residual;
  int feat_stride, n_in, nbr_stride, m_cap, K, cout, out_stride, res_stride, relu;
  unsigned wimg_bytes;
};

__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& w, const u32x4& x, f32x4 acc) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}

// eight fp32 values -> three packed bf16x8 (hi, mid, lo), by truncation (exact: see the header)
__device__ __forceinline__ void split3(const float (&v)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const unsigned b = __float_as_uint(v[e]);
    h[e] = b & 0xFFFF0000u;
    const float r1 = v[e] - __uint_as_float(h[e]);
    m[e] = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(m[e]);
    l[e] = __float_as_uint(r2);          // <= 8 significant bits: its upper half is the value
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {          // bytes [b3 b2 | a3 a2]: element 2p in the low half
    hi[p] = __builtin_amdgcn_perm(h[2 * p + 1], h[2 * p], 0x07060302u);
    mid[p] = __builtin_amdgcn_perm(m[2 * p + 1], m[2 * p], 0x07060302u);
    lo[p] = __builtin_amdgcn_perm(l[2 * p + 1], l[2 * p], 0x07060302u);
  }
}

// CINP: channels of a feature row (16 | 32 | 64 | 128); NT: 16-channel output tiles (1 | 2 | 4 | 8); MT: 16-row tiles per wave
template <int CINP, int NT, int MT, int NW>
__global__ __launch_bounds__(NW * 64) void spconv_f32x3_kernel(Args3 a) {
  constexpr int CPB = Chunks<CINP>::CPB;               // 32-element chunks per kernel offset (CINP >= 32); below, a chunk spans 32 / CINP offsets
  constexpr int NTG = NT < 4 ? NT : 4;                 // output tiles per filter group (what sits in registers at once)
  constexpr int NG = NT / NTG;
  constexpr int ROWS = 16 * MT;
  extern __shared__ int lds_nb[];                      // [NW][K][ROWS]
  const int m = a.m_dev ? (*a.m_dev < a.m_cap ? *a.m_dev : a.m_cap) : a.m_cap;
  const int ntiles = (m + ROWS - 1) / ROWS;
  const int nblk = (ntiles + NW - 1) / NW;
  // XCD-aware block map: XCD x walks a contiguous range of row tiles (neighbouring tiles gather the same rows from its L2)
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per = (nblk + 7) >> 3;
  const int blk = xcd * per + bix;
  if (bix >= per || blk >= nblk) return;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int tile = blk * NW + w;
  if (tile >= ntiles) return;                          // no workgroup barrier below: a wave may leave alone
  const int row0 = tile * ROWS;
  const int c = lane & 15, g = lane >> 4;              // MFMA layout: row of the tile, k-group
  const int lr = lane >> 2, lg = lane & 3;             // load layout: row, k-group (4 adjacent lanes = 128 contiguous bytes)
  const int perm_addr = ((lane & 15) * 4 + (lane >> 4)) * 4;
  int* nbl = lds_nb + (size_t)w * a.K * ROWS;

  // the tile's neighbour table -> LDS in one burst (it is streamed exactly once: one cold-miss latency per tile, not per step)
  {
    const int total = a.K * ROWS;
    for (int base = 0; base < total; base += 64 * 8) {
      int v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int idx = base + u * 64 + lane;
        idx = idx < total ? idx : total - 1;
        const int k = idx / ROWS, r = idx - k * ROWS;
        const int row = row0 + r;
        v[u] = a.nbr[(size_t)k * a.nbr_stride + (row < m ? row : m - 1)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 64 + lane;
        if (idx < total) nbl[idx] = v[u];
      }
    }
  }

  const unsigned row_bytes = (unsigned)a.feat_stride * 4u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.feat, 0, (unsigned)a.n_in * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wimg, 0, a.wimg_bytes, 0x00020000);
  const int nchunks = (Chunks<CINP>::count(a.K) + 1) & ~1;   // an even number of trips; the image is zero-padded far beyond it

  f32x4 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};

  // rows of chunk j, load layout: this lane's 8 channels of row lr of every tile (two 16-byte loads; OOB offsets return zeros)
  auto gather = [&](int j, u32x4 (&x)[MT][2]) {
    const int k = CINP >= 32 ? j / CPB : (j * 32 + lg * 8) / CINP;   // the kernel offset this lane's 8 channels belong to
    const int kc = k < a.K ? k : a.K - 1;
    const unsigned elem = CINP >= 32 ? (unsigned)((j % CPB) * 32 + lg * 8) : (unsigned)((j * 32 + lg * 8) % CINP);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int nb = nbl[kc * ROWS + mt * 16 + lr];
      const bool valid = row0 + mt * 16 + lr < m && nb >= 0 && k < a.K;
      const unsigned base = valid ? (unsigned)nb * row_bytes + elem * 4u : OOB;
      x[mt][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, base, 0, 0);
      x[mt][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, base + 16u, 0, 0);
    }
  };
  // filter fragments of chunk j, group q: 3 parts x NTG output tiles, coalesced 16-byte loads
  auto load_w = [&](int j, int q, u32x4 (&wf)[3][NTG]) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < NTG; ++t)
        wf[p][t] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)lane * 16u, (unsigned)(((j * 3 + p) * NT + q * NTG + t) * 1024), 0);
  };
  // load layout -> MFMA layout (source lane 4c + g), then the split
  auto to_operands = [&](const u32x4 (&x)[MT][2], u32x4 (&xh)[MT], u32x4 (&xm)[MT], u32x4 (&xl)[MT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float v[8];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          v[h * 4 + e] = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)x[mt][h][e]));
      split3(v, xh[mt], xm[mt], xl[mt]);
    }
  };
  auto multiply = [&](int q, const u32x4 (&wf)[3][NTG], const u32x4 (&xh)[MT], const u32x4 (&xm)[MT], const u32x4 (&xl)[MT]) {
#pragma unroll
    for (int t = 0; t < NTG; ++t)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4 s = acc[mt][q * NTG + t];
        s = mfma_bf16(wf[2][t], xh[mt], s);   // smallest terms first
        s = mfma_bf16(wf[1][t], xm[mt], s);
        s = mfma_bf16(wf[0][t], xl[mt], s);
        s = mfma_bf16(wf[1][t], xh[mt], s);
        s = mfma_bf16(wf[0][t], xm[mt], s);
        s = mfma_bf16(wf[0][t], xh[mt], s);
        acc[mt][q * NTG + t] = s;
      }
  };
  // one chunk: operands from the gathered set, its NG filter groups; the NEXT thing's filter group is requested first
  u32x4 wA[3][NTG], wB[3][NTG];
  auto chunk = [&](int j, const u32x4 (&x)[MT][2], u32x4 (&w0)[3][NTG], u32x4 (&w1)[3][NTG], bool more) {
    u32x4 xh[MT], xm[MT], xl[MT];
    to_operands(x, xh, xm, xl);
    if constexpr (NG == 1) {
      if (more) load_w(j + 1, 0, w1);
      multiply(0, w0, xh, xm, xl);
    } else {
      load_w(j, 1, w1);
      multiply(0, w0, xh, xm, xl);
      if (more) load_w(j + 1, 0, w0);
      multiply(1, w1, xh, xm, xl);
    }
  };

  u32x4 xA[MT][2], xB[MT][2];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's table is in LDS (wave-private: no barrier)
  gather(0, xA);
  load_w(0, 0, wA);
  for (int j = 0; j < nchunks; j += 2) {
    gather(j + 1, xB);
    if constexpr (NG == 1) chunk(j, xA, wA, wB, true);
    else chunk(j, xA, wA, wB, true);
    const bool more = j + 2 < nchunks;
    if (more) gather(j + 2, xA);
    if constexpr (NG == 1) chunk(j + 1, xB, wB, wA, more);
    else chunk(j + 1, xB, wA, wB, more);
  }

  // epilogue: lane (c, g) holds channels nt*16 + 4g .. + 3 of row row0 + mt*16 + c
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = row0 + mt * 16 + c;
    if (row >= m) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int col0 = nt * 16 + g * 4;
      if (col0 >= a.cout) continue;
      f32x4 v = acc[mt][nt];
      float* op = a.out + (size_t)row * a.out_stride + col0;
      const bool vec = col0 + 4 <= a.cout && ((a.out_stride | a.res_stride) & 3) == 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (col0 + e >= a.cout) break;
        float y = v[e];
        if (a.bias) y += a.bias[col0 + e];
        if (a.scale) y = __builtin_fmaf(y, a.scale[col0 + e], a.shift[col0 + e]);
        if (a.residual) y += a.residual[(size_t)row * a.res_stride + col0 + e];
        if (a.relu) y = y > 0.f ? y : (y != y ? y : 0.f);
        v[e] = y;
      }
      if (vec) *(f32x4*)op = v;
      else
        for (int e = 0; e < 4 && col0 + e < a.cout; ++e) op[e] = v[e];
    }
  }
}

// filters [K][cin][cout] fp32 -> image [chunk][3 parts][nt][lane][8] bf16: element e of lane (c = lane & 15, g = lane >> 4) of
// chunk j is part p of W[k][ci][nt*16 + c] with flat = 32 j + 8 g + e, k = flat / CINP, ci = flat % CINP (the flattening of
// spconv_tile.h's image); zero outside K / cin / cout and in the padding chunks.  transpose_io: input-gradient pass.
template <int CINP>
__global__ __launch_bounds__(256) void spconv_filter_image3_kernel(const float* __restrict__ w, int K, int cin, int cout,
                                                                   int nt_count, int nchunks_padded, int transpose_io,
                                                                   uint16_t* __restrict__ img) {
  const size_t per_part = (size_t)nt_count * 64 * 8;
  const size_t total = (size_t)nchunks_padded * per_part;            // elements of ONE part
  const int rows = transpose_io ? cin : cout, cols = transpose_io ? cout : cin;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int e = (int)(i & 7);
    const int lane = (int)((i >> 3) & 63);
    size_t t = i >> 9;
    const int nt = (int)(t % nt_count);
    const long long j = (long long)(t / nt_count);
    const int c = lane & 15, g = lane >> 4;
    const long long flat = j * 32 + g * 8 + e;
    const int k = (int)(flat / CINP), ci = (int)(flat % CINP);
    const int co = nt * 16 + c;
    float v = 0.f;
    if (k < K && ci < cols && co < rows) {
      const int wi = transpose_io ? co : ci, wo = transpose_io ? ci : co;
      v = w[((size_t)k * cin + wi) * cout + wo];
    }
    const unsigned hb = __float_as_uint(v) & 0xFFFF0000u;
    const float r1 = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mb);
    const size_t within = ((size_t)nt * 64 + lane) * 8 + e;
    uint16_t* base = img + (size_t)j * 3 * per_part + within;
    base[0] = (uint16_t)(hb >> 16);
    base[per_part] = (uint16_t)(mb >> 16);
    base[2 * per_part] = (uint16_t)(__float_as_uint(r2) >> 16);
  }
}

static inline int pad_cin3(int cin) { return cin == 16 || cin == 32 || cin == 64 || cin == 128 ? cin : 0; }

template <int CINP, int NT, int MT, int NW>
static int run(const Args3& a, hipStream_t stream) {
  const size_t lds = (size_t)NW * a.K * 16 * MT * sizeof(int);
  if (lds > 64 * 1024) {
    set_error("spconv f32x3: %zu bytes of neighbour tables exceed 64 KiB (K = %d)", lds, a.K);
    return BEVAMD_ERR_UNSUPPORTED;
  }
  const long long ntiles = ((long long)a.m_cap + 16 * MT - 1) / (16 * MT);
  const long long nblk = (ntiles + NW - 1) / NW;
  const long long blocks = (nblk + 7) / 8 * 8;
  spconv_f32x3_kernel<CINP, NT, MT, NW><<<dim3((unsigned)blocks), dim3(NW * 64), lds, stream>>>(a);
  BEVAMD_LAUNCH_CHECK("spconv_f32x3");
  return BEVAMD_OK;
}

template <int CINP>
static int run_cin(const Args3& a, int nt, hipStream_t stream) {
  switch (nt) {
    case 1: return run<CINP, 1, 2, 4>(a, stream);
    case 2: return run<CINP, 2, 2, 4>(a, stream);
    case 4: return run<CINP, 4, 2, 4>(a, stream);
    case 8: return run<CINP, 8, 2, 4>(a, stream);
    default: set_error("spconv f32x3: %d output tiles", nt); return BEVAMD_ERR_UNSUPPORTED;
  }
}

}  // namespace tile3
}  // namespace bevamd

using namespace bevamd;

extern "C" {

/* 1 if bevamd_spconv_conv_forward_f32x3 serves cin -> cout (fp32 rows of exactly 16 | 32 | 64 | 128 channels, cout <= 128) */
int bevamd_spconv_f32x3_supported(int cin, int cout) {
  return tile3::pad_cin3(cin) != 0 && cout > 0 && tile::pad_nt(cout) != 0;
}

/* ELEMENTS (uint16) of the three-part filter image of a K-offset cin -> cout convolution */
size_t bevamd_spconv_filter_image3_elems(int kernel_volume, int cin, int cout, int transpose_io) {
  const int rows = transpose_io ? cin : cout, cols = transpose_io ? cout : cin;
  const int cinp = tile3::pad_cin3(cols), nt = tile::pad_nt(rows);
  if (!cinp || !nt || kernel_volume <= 0) return 0;
  return 3 * tile::image_elems(kernel_volume, cinp, nt);
}

/* filters [K, cin, cout] fp32 (the reference layout [kx,ky,kz,cin,cout], conv.py:100) -> the three bf16 images (hi, mid, lo by
 * truncation) in MFMA-fragment order.  transpose_io != 0: the image of the input-gradient pass (W^T). */
int bevamd_spconv_make_filter_image3(const float* filters, int kernel_volume, int cin, int cout, int transpose_io, void* image,
                                     void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(kernel_volume > 0 && cin > 0 && cout > 0, "spconv_make_filter_image3: bad sizes");
  BEVAMD_REQUIRE(filters && image, "spconv_make_filter_image3: null buffer");
  const int rows = transpose_io ? cin : cout, cols = transpose_io ? cout : cin;
  const int cinp = tile3::pad_cin3(cols), nt = tile::pad_nt(rows);
  BEVAMD_REQUIRE(cinp && nt, "spconv_make_filter_image3: %d -> %d channels (reduction width 16 | 32 | 64 | 128, <= 128 outputs)", cols, rows);
  const int nchunks = tile::image_chunks(kernel_volume, cinp);
  const size_t total = (size_t)nchunks * nt * 64 * 8;
  dim3 grid((unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048)), block(256);
#define BEVAMD_IMG3(C) case C: tile3::spconv_filter_image3_kernel<C><<<grid, block, 0, stream>>>(filters, kernel_volume, cin, cout, nt, nchunks, transpose_io, (uint16_t*)image); break
  switch (cinp) {
    BEVAMD_IMG3(16);
    BEVAMD_IMG3(32);
    BEVAMD_IMG3(64);
    default: BEVAMD_IMG3(128);
  }
#undef BEVAMD_IMG3
  BEVAMD_LAUNCH_CHECK("spconv_filter_image3");
  return BEVAMD_OK;
}

/* Replaces sparse_conv_ext.indice_conv_fp32 / the input-gradient half of indice_conv_backward_fp32 (spconv/src/all.cc:28-31 ->
 * spconv_ops.h:260-456) for fp32 rows of 16 | 32 | 64 | 128 channels on the bf16 matrix cores by three-way operand splitting
 * (six MFMAs per product, fp32 accumulate; error of the order of one fp32 rounding).  Same table and epilogue contract as
 * bevamd_spconv_conv_forward; `image` from bevamd_spconv_make_filter_image3. */
int bevamd_spconv_conv_forward_f32x3(const float* features, int feat_stride, int num_in, const void* image, const int* nbr,
                                     int nbr_stride, int num_out, const int* num_out_dev, int kernel_volume, int cin, int cout,
                                     float* out, int out_stride, const float* bias, const float* bn_scale, const float* bn_shift,
                                     const float* residual, int residual_stride, int relu, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(kernel_volume > 0 && cin > 0 && cout > 0 && num_out >= 0 && num_in >= 0, "spconv_conv_forward_f32x3: bad sizes");
  if (num_out == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(features && image && nbr && out, "spconv_conv_forward_f32x3: null buffer");
  BEVAMD_REQUIRE(nbr_stride >= num_out, "spconv_conv_forward_f32x3: nbr_stride %d < num_out %d", nbr_stride, num_out);
  const int cinp = tile3::pad_cin3(cin), nt = tile::pad_nt(cout);
  BEVAMD_REQUIRE(cinp && nt, "spconv_conv_forward_f32x3: %d -> %d channels (cin 16 | 32 | 64 | 128, cout <= 128)", cin, cout);
  BEVAMD_REQUIRE(feat_stride >= cin && feat_stride % 4 == 0 && ((uintptr_t)features & 15) == 0,
                 "spconv_conv_forward_f32x3: feature pitch %d must be a multiple of 4 and >= %d, 16-byte aligned", feat_stride, cin);
  BEVAMD_REQUIRE((unsigned long long)num_in * feat_stride * 4ull < 0x80000000ull, "spconv_conv_forward_f32x3: feature matrix must be < 2 GiB");
  BEVAMD_REQUIRE(out_stride >= cout && (!residual || residual_stride >= cout), "spconv_conv_forward_f32x3: bad output pitch");
  BEVAMD_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "spconv_conv_forward_f32x3: scale and shift go together");
  BEVAMD_REQUIRE(((uintptr_t)out & 15) == 0 && ((uintptr_t)image & 15) == 0, "spconv_conv_forward_f32x3: out / image must be 16-byte aligned");
  tile3::Args3 a;
  a.feat = features; a.wimg = image; a.nbr = nbr; a.m_dev = num_out_dev; a.out = out;
  a.bias = bias; a.scale = bn_scale; a.shift = bn_shift; a.residual = residual;
  a.feat_stride = feat_stride; a.n_in = num_in; a.nbr_stride = nbr_stride; a.m_cap = num_out; a.K = kernel_volume;
  a.cout = cout; a.out_stride = out_stride; a.res_stride = residual_stride; a.relu = relu;
  a.wimg_bytes = (unsigned)(3 * tile::image_elems(kernel_volume, cinp, nt) * 2);
  switch (cinp) {
    case 16: return tile3::run_cin<16>(a, nt, stream);
    case 32: return tile3::run_cin<32>(a, nt, stream);
    case 64: return tile3::run_cin<64>(a, nt, stream);
    default: return tile3::run_cin<128>(a, nt, stream);
  }
}

}  // extern "C"
