// Tiled sparse-convolution forward for gfx950, 16-bit features (fp16 / bf16), fp32 accumulate.
//
// Same operator as spconv_conv.hip (reference: spconv_ops.h:260-361 indiceConv<T>, one launch per
// convolution, output-stationary rulebook nbr[k][o]) with the inner loop rebuilt around what bounds it
// on MI355X — the row gather and its latency, not the MFMA rate:
//
//   * the reduction dimension is the FLATTENED (offset, input channel) axis, cut into 32-wide chunks;
//     with Cin = 16 a chunk spans two kernel offsets, with Cin = 8 four, so small-channel layers do not
//     multiply zeros (a lane group g of an MFMA operand carries 8 consecutive channels of ONE offset);
//   * the filter is pre-arranged into MFMA-fragment order ("filter image", 1 KiB per (chunk, 16 output
//     channels)): staging it into LDS is a linear copy and every B-fragment read is one conflict-free
//     ds_read_b128 at lane*16;
//   * RESIDENT kernel (filter image <= 64 KiB): the whole image lives in LDS, a workgroup is persistent
//     over row tiles, no barrier in the main loop;  STREAM kernel: the image of one step is double-
//     buffered in LDS (global -> registers issued before the MFMAs, written after them, one barrier);
//   * rows are gathered with raw buffer loads: a missing neighbour (-1) becomes an out-of-range offset
//     that returns zeros, so the loop is branch-free and the loads of step s+1 and the indices of step
//     s+2 are in flight while step s multiplies;
//   * operands are swapped (D^T = W^T * X^T): a lane ends up with 4 consecutive output channels of one
//     row -> 8-byte stores, 16-byte epilogue-vector loads;
//   * blockIdx is remapped so that each XCD (own L2) walks a contiguous range of row tiles;
//   * the row count may live on the device (`m_dev`): launches are sized by capacity and never need a
//     host sync.
// Summation order is fixed (step-major, then chunk) -> bit-reproducible.
#pragma once
#include "common.h"

namespace bevamd {
namespace tile {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { T_F16 = 1, T_BF16 = 2 };  // same codes as the C ABI's dtype

template <int DT> struct Num;
template <> struct Num<T_F16> {
  typedef _Float16 T;
  __device__ static float to_f32(_Float16 v) { return (float)v; }
  __device__ static _Float16 from_f32(float v) { return (_Float16)v; }
};
template <> struct Num<T_BF16> {
  typedef uint16_t T;
  __device__ static float to_f32(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
  __device__ static uint16_t from_f32(float v) {  // round to nearest even
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
  }
};

template <int DT>
__device__ __forceinline__ f32x4 mfma(const u32x4& w, const u32x4& x, f32x4 acc) {
  if constexpr (DT == T_F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}

// Rounding point of the epilogue: the fp32 value is made opaque before it is narrowed, so that hipcc cannot fuse the producing
// add / fma with the conversion into v_fma_mixlo_f16 (ONE rounding from the exact result; measured: 1-ulp differences in 1e-4 of
// the elements between the two epilogue flavours below, depending on which pattern the instruction selector happened to see).
// Every kernel thus rounds twice (fp32 operation, then the cast) — the arithmetic of the unfused reference pipeline.
template <int DT>
__device__ __forceinline__ float round_to_storage(float v) {
  asm volatile("" : "+v"(v));
  return Num<DT>::to_f32(Num<DT>::from_f32(v));
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

// The tail of every 16-bit epilogue: 8 consecutive channels of one row, given as the conv result ALREADY rounded to 16 bits
// (`raw`), through bias add, folded BatchNorm, residual add and ReLU — each a stored 16-bit tensor in the unfused reference
// pipeline, so each rounds once.  fp16 runs on packed halves: the adds are v_pk_add_f16 (the reference's own half add: one
// rounding from the exact sum), BatchNorm is an fp32 fma rounded to half (kept opaque so that no v_fma_mix fuses the two
// roundings), ReLU clears the halves that are negative and not a NaN (integer test; NaNs of either sign pass through, as in the
// reference's torch ReLU).  4-5 VALU instructions per channel instead of 14: measured
// 25-50 us of the 170-190 us 32-channel layers went into this tail.  bf16 has no packed arithmetic and keeps fp32 steps.
template <int DT>
__device__ __forceinline__ unsigned finish_pair(unsigned raw, bool has_bias, unsigned bias, bool has_scale, float s0, float s1, float h0, float h1,
                                                bool has_res, unsigned res, bool relu) {
  typedef typename Num<DT>::T T;
  if constexpr (DT == T_F16) {
    f16x2 y = __builtin_bit_cast(f16x2, raw);
    if (has_bias) y = y + __builtin_bit_cast(f16x2, bias);
    if (has_scale) {
      float f0 = __builtin_fmaf((float)y[0], s0, h0), f1 = __builtin_fmaf((float)y[1], s1, h1);
      asm volatile("" : "+v"(f0), "+v"(f1));
      y = f16x2{(_Float16)f0, (_Float16)f1};
    }
    if (has_res) y = y + __builtin_bit_cast(f16x2, res);
    if (relu) {
      // a half is cleared when it is negative and NOT a NaN: as a signed integer that is b <= -1024 (0xFC00 = -inf; the negative
      // NaNs are -1023 .. -1), i.e. the saturating b + 1023 is negative.  torch's ReLU keeps NaNs of either sign, so does this.
      const s16x2 b = __builtin_bit_cast(s16x2, y);
      s16x2 m = __builtin_elementwise_add_sat(b, (s16x2){1023, 1023}) >> 15;
      asm volatile("" : "+v"(m));   // keeps v_pk_add_i16 clamp + v_pk_ashrrev_i16 (hipcc otherwise splits the halves into compares)
      y = __builtin_bit_cast(f16x2, (s16x2)(b & ~m));
    }
    return __builtin_bit_cast(unsigned, y);
  } else {
    const float sv[2] = {s0, s1}, hv[2] = {h0, h1};
    unsigned o = 0;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      float x = Num<DT>::to_f32((T)(raw >> (16 * e)));
      if (has_bias) x = round_to_storage<DT>(x + Num<DT>::to_f32((T)(bias >> (16 * e))));
      if (has_scale) x = round_to_storage<DT>(__builtin_fmaf(x, sv[e], hv[e]));
      if (has_res) x = round_to_storage<DT>(x + Num<DT>::to_f32((T)(res >> (16 * e))));
      const unsigned h = (unsigned)Num<DT>::from_f32(x);
      o |= (relu && (h - 0x8000u) <= 0x7F80u ? 0u : h) << (16 * e);   // negative and not a NaN (0x8000 .. 0xFF80) -> +0
    }
    return o;
  }
}
template <int DT>
__device__ __forceinline__ u32x4 finish8(u32x4 raw, bool has_bias, u32x4 bias, bool has_scale, const float (&sv)[8], const float (&hv)[8],
                                         bool has_res, u32x4 res, bool relu) {
  u32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    o[i] = finish_pair<DT>(raw[i], has_bias, bias[i], has_scale, sv[2 * i], sv[2 * i + 1], hv[2 * i], hv[2 * i + 1], has_res, res[i], relu);
  return o;
}

struct Args {
  const void* feat;     // [n_in, feat_stride] 16-bit
  const void* wimg;     // filter image (see make_filter_image)
  const int* nbr;       // [K, nbr_stride]
  const int* m_dev;     // optional device row count
  void* out;            // [m, out_stride] 16-bit
  const void* bias;     // [cout] 16-bit or null
  const float* scale;   // [cout] fp32 or null  } folded BatchNorm
  const float* shift;   // [cout] fp32 or null  }
  const void* residual; // [m, res_stride] 16-bit or null
  // dense tail (optional): out_dense[b][c*dz + z][x][y] = value, rows addressed through out_indices
  int feat_stride, n_in, nbr_stride, m_cap, K, cout, out_stride, res_stride, relu;
  int row_epilogue;  // 1: cout, pitches and pointers allow the LDS-transposed 16-byte epilogue (set by the host)
  // optional: the rulebook as slab metadata instead of the int32 table (nbr == nullptr): per block of slab_rows output rows and
  // kernel plane kx the first input row (hdr[blk*3 + kx].x) and 16-bit slots relative to it ([blk][27][slab_rows], 0xFFFF = none)
  // — half the bytes of the table, built by sorted-key search without clearing / scattering one (spconv_indice.hip).  K = 27.
  const int2* hdr;
  const uint16_t* slots;
  int slab_rows;
};

constexpr unsigned OOB = 0x80000000u;  // buffer offset that is always out of range (feature bytes < 2 GiB)

// The reduction axis is the flattened (kernel offset, input channel) axis in 32-wide CHUNKS (chunk j covers
// flat elements [32j, 32j+32); lane group g carries 8 of them, all of one offset).  A STEP multiplies CPO
// consecutive chunks: the unit of software pipelining (and of LDS staging in the stream kernel).  More chunks
// per step = more gathers in flight per wave and fewer barriers, at the price of registers.
template <int CINP> struct Chunks {
  static constexpr int CPB = CINP >= 32 ? CINP / 32 : 1;   // chunks per kernel offset (CINP >= 32)
  static constexpr int OPC = CINP >= 32 ? 1 : 32 / CINP;   // kernel offsets per chunk (CINP < 32)
  __host__ __device__ static int count(int K) { return CINP >= 32 ? K * CPB : (K + OPC - 1) / OPC; }
};
constexpr int IMAGE_CHUNK_PAD = 24;  // images are zero-padded to a multiple of this many chunks (every CPO divides it)

// index slots: the distinct kernel offsets a lane touches in one step
template <int CINP, int CPO> struct StepShape {
  static_assert(CINP < 32 || CPO % (CINP / 32) == 0, "a step must hold whole kernel offsets");
  static constexpr int CPB = Chunks<CINP>::CPB;
  static constexpr int NK = CINP >= 32 ? CPO / CPB : CPO;
  __host__ __device__ static int nsteps(int K) { return (Chunks<CINP>::count(K) + CPO - 1) / CPO; }
  // kernel offset of index slot q of step s for lane group g
  __device__ __forceinline__ static int offset_of(int s, int q, int g) {
    if constexpr (CINP >= 32) return s * NK + q;
    else return ((s * CPO + q) * 32 + g * 8) / CINP;
  }
  // index slot and byte offset inside the feature row of chunk cc of a step for lane group g
  __device__ __forceinline__ static int slot_of(int cc) { return CINP >= 32 ? cc / CPB : cc; }
  __device__ __forceinline__ static unsigned byte_of(int cc, int g) {
    if constexpr (CINP >= 32) return (unsigned)((cc % CPB) * 64 + g * 16);
    else return (unsigned)(((cc * 32 + g * 8) % CINP) * 2);   // (s*CPO*32) % CINP == 0 whenever it matters: 32 % CINP == 0
  }
};

// 4 consecutive output channels [col0, col0+4) of one row.  Rounding points follow the unfused reference
// pipeline (the conv result, the bias add, BatchNorm and the residual add are each a stored 16-bit tensor).
template <int DT>
__device__ __forceinline__ void epilogue_store(const Args& a, int row, int col0, f32x4 v) {
  typedef typename Num<DT>::T T;
  if (col0 >= a.cout) return;
  T p[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) p[j] = Num<DT>::from_f32(v[j]);
  T* op = (T*)a.out + (size_t)row * a.out_stride + col0;
  const bool fast = col0 + 4 <= a.cout && ((a.out_stride | a.res_stride) & 3) == 0;  // wave-uniform except at the cout edge
  if (fast) {
    const uint2 raw = *(const uint2*)p;
    uint2 b = make_uint2(0u, 0u), r = make_uint2(0u, 0u);
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) b = *(const uint2*)((const T*)a.bias + col0);
    if (a.scale) { sc = *(const float4*)(a.scale + col0); sh = *(const float4*)(a.shift + col0); }
    if (a.residual) r = *(const uint2*)((const T*)a.residual + (size_t)row * a.res_stride + col0);
    uint2 o;
    o.x = finish_pair<DT>(raw.x, a.bias != nullptr, b.x, a.scale != nullptr, sc.x, sc.y, sh.x, sh.y, a.residual != nullptr, r.x, a.relu != 0);
    o.y = finish_pair<DT>(raw.y, a.bias != nullptr, b.y, a.scale != nullptr, sc.z, sc.w, sh.z, sh.w, a.residual != nullptr, r.y, a.relu != 0);
    *(uint2*)op = o;
    return;
  }
  for (int j = 0; j < 4; ++j) {   // the cout edge, element by element through the same arithmetic (low half of a pair)
    if (col0 + j >= a.cout) break;
    const unsigned raw = (unsigned)*(const uint16_t*)&p[j];
    const unsigned b = a.bias ? (unsigned)((const uint16_t*)a.bias)[col0 + j] : 0u;
    const unsigned r = a.residual ? (unsigned)((const uint16_t*)a.residual)[(size_t)row * a.res_stride + col0 + j] : 0u;
    const float s0 = a.scale ? a.scale[col0 + j] : 1.f, h0 = a.scale ? a.shift[col0 + j] : 0.f;
    const unsigned o = finish_pair<DT>(raw, a.bias != nullptr, b, a.scale != nullptr, s0, 1.f, h0, 0.f, a.residual != nullptr, r, a.relu != 0);
    *(uint16_t*)&op[j] = (uint16_t)o;
  }
}

// wave-private LDS scratch of the row-wise epilogue: 16 rows x (NT*32 + 16) bytes, in u32x4 units
template <int NT> struct EpiScratch { static constexpr int U4 = 16 * (NT * 2 + 1); };

// One wave, one tile of 16*MT output rows.  The reduction runs two steps per loop trip on two register sets
// (A/B): while set A multiplies, the rows of the next step land in set B and the indices of the step after
// that are being fetched — no register copies, no conditional loads, so hipcc keeps counted vmcnt waits and
// the prefetches stay in flight across the back-edge.
// ABL: ablation mask for profiling builds only (tools/sweep_spconv.py --ablate): 1 = no MFMA, 2 = no row gather,
// 4 = no filter staging, 8 = no ds_bpermute, 16 = no LDS fragment reads, 32 = no neighbour preload.  Results are wrong
// for ABL != 0; the shipped kernels are ABL = 0.
template <int DT, int CINP, int NT, int MT, int CPO, int ABL = 0>
struct WaveTile {
  typedef StepShape<CINP, CPO> SS;
  static constexpr int NK = SS::NK;
  f32x4 acc[MT][NT];
  // Two lane identities.  MFMA operand/result layout: lane = (c = lane & 15: row of the tile, g = lane >> 4: k-group).
  // LOAD layout: lane = (lr = lane >> 2: row, lg = lane & 3: k-group), i.e. the four 16-byte pieces of one row sit in
  // four ADJACENT lanes — the texture addresser then sees one contiguous 64-byte segment per lane quad instead of
  // four unrelated rows (a random dwordx4 gather costs ~1 cycle per distinct segment: 64 -> 16 cycles per wave load).
  // ds_bpermute moves the data from load layout to MFMA layout (source lane 4c + g) just before the multiply.
  int row0, m, lane, c, g, lr, lg, perm_addr;
  int* nbl;  // wave-private LDS copy of the tile's neighbour table: [K][16*MT]
  u32x4* eps;  // wave-private LDS scratch of the row-wise epilogue
  unsigned row_bytes;
  __amdgpu_buffer_rsrc_t rs;

  __device__ __forceinline__ void init(const Args& a, int row0_, int m_, int* nbl_, u32x4* eps_) {
    row0 = row0_;
    m = m_;
    nbl = nbl_;
    eps = eps_;
    lane = threadIdx.x & 63;
    c = lane & 15;
    g = lane >> 4;
    lr = lane >> 2;
    lg = lane & 3;
    perm_addr = ((lane & 15) * 4 + (lane >> 4)) * 4;
    row_bytes = (unsigned)a.feat_stride * 2u;
    rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.feat, 0, (unsigned)a.n_in * row_bytes, 0x00020000);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // The tile's whole neighbour table -> LDS in one burst.  The table is streamed exactly once (every access is a cold
  // miss served by HBM / Infinity Cache, ~2-4k cycles), so fetching it step by step put one such latency on every
  // step of every wave — the time of all layers was set by that chain, whatever the tiling.  Here all K*16*MT
  // loads of a tile are independent and in flight together: one latency per tile.
  __device__ __forceinline__ void preload_nb(const Args& a) {
    if constexpr (ABL & 32) {
      for (int i = lane; i < a.K * 16 * MT; i += 64) nbl[i] = (row0 + i) % (m > 0 ? m : 1);
      return;
    }
    constexpr int R = 16 * MT, UNR = 8;
    const int total = a.K * R;
    if (a.nbr == nullptr) {   // slot metadata: a tile never straddles a block (slab_rows % R == 0, checked by the host)
      const int blk = row0 / a.slab_rows, t0 = row0 - blk * a.slab_rows;
      const int lo0 = a.hdr[(size_t)blk * 3 + 0].x, lo1 = a.hdr[(size_t)blk * 3 + 1].x, lo2 = a.hdr[(size_t)blk * 3 + 2].x;
      const uint16_t* sl = a.slots + (size_t)blk * 27 * a.slab_rows + t0;
      for (int base = 0; base < total; base += 64 * UNR) {
        unsigned v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          int idx = base + u * 64 + lane;
          idx = idx < total ? idx : total - 1;
          const int k = idx / R, r = idx - k * R;
          v[u] = row0 + r < m ? (unsigned)sl[(size_t)k * a.slab_rows + r] : 0xFFFFu;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int idx = base + u * 64 + lane;
          if (idx < total) {
            const int k = idx / R;
            nbl[idx] = v[u] == 0xFFFFu ? -1 : (k < 9 ? lo0 : k < 18 ? lo1 : lo2) + (int)v[u];
          }
        }
      }
      return;
    }
    for (int base = 0; base < total; base += 64 * UNR) {
      int v[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        int idx = base + u * 64 + lane;
        idx = idx < total ? idx : total - 1;
        const int k = idx / R, r = idx - k * R;
        const int row = row0 + r;
        v[u] = a.nbr[(size_t)k * a.nbr_stride + (row < m ? row : m - 1)];
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int idx = base + u * 64 + lane;
        if (idx < total) nbl[idx] = v[u];
      }
    }
  }
  // neighbour indices of step s for this lane's load slot (validity is decided in gather())
  __device__ __forceinline__ void load_nb(const Args& a, int s, int (&nb)[MT][NK]) {
#pragma unroll
    for (int q = 0; q < NK; ++q) {
      const int k = SS::offset_of(s, q, lg);
      const int kc = k < a.K ? k : a.K - 1;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) nb[mt][q] = nbl[kc * (16 * MT) + mt * 16 + lr];
    }
  }
  // callers pass a step that exists (clamped); offsets past K inside the last step gather zeros
  __device__ __forceinline__ void gather(const Args& a, int s, const int (&nb)[MT][NK], u32x4 (&dst)[MT][CPO]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const bool live = row0 + mt * 16 + lr < m;
      unsigned base[NK];
#pragma unroll
      for (int q = 0; q < NK; ++q) {
        const bool valid = live && nb[mt][q] >= 0 && SS::offset_of(s, q, lg) < a.K;
        base[q] = valid ? (unsigned)nb[mt][q] * row_bytes : OOB;
      }
#pragma unroll
      for (int cc = 0; cc < CPO; ++cc) {
        if constexpr (ABL & 2) dst[mt][cc] = u32x4{base[SS::slot_of(cc)], 1u, 2u, 3u};
        else dst[mt][cc] = __builtin_amdgcn_raw_buffer_load_b128(rs, base[SS::slot_of(cc)] + SS::byte_of(cc, lg), 0, 0);
      }
    }
  }
  __device__ __forceinline__ u32x4 to_mfma_layout(const u32x4& v) const {
    if constexpr (ABL & 8) return v;
    u32x4 r;
    r.x = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)v.x);
    r.y = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)v.y);
    r.z = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)v.z);
    r.w = (unsigned)__builtin_amdgcn_ds_bpermute(perm_addr, (int)v.w);
    return r;
  }
  // multiply gathered rows (load layout) by the step's filter fragments at `wl` (LDS).
  // LDS latency (~100+ cycles per ds_read_b128 / ds_bpermute) is the hazard here: with one fragment read in
  // flight per MFMA the wave idles on lgkmcnt for most of a step (measured: MFMA busy 19 %, 58 % of wave cycles in
  // s_waitcnt).  So the NT filter fragments and the MT permuted row fragments of chunk cc+1 are all issued
  // BEFORE the MT*NT MFMAs of chunk cc (two register sets), and sched_barrier keeps hipcc from re-serialising.
  __device__ __forceinline__ void fetch_chunk(const u32x4* __restrict__ wl, const u32x4 (&x)[MT][CPO], int cc,
                                              u32x4 (&b)[NT], u32x4 (&xm)[MT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if constexpr (ABL & 16) b[nt] = u32x4{(unsigned)(cc + nt), 0x3C003C00u, 0x3C003C00u, (unsigned)lane};
      else b[nt] = wl[(cc * NT + nt) * 64 + lane];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xm[mt] = to_mfma_layout(x[mt][cc]);
  }
  __device__ __forceinline__ void mma_chunk(const u32x4 (&b)[NT], const u32x4 (&xm)[MT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if constexpr (ABL & 1) {
          acc[mt][nt][0] += __uint_as_float(b[nt].x ^ xm[mt].x);
          acc[mt][nt][1] += __uint_as_float(b[nt].y ^ xm[mt].y);
          acc[mt][nt][2] += __uint_as_float(b[nt].z ^ xm[mt].z);
          acc[mt][nt][3] += __uint_as_float(b[nt].w ^ xm[mt].w);
        } else {
          acc[mt][nt] = mfma<DT>(b[nt], xm[mt], acc[mt][nt]);
        }
      }
  }
  __device__ __forceinline__ void multiply(const u32x4* __restrict__ wl, const u32x4 (&x)[MT][CPO]) {
    u32x4 b0[NT], b1[NT], x0[MT], x1[MT];
    fetch_chunk(wl, x, 0, b0, x0);
#pragma unroll
    for (int cc = 0; cc < CPO; cc += 2) {
      if (cc + 1 < CPO) fetch_chunk(wl, x, cc + 1, b1, x1);
      __builtin_amdgcn_sched_barrier(0);
      mma_chunk(b0, x0);
      __builtin_amdgcn_sched_barrier(0);
      if (cc + 1 < CPO) {
        if (cc + 2 < CPO) fetch_chunk(wl, x, cc + 2, b0, x0);
        __builtin_amdgcn_sched_barrier(0);
        mma_chunk(b1, x1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // Row-wise epilogue.  In the MFMA result layout a lane owns 4 channels of a row: 8-byte stores (and residual loads)
  // 32 bytes per row segment, a quarter of an L2 line each — measured 8 us of a 40 us layer.  The tile is instead
  // rounded to 16 bits (= the reference's stored conv result), transposed through wave-private LDS, and every lane
  // finishes 8 consecutive channels of one row: 16-byte bias / residual loads, 16-byte stores, whole rows contiguous.
  // the residual pieces this lane adds in store_rows (same lane -> (row, 8 channels) map): callers with registers to spare
  // request them long before the epilogue (the slab kernels: at the start of a block's last plane) and pass them to store()
  static constexpr int RES_LPR = NT * 2;
  static constexpr int RES_RPP = 64 / RES_LPR > 16 ? 16 : 64 / RES_LPR;
  static constexpr int RES_PASSES = 16 / RES_RPP;
  struct Residual { u32x4 v[MT][RES_PASSES]; };
  __device__ __forceinline__ void load_residual(const Args& a, Residual& res) const {
    typedef typename Num<DT>::T T;
    const int j = lane % RES_LPR, rsub = lane / RES_LPR, col0 = j * 8;
    const bool col_ok = rsub < RES_RPP && col0 < a.cout;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int pass = 0; pass < RES_PASSES; ++pass) {
        const int row = row0 + mt * 16 + pass * RES_RPP + rsub;
        res.v[mt][pass] = (a.residual && col_ok && row < m) ? *(const u32x4*)((const T*)a.residual + (size_t)row * a.res_stride + col0)
                                                            : u32x4{0u, 0u, 0u, 0u};
      }
  }
  __device__ __forceinline__ void store_rows(const Args& a, const Residual* pre = nullptr) {
    typedef typename Num<DT>::T T;
    constexpr int RB = NT * 32 + 16;          // padded row pitch in the scratch, bytes
    constexpr int LPR = NT * 2;               // lanes (16-byte pieces) per row
    constexpr int RPP = 64 / LPR > 16 ? 16 : 64 / LPR;
    constexpr int PASSES = 16 / RPP;
    char* sc = (char*)eps;
    // A lane finishes the same 8 channels [col0, col0 + 8) of every row it touches: the per-channel operands are loaded once,
    // and the residual pieces of ALL the wave's rows are requested up front — one memory round trip for the epilogue instead of
    // one per 16-row tile (measured on the 32-channel slab layers: the residual add cost 40 us of 240).
    const int j = lane % LPR, rsub = lane / LPR, col0 = j * 8;
    const bool col_ok = rsub < RPP && col0 < a.cout;
    Residual res;
    if (pre) res = *pre;
    else if (a.residual) load_residual(a, res);
    float sv[8], hv[8];
    u32x4 bias16 = u32x4{0u, 0u, 0u, 0u};
    if (a.bias && col_ok) bias16 = *(const u32x4*)((const T*)a.bias + col0);
    if (a.scale && col_ok) {
      const float4 s0 = *(const float4*)(a.scale + col0), s1 = *(const float4*)(a.scale + col0 + 4);
      const float4 h0 = *(const float4*)(a.shift + col0), h1 = *(const float4*)(a.shift + col0 + 4);
      sv[0] = s0.x; sv[1] = s0.y; sv[2] = s0.z; sv[3] = s0.w; sv[4] = s1.x; sv[5] = s1.y; sv[6] = s1.z; sv[7] = s1.w;
      hv[0] = h0.x; hv[1] = h0.y; hv[2] = h0.z; hv[3] = h0.w; hv[4] = h1.x; hv[5] = h1.y; hv[6] = h1.z; hv[7] = h1.w;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        T p[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = Num<DT>::from_f32(acc[mt][nt][q]);
        *(uint2*)(sc + c * RB + nt * 32 + g * 8) = *(const uint2*)p;
      }
#pragma unroll
      for (int pass = 0; pass < PASSES; ++pass) {
        const int r = pass * RPP + rsub;
        const int row = row0 + mt * 16 + r;
        if (col_ok && row < m) {
          const u32x4 raw = *(const u32x4*)(sc + r * RB + j * 16);
          const u32x4 o = finish8<DT>(raw, a.bias != nullptr, bias16, a.scale != nullptr, sv, hv, a.residual != nullptr, res.v[mt][pass], a.relu != 0);
          *(u32x4*)((T*)a.out + (size_t)row * a.out_stride + col0) = o;
        }
      }
    }
  }
  __device__ __forceinline__ void store(const Args& a, const Residual* pre = nullptr) {
    if constexpr (ABL & 64) {  // profiling: keep the accumulators alive with one store per wave
      float t = 0.f;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) t += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3];
      if (t == 12345.678f) ((float*)a.out)[0] = t;
      return;
    }
    // NT == 1: a row is 32 bytes and the tile's 16 rows are already one contiguous 512-byte store per instruction
    if (NT >= 2 && a.row_epilogue) {  // wave-uniform
      store_rows(a, pre);
      return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = row0 + mt * 16 + c;
      if (row >= m) continue;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) epilogue_store<DT>(a, row, nt * 16 + g * 4, acc[mt][nt]);
    }
  }
};

// ---- RESIDENT: whole filter image in LDS, persistent over row tiles -----------------------------------
template <int DT, int CINP, int NT, int MT, int NW, int CPO>
__global__ __launch_bounds__(NW * 64) void spconv_resident_kernel(Args a) {
  extern __shared__ u32x4 lds[];
  typedef WaveTile<DT, CINP, NT, MT, CPO> WT;
  constexpr int NK = WT::NK;
  constexpr int STEP = CPO * NT * 64;
  const int nsteps = StepShape<CINP, CPO>::nsteps(a.K);
  const int total = nsteps * STEP;  // <= the zero-padded image
  for (int i = threadIdx.x; i < total; i += NW * 64) lds[i] = ((const u32x4*)a.wimg)[i];
  __syncthreads();
  const int m = a.m_dev ? (*a.m_dev < a.m_cap ? *a.m_dev : a.m_cap) : a.m_cap;
  constexpr int ROWS = 16 * MT;
  const int ntiles = (m + ROWS - 1) / ROWS;
  const int nxb = gridDim.x >> 3;  // workgroups per XCD (grid is a multiple of 8; block b runs on XCD b % 8)
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per = (ntiles + 7) >> 3;
  const int tend = (xcd + 1) * per < ntiles ? (xcd + 1) * per : ntiles;
  const int w = threadIdx.x >> 6;
  int* nbl = (int*)(lds + total) + w * a.K * ROWS;
  u32x4* eps = lds + total + (NW * a.K * ROWS + 3) / 4 + w * EpiScratch<NT>::U4;
  for (int t = xcd * per + bix * NW + w; t < tend; t += nxb * NW) {
    WT wt;
    wt.init(a, t * ROWS, m, nbl, eps);
    wt.preload_nb(a);
    int nbA[MT][NK], nbB[MT][NK];
    u32x4 xA[MT][CPO], xB[MT][CPO];
    const int last = nsteps - 1;
    auto cl = [&](int st) { return st < last ? st : last; };  // clamped prefetches past the end are never multiplied
    wt.load_nb(a, 0, nbA);
    wt.load_nb(a, cl(1), nbB);
    wt.gather(a, 0, nbA, xA);
    wt.load_nb(a, cl(2), nbA);
    for (int s = 0; s + 1 < nsteps; s += 2) {
      wt.gather(a, s + 1, nbB, xB);
      wt.load_nb(a, cl(s + 3), nbB);
      wt.multiply(lds + (size_t)s * STEP, xA);
      wt.gather(a, cl(s + 2), nbA, xA);
      wt.load_nb(a, cl(s + 4), nbA);
      wt.multiply(lds + (size_t)(s + 1) * STEP, xB);
    }
    if (nsteps & 1) wt.multiply(lds + (size_t)last * STEP, xA);
    wt.store(a);
  }
}

// ---- STREAM: one step's filter fragments double-buffered in LDS ----------------------------------------
// main loop of the stream kernel for one wave tile `wt` (neighbour table already in LDS); `lds` = the 2*STEP filter ring
template <int DT, int CINP, int NT, int MT, int NW, int CPO, int ABL>
__device__ __forceinline__ void stream_loop(const Args& a, WaveTile<DT, CINP, NT, MT, CPO, ABL>& wt, u32x4* lds) {
  constexpr int NK = WaveTile<DT, CINP, NT, MT, CPO, ABL>::NK;
  constexpr int STEP = CPO * NT * 64;                    // u32x4 per step
  constexpr int WPT = (STEP + NW * 64 - 1) / (NW * 64);  // staged u32x4 per thread
  constexpr bool EXACT = STEP % (NW * 64) == 0;
  const int nsteps = StepShape<CINP, CPO>::nsteps(a.K);
  const u32x4* wg = (const u32x4*)a.wimg;
  u32x4 wreg[WPT];
  // filter fragments of step `sc` -> registers / registers -> LDS buffer `buf`
  auto fetch_w = [&](int sc) {
    if constexpr (ABL & 4) return;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = threadIdx.x + i * NW * 64;
      wreg[i] = (EXACT || e < STEP) ? wg[(size_t)sc * STEP + e] : u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto put_w = [&](int buf) {
    if constexpr (ABL & 4) return;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int e = threadIdx.x + i * NW * 64;
      if (EXACT || e < STEP) lds[buf * STEP + e] = wreg[i];
    }
  };
  int nbA[MT][NK], nbB[MT][NK];
  u32x4 xA[MT][CPO], xB[MT][CPO];
  const int last = nsteps - 1;
  auto cl = [&](int st) { return st < last ? st : last; };
  fetch_w(0);
  wt.load_nb(a, 0, nbA);
  wt.load_nb(a, cl(1), nbB);
  put_w(0);
  wt.gather(a, 0, nbA, xA);
  wt.load_nb(a, cl(2), nbA);
  __syncthreads();
  for (int s = 0; s + 1 < nsteps; s += 2) {
    fetch_w(s + 1);
    wt.gather(a, s + 1, nbB, xB);
    wt.load_nb(a, cl(s + 3), nbB);
    wt.multiply(lds, xA);  // step s from buffer 0
    put_w(1);
    if constexpr (!(ABL & 128)) __syncthreads();
    fetch_w(cl(s + 2));
    wt.gather(a, cl(s + 2), nbA, xA);
    wt.load_nb(a, cl(s + 4), nbA);
    wt.multiply(lds + STEP, xB);  // step s+1 from buffer 1
    put_w(0);
    if constexpr (!(ABL & 128)) __syncthreads();
  }
  if (nsteps & 1) wt.multiply(lds, xA);  // odd count: the last step sits in buffer 0
}

template <int DT, int CINP, int NT, int MT, int NW, int CPO, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void spconv_stream_kernel(Args a) {
  extern __shared__ u32x4 lds[];
  typedef WaveTile<DT, CINP, NT, MT, CPO, ABL> WT;
  constexpr int STEP = CPO * NT * 64;
  const int m = a.m_dev ? (*a.m_dev < a.m_cap ? *a.m_dev : a.m_cap) : a.m_cap;
  constexpr int BM = NW * 16 * MT;
  const int nblk = (m + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per = (nblk + 7) >> 3;
  const int tb = xcd * per + bix;
  if (bix >= per || tb >= nblk) return;  // whole workgroup leaves together: no barrier is skipped
  const int w = threadIdx.x >> 6;
  WT wt;
  wt.init(a, tb * BM + w * 16 * MT, m, (int*)(lds + 2 * STEP) + w * a.K * 16 * MT,
          lds + 2 * STEP + (NW * a.K * 16 * MT + 3) / 4 + w * EpiScratch<NT>::U4);
  wt.preload_nb(a);
  stream_loop<DT, CINP, NT, MT, NW, CPO, ABL>(a, wt, lds);
  wt.store(a);
}

// ---- filter image ---------------------------------------------------------------------------------------
// filters [K][cin][cout] (reference layout [kx,ky,kz,cin,cout], conv.py:100) -> image [chunk][nt][lane][8]:
// element e of lane (c = lane&15, g = lane>>4) of chunk j is W[k][ci][nt*16 + c] with flat = 32j + 8g + e,
// k = flat / CINP, ci = flat % CINP; zero outside K / cin / cout and in the padding chunks.
// transpose_io swaps the roles of cin and cout (input gradient).
template <int DT, int CINP>
__global__ __launch_bounds__(256) void spconv_filter_image_kernel(const typename Num<DT>::T* __restrict__ w, int K,
                                                                  int cin, int cout, int nt_count, int nchunks_padded,
                                                                  int transpose_io,
                                                                  typename Num<DT>::T* __restrict__ img) {
  const size_t total = (size_t)nchunks_padded * nt_count * 64 * 8;
  const int rows = transpose_io ? cin : cout;   // output channels of this pass
  const int cols = transpose_io ? cout : cin;   // reduction channels of this pass
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
    typename Num<DT>::T v = Num<DT>::from_f32(0.f);
    if (k < K && ci < cols && co < rows) {
      const int wi = transpose_io ? co : ci, wo = transpose_io ? ci : co;
      v = w[((size_t)k * cin + wi) * cout + wo];
    }
    img[i] = v;
  }
}

// ---- host side --------------------------------------------------------------------------------------------
static inline int pad_cin(int cin) { return cin <= 8 ? 8 : cin <= 16 ? 16 : cin <= 32 ? 32 : cin <= 64 ? 64 : cin <= 128 ? 128 : 0; }
static inline int pad_nt(int cout) {
  int nt = (cout + 15) / 16;
  return nt <= 1 ? 1 : nt <= 2 ? 2 : nt <= 4 ? 4 : nt <= 8 ? 8 : 0;
}
static inline int image_chunks(int K, int cinp) {  // padded chunk count
  const int n = cinp >= 32 ? K * (cinp / 32) : (K + 32 / cinp - 1) / (32 / cinp);
  return (n + IMAGE_CHUNK_PAD - 1) / IMAGE_CHUNK_PAD * IMAGE_CHUNK_PAD;
}
static inline size_t image_elems(int K, int cinp, int nt) { return (size_t)image_chunks(K, cinp) * nt * 64 * 8; }

// implemented once per dtype (spconv_tile_f16.hip / spconv_tile_bf16.hip)
int launch_f16(const Args& a, int cinp, int nt, int variant, hipStream_t stream);
int launch_bf16(const Args& a, int cinp, int nt, int variant, hipStream_t stream);
int image_f16(const void* w, int K, int cin, int cout, int transpose_io, void* img, hipStream_t stream);
int image_bf16(const void* w, int K, int cin, int cout, int transpose_io, void* img, hipStream_t stream);

// variant encoding: 0 = auto; otherwise kind*1000 + MT*100 + (NW/4)*10 + SPS  (kind 1 = resident, 2 = stream;
// SPS = kernel offsets per step for Cin >= 32, chunks per step below)
template <int DT>
int launch_impl(const Args& a, int cinp, int nt, int variant, hipStream_t stream);

}  // namespace tile
}  // namespace bevamd
