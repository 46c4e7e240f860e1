// Filter gradient of a 3x3x3 submanifold convolution over rows in ASCENDING LINEAR INDEX, from the slab metadata of the
// staged-rows forward kernels (spconv_slab_meta.h) — round 6.
//
// Reference: spconv_ops.h:363-456 (indiceConvBackward<T>, the filter half): per kernel offset, gather the input rows and the
// out_grad rows of its pairs and torch::mm their transposed product.  dW[k] = X_k^T gY with X_k[o] = X[nbr[k][o]].
//
// What the gather kernel (spconv_wgrad16_kernel, spconv_conv.hip) pays: every (row, offset) pair fetches its neighbour row from
// L2 into registers, writes it to a wave-private LDS tile and reads it back transposed — 19 of 27 offsets exist per row, so every
// input row crosses the texture path ~19 times in 64-byte pieces, and each 32-row tile costs a wave a dependent
// index -> gather -> ds_write -> ds_read_tr chain per offset (2 % of the MFMA peak).
//
// Here: on a level in linear order the inputs that a block of 128 output rows reads through one kernel plane kx are ONE
// contiguous range of rows (hdr = (lo, cnt) per block and plane; slots = the neighbour table as 16-bit offsets into that range).
// A workgroup copies the three ranges of its block global -> LDS with LDS-DMA (coalesced 1 KiB pieces, no registers), one block
// ahead of the MFMAs, together with the block's out_grad rows; the slot table travels two blocks ahead and is turned into LDS
// byte addresses ("baked") one block ahead.  The reduction index of this GEMM is the ROW:
// ds_read_b64_tr_b16 takes a per-lane address and transposes across its 16-lane group, so lane (c, g) hands in the address of
// the STAGED ROW of neighbour 8g + (c >> 2) — the gather happens in the address of the transposing read.  No per-offset global
// load, no ds_write, no wait in the inner loop: per (32-row chunk, offset) a wave issues 2 * CIT transposing reads and
// CIT * COT MFMAs (v_mfma_f32_16x16x32_{f16,bf16}), the out_grad fragments are read once per chunk for all its offsets.
//
// Wave w owns offsets w, w + 8, w + 16, w + 24 (accumulators: KPW * CIT * COT tiles of 16 x 16 fp32 in registers for the whole
// slab of blocks); a workgroup owns CIT * 16 input channels x COT * 16 output channels; slab partials in fp32, reduced in fixed
// order by spconv_wgrad_reduce_kernel — same workspace contract and bit-reproducibility as the gather kernel.
//
// A range longer than CAP rows (1-2 % of the planes at 128-row blocks) is walked in pieces: slots outside the staged piece read
// the zero row and add exact zeros.
#pragma once
#include <type_traits>

#include "common.h"
#include "spconv_slab_meta.h"

namespace bevamd {
namespace wgslab {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef BEVAMD_WGS_ABL
#define BEVAMD_WGS_ABL 0   // experiment builds (tools/exp_build.sh ... -DBEVAMD_WGS_ABL=mask): parts compiled out, wrong results by design
#endif
constexpr int ABL = BEVAMD_WGS_ABL;   // 1 no prefetch DMA (first block only), 2 no bake after the first, 4 no MFMA, 8 no X fragment reads, 16 no out_grad fragment reads

constexpr int BM = 128;          // output rows per block (the metadata's block size, raw 16-bit slots)
constexpr int NW = 8;            // waves per workgroup
constexpr int KPW = 4;           // kernel offsets per wave (8 x 4 >= 27)
constexpr int NQ = BM / 32;      // 32-row chunks (one MFMA reduction each) per block
constexpr int SLOT_BYTES = 27 * BM * 2;                    // one block's slot table
constexpr int SLOT_PIECES = (SLOT_BYTES + 1023) / 1024;    // ... in 1 KiB DMA pieces

template <bool F16>
__device__ __forceinline__ f32x4 mfma(const s16x8& a, const s16x8& b, f32x4 acc) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// LDS plan for CIT x 16 staged input channels, COT x 16 out_grad channels, CAP staged rows per plane
template <int CIT, int COT, int CAP>
struct Plan {
  static constexpr int RBX = CIT * 32;                 // bytes of a staged input row
  static constexpr int PPX = RBX / 16;                 // 16-byte pieces per row
  static constexpr int RPX = 64 / PPX;                 // rows per DMA instruction
  static constexpr int RBG = COT * 32;                 // bytes of an out_grad row
  static constexpr int PPG = RBG / 16;
  static constexpr int RPG = 64 / PPG;
  static constexpr int XPLANE = CAP * RBX;             // one plane's buffer
  static constexpr int ZERO_OFF = 3 * XPLANE;          // the zero row of a stage, behind its three planes
  static constexpr int XS = 3 * XPLANE + 1024;         // one stage of X
  static constexpr int GS = BM * RBG;                  // one stage of out_grad
  static constexpr int SS = SLOT_PIECES * 1024;        // one stage of raw slots
  static constexpr int OFF_X = 0;
  static constexpr int OFF_G = 2 * XS;
  static constexpr int OFF_S = OFF_G + 2 * GS;
  static constexpr int OFF_B = OFF_S + 3 * SS;         // baked addresses [27][4 g][4 cc][NQ][2] u16, two tables
  static constexpr int BS = (SLOT_BYTES + 15) / 16 * 16;
  static constexpr int BYTES = OFF_B + 2 * BS;
  static_assert(CAP % RPX == 0 && BM % RPG == 0, "whole DMA instructions");
  static_assert(ZERO_OFF + RBX <= 0xFFFF, "baked addresses are 16-bit");
  static_assert(XPLANE % 1024 == 0 && GS % 1024 == 0, "KiB-aligned stages");
  static_assert(BYTES <= 160 * 1024, "LDS");
};

struct Args {
  const void* feat;       // [n_in, feat_stride] 16-bit
  const void* gout;       // [m, gout_stride] 16-bit
  const int2* hdr;        // [nblk][3] (lo, cnt)
  const uint16_t* slots;  // [nblk][27][BM]
  float* part;            // [nslabs][27][cinp_tot][coutp_tot]
  int feat_stride, gout_stride, n_in, m;
  int nblk, blocks_per_slab, nslabs, ncb, nco;
  int cinp_tot, coutp_tot;
  unsigned slot_bytes;
};

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, (int)voff, (int)soff, 0, 0);
}

template <bool F16, int CIT, int COT, int CAP>
__global__ __launch_bounds__(NW * 64, 1) void spconv_wgrad_slab_kernel(Args a) {
  typedef Plan<CIT, COT, CAP> P;
  extern __shared__ u32x4 lds_[];
  char* const L = (char*)lds_;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g = lane >> 4, cc = c >> 2;
  // workgroup -> (slab, channel block): the channel blocks of one slab share an XCD (ids congruent mod 8), a slab is a
  // contiguous run of blocks, so an XCD's L2 sees each staged row from all the workgroups that want it
  const int id = blockIdx.x;
  const int grp = id >> 3;
  const int slab = (grp / a.ncb) * 8 + (id & 7);
  const int cb = grp % a.ncb;
  if (slab >= a.nslabs) return;
  const int ci0 = (cb / a.nco) * CIT * 16, co0 = (cb % a.nco) * COT * 16;
  const int blk_beg = slab * a.blocks_per_slab;
  const int blk_end = blk_beg + a.blocks_per_slab < a.nblk ? blk_beg + a.blocks_per_slab : a.nblk;

  f32x4 acc[KPW][CIT][COT];
#pragma unroll
  for (int kk = 0; kk < KPW; ++kk)
#pragma unroll
    for (int i = 0; i < CIT; ++i)
#pragma unroll
      for (int j = 0; j < COT; ++j) acc[kk][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const unsigned xrow = (unsigned)a.feat_stride * 2u, grow = (unsigned)a.gout_stride * 2u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.feat, 0, (unsigned)a.n_in * xrow, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc((void*)a.gout, 0, (unsigned)a.m * grow, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)a.slots, 0, a.slot_bytes, 0x00020000);

  // zero rows of both stages (never overwritten by a DMA)
  if (tid < 2 * P::PPX) *(u32x4*)(L + P::OFF_X + (tid / P::PPX) * P::XS + P::ZERO_OFF + (tid % P::PPX) * 16) = u32x4{0u, 0u, 0u, 0u};

  // block header: (lo, cnt) of the three planes.  Requested TWO blocks ahead as plain loads (raw) and made wave-uniform one block
  // later (use): asked for at the top of the block that needs it, the header was a full memory round trip in front of every
  // block's DMA requests.
  struct Hdr { int2 h[3]; };
  auto load_hdr_raw = [&](int blk) {
    Hdr r;
#pragma unroll
    for (int j = 0; j < 3; ++j) r.h[j] = a.hdr[(size_t)blk * 3 + j];
    return r;
  };
  auto use_hdr = [&](const Hdr& r, int (&lo)[3], int (&cnt)[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      lo[j] = __builtin_amdgcn_readfirstlane(r.h[j].x);
      cnt[j] = __builtin_amdgcn_readfirstlane((int)((unsigned)r.h[j].y & ~slab::HDR_RAW));
    }
  };
  // rows [pbase, pbase + CAP) of the planes' ranges -> X stage st (piece `pbase / CAP` of the block)
  // Bank swizzles (ds_read_b64_tr_b16 serves 32 lanes = the rows {8 g + cc : g in 0..1, cc in 0..3} x 32 bytes per LDS cycle, bank =
  // (address / 4) % 64).  64-byte staged rows: rows r and r + 8 share their banks -> the two 32-byte halves of a row are stored
  // swapped when bit 3 of its slot is set.  The LDS image stays lane-linear (LDS-DMA); the DMA applies the swap on its SOURCE address.
  const unsigned xlr = (unsigned)(lane / P::PPX), xsp0 = (unsigned)(lane % P::PPX);
  const unsigned xsp = CIT == 2 ? xsp0 ^ (((xlr >> 3) & 1u) << 1) : xsp0;
  static_assert(CIT == 1 || P::RPX == 16, "the swap of a row depends on the lane only");
  auto stage_x = [&](const int (&lo)[3], const int (&cnt)[3], int pbase, int st) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int left = cnt[j] - pbase;
      if (left <= 0) continue;
      const unsigned rows = (unsigned)(left < CAP ? left : CAP);
      const int n = (int)((rows + P::RPX - 1) / P::RPX);
      const unsigned soff = (unsigned)(lo[j] + pbase) * xrow + (unsigned)ci0 * 2u;
      char* dst = L + P::OFF_X + st * P::XS + j * P::XPLANE;
      for (int i = w; i < n; i += NW) {
        unsigned r = (unsigned)i * P::RPX + xlr;
        r = r < rows ? r : rows - 1u;
        dma16(rs_x, r * xrow + xsp * 16u, soff, dst + i * 1024);
      }
    }
  };
  // out_grad rows: 128-byte rows (COT = 4) put rows of equal parity on the same banks and a 16-channel tile covers 8 of them -> tile t
  // of row r is stored at tile position t ^ f(r), f = bit 1 of r | bit 3 of r << 1 (the 8 rows of a read then cover all 64 banks);
  // 64-byte rows (COT = 2): halves swapped by bit 3 of the row, as for the input rows
  const unsigned glr = (unsigned)(lane / P::PPG), gsp0 = (unsigned)(lane % P::PPG);
  const unsigned gsp = COT == 4 ? gsp0 ^ (((glr >> 1) & 1u) << 1) : COT == 2 ? gsp0 ^ (((glr >> 3) & 1u) << 1) : gsp0;
  const unsigned g_tile_swz = COT == 4 ? ((unsigned)(cc >> 1) & 1u) | (((unsigned)g & 1u) << 1) : COT == 2 ? ((unsigned)g & 1u) : 0u;
  auto stage_g = [&](int blk, int st) {
    const unsigned r0 = (unsigned)blk * BM;
    const unsigned last = (unsigned)a.m - 1u;
    char* dg = L + P::OFF_G + st * P::GS;
    for (int i = w; i < BM / P::RPG; i += NW) {
      unsigned r = r0 + (unsigned)i * P::RPG + glr;
      r = r < last ? r : last;     // rows past the end re-read the last row: their slots are NO_SLOT, so they multiply zeros
      const unsigned piece = COT == 4 ? gsp ^ (((unsigned)i & 1u) << 2) : gsp;   // COT = 4: 8 rows per instruction, bit 3 of the row = i & 1
      dma16(rs_g, r * grow + (unsigned)co0 * 2u + piece * 16u, 0u, dg + i * 1024);
    }
  };
  auto stage_slots = [&](int blk, int sb) {      // sb: one of the three slot stages
    char* ds = L + P::OFF_S + sb * P::SS;
    const unsigned sbase = (unsigned)blk * SLOT_BYTES;
    for (int i = w; i < SLOT_PIECES; i += NW) {
      unsigned off = sbase + (unsigned)i * 1024u + (unsigned)lane * 16u;
      off = off + 16u <= a.slot_bytes ? off : a.slot_bytes - 16u;
      dma16(rs_s, off, 0u, ds + i * 1024);
    }
  };
  // raw slots of stage st -> baked LDS offsets of piece pbase: entry (k, r) = byte offset of the staged row inside an X stage (or the
  // zero row), stored where lane (g, cc) finds its 8 rows {32 q + 8 g + cc + 4 h} of offset k in ONE 16-byte read
  auto bake = [&](const int (&cnt)[3], int pbase, int sb, int bt) {     // slot stage sb -> baked table bt
    const uint16_t* raw = (const uint16_t*)(L + P::OFF_S + sb * P::SS);
    uint16_t* const baked = (uint16_t*)(L + P::OFF_B + bt * P::BS);
#pragma unroll
    for (int j = 0; j < 3; ++j) {            // plane by plane: its row count stays in a scalar register
      const int left = cnt[j] - pbase;
      const unsigned lim = (unsigned)(left < 0 ? 0 : left < CAP ? left : CAP);
      for (int e = tid; e < 9 * BM; e += NW * 64) {
        const int k = j * 9 + e / BM, r = e % BM;
        const unsigned s = (unsigned)raw[j * 9 * BM + e] - (unsigned)pbase;         // NO_SLOT stays above every limit
        const unsigned off = s < lim ? (unsigned)(j * P::XPLANE) + s * P::RBX + (CIT == 2 ? ((s >> 3) & 1u) << 5 : 0u) : (unsigned)P::ZERO_OFF;
        const int q = r >> 5, gg = (r >> 3) & 3, h = (r >> 2) & 1, c2 = r & 3;
        baked[((k * 4 + gg) * 4 + c2) * (NQ * 2) + q * 2 + h] = (uint16_t)off;
      }
    }
  };
  auto tr = [&](const char* p) -> s16x4 {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  };
  // the MFMAs of one staged piece of a block
  // The (chunk, offset) units of a block form one flat list; the fragments of unit u + 1 (2 * CIT gathered transposing reads, and
  // the 2 * COT out_grad reads when it opens a chunk) are requested BEFORE the CIT * COT MFMAs of unit u are issued — two register
  // sets, sched_barrier keeps hipcc from re-serialising them behind one s_waitcnt (left to itself it waits for every read of an
  // offset before its first MFMA and requests the next offset's after the last: a wave's LDS round trip per offset was exposed).
  // Every wave runs its first three offsets in one unrolled body; waves 0-2 then run their fourth (offsets 24-26) in a second, short
  // one (two whole bodies for 3 and 4 offsets behind one branch made hipcc rename every accumulator: 206 registers instead of 122).
  auto compute_n = [&](int st, auto k0_, auto nt_) __attribute__((always_inline)) {
    constexpr int KK0 = decltype(k0_)::value, NT = decltype(nt_)::value, U = NQ * NT;
    const uint16_t* const baked = (const uint16_t*)(L + P::OFF_B + st * P::BS);
    const char* X = L + P::OFF_X + st * P::XS + (c & 3) * 8;
    const char* G = L + P::OFF_G + st * P::GS + (8 * g + cc) * P::RBG + (c & 3) * 8;
    unsigned gto[COT];   // byte offset of out_grad tile t inside this lane's rows
#pragma unroll
    for (int t = 0; t < COT; ++t) gto[t] = ((unsigned)t ^ g_tile_swz) << 5;
    u32x4 ad[NT];
#pragma unroll
    for (int kk = 0; kk < NT; ++kk) ad[kk] = *(const u32x4*)(baked + (((w + (KK0 + kk) * NW) * 4 + g) * 4 + cc) * (NQ * 2));
    auto load_b = [&](int q, s16x8 (&b)[COT]) {
#pragma unroll
      for (int t = 0; t < COT; ++t) {
        if constexpr (ABL & 16) { b[t] = s16x8{(short)(q + t), 1, 2, 3, 4, 5, 6, (short)lane}; continue; }
        const s16x4 lo = tr(G + q * 32 * P::RBG + gto[t]);
        const s16x4 hi = tr(G + (q * 32 + 4) * P::RBG + gto[t]);
        b[t] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    };
    auto load_x = [&](int u, s16x8 (&x)[CIT]) {
      const unsigned pair = ad[u % NT][u / NT];
      const unsigned o_lo = pair & 0xFFFFu, o_hi = pair >> 16;
#pragma unroll
      for (int t = 0; t < CIT; ++t) {
        if constexpr (ABL & 8) { x[t] = s16x8{(short)o_lo, (short)o_hi, 2, 3, 4, 5, (short)t, (short)lane}; continue; }
        const s16x4 lo = tr(X + (o_lo ^ (unsigned)(t * 32)));   // xor: the baked offset carries the half swap in bit 5
        const s16x4 hi = tr(X + (o_hi ^ (unsigned)(t * 32)));
        x[t] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    };
    auto mma = [&](int kk, const s16x8 (&x)[CIT], const s16x8 (&b)[COT]) {
#pragma unroll
      for (int ta = 0; ta < CIT; ++ta)
#pragma unroll
        for (int tb = 0; tb < COT; ++tb) {
          if constexpr (ABL & 4) acc[kk][ta][tb][0] += (float)(x[ta][0] ^ b[tb][7]);
          else acc[kk][ta][tb] = mfma<F16>(x[ta], b[tb], acc[kk][ta][tb]);
        }
    };
    s16x8 b0[COT], b1[COT], x0[CIT], x1[CIT];
    load_b(0, b0);
    load_x(0, x0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = u / NT, kk = u % NT;
      if (u + 1 < U) {
        const int q1 = (u + 1) / NT;
        if ((u + 1) % NT == 0) { if (q1 & 1) load_b(q1, b1); else load_b(q1, b0); }
        if ((u + 1) & 1) load_x(u + 1, x1); else load_x(u + 1, x0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (u & 1) { if (q & 1) mma(KK0 + kk, x1, b1); else mma(KK0 + kk, x1, b0); }
      else { if (q & 1) mma(KK0 + kk, x0, b1); else mma(KK0 + kk, x0, b0); }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto compute = [&](int st) __attribute__((always_inline)) {
    compute_n(st, std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
    if (w + 3 * NW < 27) compute_n(st, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
  };

  if (blk_beg < blk_end) {
    // Pipeline: while block n multiplies, the rows and out_grad of block n + 1 and the slot table of block n + 2 are in flight,
    // and the slot table of block n + 1 (landed one block ago) is baked into the other address table — one barrier per block.
    int lo[3], cnt[3], lo_n[3], cnt_n[3];
    Hdr raw = load_hdr_raw(blk_beg);
    use_hdr(raw, lo, cnt);
    if (blk_beg + 1 < blk_end) { raw = load_hdr_raw(blk_beg + 1); use_hdr(raw, lo_n, cnt_n); }
    stage_x(lo, cnt, 0, 0);
    stage_g(blk_beg, 0);
    stage_slots(blk_beg, 0);
    if (blk_beg + 1 < blk_end) stage_slots(blk_beg + 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bake(cnt, 0, 0, 0);
    __syncthreads();
    int st = 0, sb = 0;                             // X / out_grad / baked stage, slot stage of the current block
    for (int blk = blk_beg; blk < blk_end; ++blk, st ^= 1, sb = sb == 2 ? 0 : sb + 1) {
      const bool more = blk + 1 < blk_end;
      const int sb1 = sb == 2 ? 0 : sb + 1, sb2 = sb1 == 2 ? 0 : sb1 + 1;
      if (blk + 2 < blk_end) raw = load_hdr_raw(blk + 2);
      if (more && !(ABL & 1)) {
        stage_x(lo_n, cnt_n, 0, st ^ 1);
        stage_g(blk + 1, st ^ 1);
        if (blk + 2 < blk_end) stage_slots(blk + 2, sb2);
        if (!(ABL & 2)) bake(cnt_n, 0, sb1, st ^ 1);
      }
      compute(ABL & 1 ? 0 : st);
      int longest = cnt[0] > cnt[1] ? cnt[0] : cnt[1];
      longest = longest > cnt[2] ? longest : cnt[2];
      for (int pbase = CAP; pbase < longest; pbase += CAP) {   // rare: a range longer than the stage — its next CAP rows, synchronously
        __syncthreads();
        stage_x(lo, cnt, pbase, st);
        bake(cnt, pbase, sb, st);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        compute(st);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                              // the next block has landed and is baked; everybody is done with this one
#pragma unroll
      for (int j = 0; j < 3; ++j) { lo[j] = lo_n[j]; cnt[j] = cnt_n[j]; }
      if (blk + 2 < blk_end) use_hdr(raw, lo_n, cnt_n);
    }
  }
  // D[i = ci][j = co]: lane holds column co = c, rows ci = 4 g + e of a tile
#pragma unroll
  for (int kk = 0; kk < KPW; ++kk) {
    const int k = w + kk * NW;
    if (k >= 27) continue;
    float* dst = a.part + ((size_t)slab * 27 + k) * a.cinp_tot * a.coutp_tot;
#pragma unroll
    for (int ta = 0; ta < CIT; ++ta)
#pragma unroll
      for (int tb = 0; tb < COT; ++tb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          dst[(size_t)(ci0 + ta * 16 + g * 4 + e) * a.coutp_tot + co0 + tb * 16 + c] = acc[kk][ta][tb][e];
  }
}

// Slab partials -> filter gradient: out[i] = sum_s part[s][i] in a FIXED order.  The gather kernel's reduce walks the slabs serially in
// one thread per element (256 dependent-latency loads for the 16-channel layers: 62 us for 7 MB); here 16 lanes share an element
// quad: lane group gq sums slabs gq, gq + 16, ... (ascending), the 16 partial sums are combined in ascending group order through LDS.
template <bool F16>
__global__ __launch_bounds__(256) void wgrad_slab_reduce_kernel(const float* __restrict__ part, int nslabs, int n, uint16_t* __restrict__ gw) {
  __shared__ float4 sh[16][16];
  const int q = threadIdx.x & 15, gq = threadIdx.x >> 4;
  const int i = (blockIdx.x * 16 + q) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n)
    for (int s = gq; s < nslabs; s += 16) {
      const float4 t = *(const float4*)(part + (size_t)s * n + i);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
  sh[gq][q] = v;
  __syncthreads();
  if (gq == 0 && i < n) {
    float4 r = sh[0][q];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float4 t = sh[k][q]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
    const float f[4] = {r.x, r.y, r.z, r.w};
    uint16_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (F16) { const _Float16 h = (_Float16)f[e]; o[e] = __builtin_bit_cast(uint16_t, h); }
      else {
        uint32_t u = __float_as_uint(f[e]);
        if ((u & 0x7FFFFFFFu) > 0x7F800000u) o[e] = (uint16_t)((u >> 16) | 0x40);
        else { u += 0x7FFFu + ((u >> 16) & 1u); o[e] = (uint16_t)(u >> 16); }
      }
    }
    *(uint2*)(gw + i) = make_uint2((uint32_t)o[0] | ((uint32_t)o[1] << 16), (uint32_t)o[2] | ((uint32_t)o[3] << 16));
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
struct Shape { int cit, cot, nci, nco, cinp, coutp; };
// cin == cout in {16, 32, 64, 128}: a workgroup owns at most 32 input x 64 output channels (128 accumulator registers per lane)
static inline bool shape_for(int cin, int cout, Shape& s) {
  if (cin != cout || (cin != 16 && cin != 32 && cin != 64 && cin != 128)) return false;
  s.cit = cin >= 32 ? 2 : 1;
  s.cot = cout >= 64 ? 4 : cout / 16;
  s.nci = cin / (s.cit * 16);
  s.nco = cout / (s.cot * 16);
  s.cinp = cin;
  s.coutp = cout;
  return true;
}
constexpr int MAX_SLABS = 256;
static inline int slabs_for(int nblk, const Shape& s) {
  // one workgroup per compute unit and launch: 256 / (channel blocks), a multiple of 8 (XCD map), at least 2 blocks per slab when
  // there are that many
  int n = 256 / (s.nci * s.nco);
  n = n / 8 * 8;
  if (n < 8) n = 8;
  if (n > MAX_SLABS) n = MAX_SLABS;
  while (n > 8 && nblk < 2 * n) n -= 8;
  return n;
}

template <bool F16, int CIT, int COT, int CAP>
static int launch_one(const Args& a, hipStream_t stream) {
  typedef Plan<CIT, COT, CAP> P;
  auto kern = spconv_wgrad_slab_kernel<F16, CIT, COT, CAP>;
  static int raised[64] = {0};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !raised[dev]) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
    raised[dev] = 1;
  }
  const int groups = (a.nslabs + 7) / 8 * a.ncb;
  kern<<<dim3(groups * 8), dim3(NW * 64), P::BYTES, stream>>>(a);
  BEVAMD_LAUNCH_CHECK("spconv_wgrad_slab");
  return BEVAMD_OK;
}

template <bool F16>
static int launch(const Args& a, const Shape& s, hipStream_t stream) {
  if (s.cit == 1 && s.cot == 1) return launch_one<F16, 1, 1, 256>(a, stream);
  if (s.cit == 2 && s.cot == 2) return launch_one<F16, 2, 2, 192>(a, stream);
  if (s.cit == 2 && s.cot == 4) return launch_one<F16, 2, 4, 192>(a, stream);
  set_error("spconv_conv_wgrad_slab: no kernel for %d x %d tiles", s.cit, s.cot);
  return BEVAMD_ERR_UNSUPPORTED;
}

}  // namespace wgslab
}  // namespace bevamd
