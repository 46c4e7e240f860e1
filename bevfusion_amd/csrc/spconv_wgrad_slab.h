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
// ahead of the MFMAs, together with the block's out_grad rows and its neighbour table, which the rulebook producers write as LDS byte
// addresses in the order the lanes read them (spconv_slab_meta.h, FMT_WG64 / FMT_WG32: converting raw slots inside the kernel was
// 36-59 % of the wave cycles — every wave of the one workgroup a CU holds did it at the same time).  The reduction index of this GEMM is the ROW:
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

constexpr int BM = 128;          // output rows per block (the metadata's block size)
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
  static constexpr int OFF_B = OFF_S + 2 * SS;         // addresses converted in the kernel (planes with raw slots, later pieces)
  static constexpr int BS = (SLOT_BYTES + 15) / 16 * 16;
  static constexpr int BYTES = OFF_B + BS;
  static constexpr int FMT = CIT == 2 ? slab::FMT_WG64 : slab::FMT_WG32;
  static_assert(CAP == slab::wg_cap(CIT == 2 ? slab::FMT_WG64 : slab::FMT_WG32) && RBX == slab::wg_row_bytes(CIT == 2 ? slab::FMT_WG64 : slab::FMT_WG32),
                "the metadata format bakes this plan's stage geometry");
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
  unsigned long long* prof;   // -DBEVAMD_WGS_PROF builds: cycle sums over all waves [issue + bake, multiply, dma wait, barrier] + [4] = waves
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
  auto use_hdr = [&](const Hdr& r, int (&lo)[3], int (&cnt)[3], unsigned& rawmask) {   // rawmask bit j: plane j carries raw slots
    rawmask = 0u;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      lo[j] = __builtin_amdgcn_readfirstlane(r.h[j].x);
      const unsigned c2 = (unsigned)__builtin_amdgcn_readfirstlane(r.h[j].y);
      cnt[j] = (int)(c2 & ~slab::HDR_RAW);
      rawmask |= (c2 & slab::HDR_RAW) ? 1u << j : 0u;
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
  auto stage_slots = [&](int blk, int st) {
    char* ds = L + P::OFF_S + st * P::SS;
    const unsigned sbase = (unsigned)blk * SLOT_BYTES;
    for (int i = w; i < SLOT_PIECES; i += NW) {
      unsigned off = sbase + (unsigned)i * 1024u + (unsigned)lane * 16u;
      off = off + 16u <= a.slot_bytes ? off : a.slot_bytes - 16u;
      dma16(rs_s, off, 0u, ds + i * 1024);
    }
  };
  // The table of a block arrives as LDS byte offsets (spconv_slab_meta.h, FMT_WG*), except for planes whose range is longer than
  // the stage: those carry raw slots in the same positions and are converted here, piece by piece, into the one `baked` buffer
  // (planes in `mask`; a plane without rows in this piece gets the zero row everywhere).
  uint16_t* const baked = (uint16_t*)(L + P::OFF_B);
  auto bake = [&](const int (&cnt)[3], unsigned rawmask, unsigned mask, int pbase, int st) {
    const uint16_t* tab = (const uint16_t*)(L + P::OFF_S + st * P::SS);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (!((mask >> j) & 1u)) continue;       // wave-uniform
      const int left = cnt[j] - pbase;
      const unsigned lim = ((rawmask >> j) & 1u) ? (unsigned)(left < 0 ? 0 : left < CAP ? left : CAP) : 0u;
      for (int e = j * 9 * BM + tid; e < (j + 1) * 9 * BM; e += NW * 64) {
        const unsigned s = (unsigned)tab[e] - (unsigned)pbase;         // NO_SLOT stays above every limit
        baked[e] = (uint16_t)(s < lim ? slab::wg_entry(P::FMT, j, s) : (unsigned)P::ZERO_OFF);
      }
    }
  };
  // The requests of the NEXT block as nine per-wave tasks (6 row pieces, 2 out_grad pieces, 1 table piece), issued one after each of
  // the first nine (chunk, offset) units of this block: asked for in one burst at the top of a block, the ~48 KiB of a block queue
  // up in front of the texture path (64 bytes per clock and CU) and the waves sat in the issue of those requests for 23-38 % of their
  // cycles (phase timers of -DBEVAMD_WGS_PROF builds) with nothing to multiply beside them.
  struct Next { bool on; int lo[3], cnt[3], blk, st; } nx{false, {0, 0, 0}, {0, 0, 0}, 0, 0};
  auto dma_task = [&](int t) __attribute__((always_inline)) {
    if (!nx.on) return;                              // wave-uniform
    if (t < 6) {
      const int j = t >> 1, i = w + (t & 1) * NW;
      const int cj = j == 0 ? nx.cnt[0] : j == 1 ? nx.cnt[1] : nx.cnt[2], lj = j == 0 ? nx.lo[0] : j == 1 ? nx.lo[1] : nx.lo[2];
      const unsigned rows = (unsigned)(cj < CAP ? cj : CAP);
      if ((unsigned)i * P::RPX >= rows) return;
      unsigned r = (unsigned)i * P::RPX + xlr;
      r = r < rows ? r : rows - 1u;
      dma16(rs_x, r * xrow + xsp * 16u, (unsigned)lj * xrow + (unsigned)ci0 * 2u, L + P::OFF_X + nx.st * P::XS + j * P::XPLANE + i * 1024);
    } else if (t < 8) {
      const int i = w + (t - 6) * NW;
      if (i >= BM / P::RPG) return;
      unsigned r = (unsigned)nx.blk * BM + (unsigned)i * P::RPG + glr;
      const unsigned last = (unsigned)a.m - 1u;
      r = r < last ? r : last;
      const unsigned piece = COT == 4 ? gsp ^ (((unsigned)i & 1u) << 2) : gsp;
      dma16(rs_g, r * grow + (unsigned)co0 * 2u + piece * 16u, 0u, L + P::OFF_G + nx.st * P::GS + i * 1024);
    } else {
      if (w >= SLOT_PIECES) return;
      unsigned off = (unsigned)nx.blk * SLOT_BYTES + (unsigned)w * 1024u + (unsigned)lane * 16u;
      off = off + 16u <= a.slot_bytes ? off : a.slot_bytes - 16u;
      dma16(rs_s, off, 0u, L + P::OFF_S + nx.st * P::SS + w * 1024);
    }
  };
  constexpr int NTASK = 9;
  // Transposing reads as INLINE ASSEMBLY with counted waits of our own.  Through the builtin, hipcc cannot tell the staged buffers
  // apart and puts s_waitcnt vmcnt(0) in front of every LDS read that follows an LDS-DMA request in program order (SIInsertWaitcnts:
  // "LDS DMA store may alias"): the next block's rows were fully waited for BEFORE the first MFMA of this block — no overlap at all.
  // A read it does not see cannot be made to wait; the price is that the waits for the reads' results are ours too: LDS operations of
  // a wave return in order, so "the fragments of unit u have arrived" = at most as many operations outstanding as were issued
  // after them (lgkmcnt(n), n a compile-time count; scalar loads in flight only make the wait longer, never shorter).
  auto lds_addr = [&](const char* p) -> unsigned { return (unsigned)(size_t)(__attribute__((address_space(3))) const char*)p; };
  auto tr_at = [&](unsigned addr) -> s16x4 {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
  };
  auto tr_off = [&](unsigned addr, auto off_) -> s16x4 {      // ... with a compile-time byte offset in the instruction
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(decltype(off_)::value));
    return v;
  };
  // The (chunk, offset) units of a block form one flat list; the fragments of unit u + 1 (2 * CIT gathered transposing reads, and
  // the 2 * COT out_grad reads when it opens a chunk) are requested BEFORE the CIT * COT MFMAs of unit u are issued (two register
  // sets).  Every wave runs its first three offsets in one unrolled body; waves 0-2 then run their fourth (offsets 24-26) in a
  // second, short one (two whole bodies for 3 and 4 offsets behind one branch made hipcc rename every accumulator: 206 registers
  // instead of 122).  One request of the next block's data follows each of the first nine units (dma_task).
  auto compute_n = [&](int st, const u32x4 (&ad4)[KPW], auto k0_, auto nt_) __attribute__((always_inline)) {
    constexpr int KK0 = decltype(k0_)::value, NT = decltype(nt_)::value, U = NQ * NT;
    const unsigned X = lds_addr(L + P::OFF_X + st * P::XS + (c & 3) * 8);
    // out_grad: this lane's rows of chunk 0, tile swizzle in bits 5-6 (rows are whole 32-byte tiles, so "+" is "|" there and tile t is
    // one xor away); the chunk and the +4 rows go into the instruction's offset field
    const unsigned Gs = lds_addr(L + P::OFF_G + st * P::GS + (8 * g + cc) * P::RBG + (c & 3) * 8) | (g_tile_swz << 5);
    auto load_b = [&](auto q_, s16x8 (&b)[COT]) {
      constexpr int Q = decltype(q_)::value;
#pragma unroll
      for (int t = 0; t < COT; ++t) {
        if constexpr (ABL & 16) { b[t] = s16x8{(short)(Q + t), 1, 2, 3, 4, 5, 6, (short)lane}; continue; }
        const unsigned gt = Gs ^ (unsigned)(t << 5);
        const s16x4 lo = tr_off(gt, std::integral_constant<int, Q * 32 * P::RBG>{});
        const s16x4 hi = tr_off(gt, std::integral_constant<int, (Q * 32 + 4) * P::RBG>{});
        b[t] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    };
    auto load_x = [&](int u, s16x8 (&x)[CIT]) {
      const unsigned pair = ad4[KK0 + u % NT][u / NT];
      const unsigned o_lo = pair & 0xFFFFu, o_hi = pair >> 16;
#pragma unroll
      for (int t = 0; t < CIT; ++t) {
        if constexpr (ABL & 8) { x[t] = s16x8{(short)o_lo, (short)o_hi, 2, 3, 4, 5, (short)t, (short)lane}; continue; }
        const s16x4 lo = tr_at(X + (o_lo ^ (unsigned)(t * 32)));   // xor: the baked offset carries the half swap in bit 5
        const s16x4 hi = tr_at(X + (o_hi ^ (unsigned)(t * 32)));
        x[t] = s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
    };
    // at most N LDS operations of this wave outstanding; the fragments named are results of earlier reads: tying them to the wait keeps
    // hipcc from issuing their MFMAs in front of it
    auto wait_frags = [&](auto n_, s16x8 (&x)[CIT], s16x8 (&b)[COT]) {
      constexpr int N = decltype(n_)::value;
      if constexpr (CIT == 1 && COT == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x[0]), "+v"(b[0]) : "n"(N));
      else if constexpr (CIT == 1 && COT == 2) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(x[0]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
      else if constexpr (CIT == 2 && COT == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x[0]), "+v"(x[1]), "+v"(b[0]), "+v"(b[1]) : "n"(N));
      else asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(x[0]), "+v"(x[1]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]) : "n"(N));
    };
    static_assert((CIT == 1 && (COT == 1 || COT == 2)) || (CIT == 2 && (COT == 2 || COT == 4)), "wait_frags lists the fragments of these shapes");
    auto mma = [&](int kk, const s16x8 (&x)[CIT], const s16x8 (&b)[COT]) {
#pragma unroll
      for (int ta = 0; ta < CIT; ++ta)
#pragma unroll
        for (int tb = 0; tb < COT; ++tb) {
          if constexpr (ABL & 4) acc[kk][ta][tb][0] += (float)(x[ta][0] ^ b[tb][7]);
          else acc[kk][ta][tb] = mfma<F16>(x[ta], b[tb], acc[kk][ta][tb]);
        }
    };
    constexpr int RX = (ABL & 8) ? 0 : 2 * CIT, RB = (ABL & 16) ? 0 : 2 * COT;    // LDS reads of one unit's rows / one chunk's out_grad
    s16x8 b0[COT], b1[COT], x0[CIT], x1[CIT];
    load_b(std::integral_constant<int, 0>{}, b0);
    load_x(0, x0);
    auto load_b_rt = [&](int q1) {      // q1 is a constant after unrolling; the offset field wants it as a template argument
      if (q1 == 1) load_b(std::integral_constant<int, 1>{}, b1);
      else if (q1 == 2) load_b(std::integral_constant<int, 2>{}, b0);
      else if (q1 == 3) load_b(std::integral_constant<int, 3>{}, b1);
    };
    static_assert(NQ == 4, "load_b_rt lists the chunks");
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = u / NT, kk = u % NT;
      constexpr int dummy = 0;
      (void)dummy;
      const bool nextb = u + 1 < U && (u + 1) % NT == 0;
      if (u + 1 < U) {
        const int q1 = (u + 1) / NT;
        if (nextb) load_b_rt(q1);
        if ((u + 1) & 1) load_x(u + 1, x1); else load_x(u + 1, x0);
      }
      // outstanding behind unit u's fragments: what was just requested for unit u + 1
      auto wait_u = [&](s16x8 (&x)[CIT], s16x8 (&b)[COT]) {
        if (u + 1 >= U) wait_frags(std::integral_constant<int, 0>{}, x, b);
        else if (nextb) wait_frags(std::integral_constant<int, RX + RB>{}, x, b);
        else wait_frags(std::integral_constant<int, RX>{}, x, b);
      };
      if (u & 1) { if (q & 1) { wait_u(x1, b1); mma(KK0 + kk, x1, b1); } else { wait_u(x1, b0); mma(KK0 + kk, x1, b0); } }
      else { if (q & 1) { wait_u(x0, b1); mma(KK0 + kk, x0, b1); } else { wait_u(x0, b0); mma(KK0 + kk, x0, b0); } }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (KK0 == 0) {
        static_assert(KK0 != 0 || U >= NTASK, "every task of the next block needs a unit to follow");
        if (u < NTASK) dma_task(u);
      }
    }
  };
  auto compute = [&](int st, unsigned bmask) __attribute__((always_inline)) {
    // the address quads of all four offsets, read (by ordinary loads the compiler sees) BEFORE the first request of the next block
    const uint16_t* const tab = (const uint16_t*)(L + P::OFF_S + st * P::SS);
    u32x4 ad4[KPW];
#pragma unroll
    for (int kk = 0; kk < KPW; ++kk) {
      const int k0 = w + kk * NW, k = k0 < 27 ? k0 : 26;
      const uint16_t* t = (bmask >> (k / 9)) & 1u ? baked : tab;      // wave-uniform: this offset's plane was converted in the kernel
      ad4[kk] = *(const u32x4*)(t + ((k * 4 + g) * 4 + cc) * (NQ * 2));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    compute_n(st, ad4, std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
    if (w + 3 * NW < 27) compute_n(st, ad4, std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
  };

  if (blk_beg < blk_end) {
    // Pipeline: while block n multiplies, the rows, out_grad and address table of block n + 1 are in flight (its header came a
    // block earlier) — one barrier per block, no work between a block's last MFMA and the next block's first but that barrier.
    int lo[3], cnt[3], lo_n[3], cnt_n[3];
    unsigned rawm = 0u, rawm_n = 0u;
    Hdr raw = load_hdr_raw(blk_beg);
    use_hdr(raw, lo, cnt, rawm);
    if (blk_beg + 1 < blk_end) { raw = load_hdr_raw(blk_beg + 1); use_hdr(raw, lo_n, cnt_n, rawm_n); }
    stage_x(lo, cnt, 0, 0);
    stage_g(blk_beg, 0);
    stage_slots(blk_beg, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef BEVAMD_WGS_PROF
    unsigned long long t_issue = 0, t_mul = 0, t_wait = 0, t_bar = 0, t0 = __builtin_readcyclecounter(), t1;
#define WGS_TICK(acc_) do { t1 = __builtin_readcyclecounter(); acc_ += t1 - t0; t0 = t1; } while (0)
#else
#define WGS_TICK(acc_) do { } while (0)
#endif
    int st = 0;
    for (int blk = blk_beg; blk < blk_end; ++blk, st ^= 1) {
      const bool more = blk + 1 < blk_end;
      if (blk + 2 < blk_end) raw = load_hdr_raw(blk + 2);
      nx.on = more && !(ABL & 1);
      nx.blk = blk + 1;
      nx.st = st ^ 1;
#pragma unroll
      for (int j = 0; j < 3; ++j) { nx.lo[j] = lo_n[j]; nx.cnt[j] = cnt_n[j]; }
      if (rawm) {                                   // rare: a plane with raw slots — converted before the block multiplies
        bake(cnt, rawm, rawm, 0, st);
        __syncthreads();
      }
      WGS_TICK(t_issue);
      compute(ABL & 1 ? 0 : st, rawm);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WGS_TICK(t_mul);
      if (rawm) {
        int longest = cnt[0] > cnt[1] ? cnt[0] : cnt[1];
        longest = longest > cnt[2] ? longest : cnt[2];
        for (int pbase = CAP; pbase < longest; pbase += CAP) {   // ... and its further pieces of CAP rows, synchronously
          __syncthreads();
          stage_x(lo, cnt, pbase, st);
          bake(cnt, rawm, 7u, pbase, st);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          nx.on = false;
          compute(st, 7u);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WGS_TICK(t_wait);
      __syncthreads();                              // the next block has landed; everybody is done with this one
      WGS_TICK(t_bar);
#pragma unroll
      for (int j = 0; j < 3; ++j) { lo[j] = lo_n[j]; cnt[j] = cnt_n[j]; }
      rawm = rawm_n;
      if (blk + 2 < blk_end) use_hdr(raw, lo_n, cnt_n, rawm_n);
    }
#ifdef BEVAMD_WGS_PROF
    if (a.prof && lane == 0) {
      atomicAdd(a.prof + 0, t_issue); atomicAdd(a.prof + 1, t_mul); atomicAdd(a.prof + 2, t_wait); atomicAdd(a.prof + 3, t_bar);
      atomicAdd(a.prof + 4, 1ull);
    }
#endif
#undef WGS_TICK
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
// cin == cout in {16, 32, 64, 128}, or cout == 2 cin for cin in {16, 32, 64} (round 6: the strided 3x3x3 layers, whose metadata has the
// same form): a workgroup owns at most 32 input x 64 output channels (128 accumulator registers per lane)
static inline bool shape_for(int cin, int cout, Shape& s) {
  const bool widths = cin == 16 || cin == 32 || cin == 64 || cin == 128;
  if (!widths || !(cout == cin || (cout == 2 * cin && cin <= 64))) return false;
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
  if (s.cit == 1 && s.cot == 2) return launch_one<F16, 1, 2, 256>(a, stream);
  if (s.cit == 2 && s.cot == 2) return launch_one<F16, 2, 2, 192>(a, stream);
  if (s.cit == 2 && s.cot == 4) return launch_one<F16, 2, 4, 192>(a, stream);
  set_error("spconv_conv_wgrad_slab: no kernel for %d x %d tiles", s.cit, s.cot);
  return BEVAMD_ERR_UNSUPPORTED;
}

}  // namespace wgslab
}  // namespace bevamd
