// Staged-rows convolution for the NARROW layers (Cin padded to 8 or 16: 16- or 32-byte feature rows) — level 1 of the
// SparseEncoder: the 5->16 input layer, the four 16->16 submanifold layers and the strided 16->32 convolution that leaves it.
//
// Same operator, metadata and arithmetic as the wider slab kernels (spconv_slab_regw.h): rows in ascending linear index, per
// block of BM output rows and kernel plane kx the contiguous input range (hdr) and 16-bit slots into it.  What is different:
//   * rows are so short that ALL THREE plane ranges of a block fit in LDS at once (3 x CAP x 32 B): one LDS-DMA burst, one
//     barrier, then the whole reduction without any further synchronisation;
//   * a 32-wide MFMA reduction chunk spans 2 (Cin 16) or 4 (Cin 8) kernel taps, exactly as in the gather kernels
//     (spconv_tile.h: lane group g carries 8 consecutive channels of ONE tap), so a chunk may straddle two planes — no issue
//     when every plane is resident.  Chunk order and accumulator chain are the gather kernels': results are bit-identical;
//   * the whole filter image (7 or 14 chunks) lives in REGISTERS for the lifetime of a persistent workgroup;
//   * a block whose range exceeds CAP rows in some plane (rare: measured never on LiDAR frames) reads its operands straight
//     from global memory with the same addressing — slower, same result.
// Works for the strided 3x3x3 convolution as well: only the metadata differs (spconv_indice.hip: sp_slab_from_sorted_kernel).
//
// What bounds it (round 5, compile-time ablation with -DBEVAMD_SMALL_ABL, EXPERIMENTS C.10): the LDS PIPE.  16 -> 16 at 8 frames
// (1.28 M rows): 53 us; without the reduction 21, without the MFMAs but with every operand read 52.  An operand fragment is read
// for ONE 16-channel output tile (the wide layers reuse it for 4-8), so a layer moves rows x 27 x 32 B = 1.15 GB through
// 128 B/clk/CU.  Neither a bank swizzle, nor all slots up front + operands two chunks ahead, nor slot entries baked into LDS rows
// (one shift-add per fragment instead of six instructions), nor a software pipeline over the blocks moved the time.
#pragma once
#include <type_traits>
#include "spconv_slab.h"

#ifndef BEVAMD_SMALL_ABL
#define BEVAMD_SMALL_ABL 0   // profiling builds only (wrong results by design): 1 no row DMA, 2 no slot DMA, 4 no reduction,
#endif                       // 8 no epilogue, 16 no MFMA (operands still read), 32 no header load

namespace bevamd {
namespace slab {
constexpr int SMALL_ABL = BEVAMD_SMALL_ABL;

// NW waves form an (NW / CW) x CW grid: wave (r, c) owns 16*MT rows x (NT / CW) 16-channel output tiles (the 16->32 layer
// splits its two output tiles over two waves: 56 filter registers per wave instead of 112, 4 waves per SIMD instead of 1).
template <int CIN, int NT, int MT, int NW, int CW, int CAP>
struct PlanS {
  static_assert(CIN == 8 || CIN == 16, "narrow rows: 8 or 16 channels");
  static_assert(NW % CW == 0 && NT % CW == 0, "bad wave grid");
  static constexpr int RWS = NW / CW;                // wave rows
  static constexpr int NTW = NT / CW;                // output tiles per wave
  static constexpr int BM = RWS * 16 * MT;
  static constexpr int RB = CIN * 2;                 // bytes per staged row
  static constexpr int PPR = RB / 16;                // 16-byte pieces per row
  static constexpr int RPI = 64 / PPR;               // rows per DMA instruction (1 KiB)
  static constexpr int TPC = 32 / CIN;               // kernel taps per 32-wide reduction chunk
  static constexpr int GPT = 4 / TPC;                // lane groups (8 channels each) per tap
  static constexpr int NCH = (27 + TPC - 1) / TPC;   // reduction chunks
  static constexpr int ZROW = PLANES * CAP;          // the all-zero row (missing neighbour, taps past the 27th)
  static constexpr int OFF_X = 0;
  static constexpr int XBYTES = ((ZROW + 1) * RB + 1023) / 1024 * 1024;
  static constexpr int OFF_SLOT = XBYTES;
  static constexpr int SLOT_KIB = (27 * BM * 2 + 1023) / 1024;   // DMA pieces of a block's slot table
  static constexpr int OFF_EPI = OFF_SLOT + 27 * BM * 2;
  static constexpr int EPI_BYTES = NW * EpiScratch<NTW>::U4 * 16;
  static constexpr int BYTES = OFF_EPI + (EPI_BYTES > 1024 ? EPI_BYTES : 1024);
  static_assert(CAP % RPI == 0, "CAP must be a whole number of DMA instructions");
  static_assert((ZROW + 1) * RB < (1 << 20), "LDS row offsets");
};

template <int DT, int CIN, int NT, int MT, int NW, int CW, int CAP>
__global__ __launch_bounds__(NW * 64) void spconv_slabs_kernel(SlabArgs sa) {
  typedef PlanS<CIN, NT, MT, NW, CW, CAP> P;
  typedef typename Num<DT>::T T;
  constexpr int NTW = P::NTW;
  typedef WaveTile<DT, 32, NTW, MT, 1> WT;   // accumulators + epilogue only
  extern __shared__ u32x4 lds[];
  char* const L = (char*)lds;
  const Args& a = sa.a;
  const int m = a.m_dev ? (*a.m_dev < a.m_cap ? *a.m_dev : a.m_cap) : a.m_cap;
  const int nblk = (m + P::BM - 1) / P::BM;
  // persistent: XCD x walks every gx-th block of its contiguous eighth of the blocks (neighbouring blocks share rows in its L2)
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int per = (nblk + 7) >> 3;
  const int bend = (xcd + 1) * per < nblk ? (xcd + 1) * per : nblk;
  int blk = xcd * per + bix;
  if (blk >= bend) return;   // the whole workgroup leaves together
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w / CW, wc = w % CW;
  const int c = lane & 15, g4 = lane >> 4;

  // this wave's share of the filter image -> registers (fragment order: one coalesced 16-byte load per (chunk, 16 output channels))
  u32x4 wf[P::NCH][NTW];
#pragma unroll
  for (int ch = 0; ch < P::NCH; ++ch)
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) wf[ch][nt] = ((const u32x4*)a.wimg)[(ch * NT + wc * NTW + nt) * 64 + lane];
  // output channels [c0, c0 + 16*NTW) of the rows: the epilogue sees a narrower convolution
  Args aw = a;
  const int c0 = wc * NTW * 16;
  aw.out = (void*)((T*)a.out + c0);
  if (a.bias) aw.bias = (const void*)((const T*)a.bias + c0);
  if (a.scale) { aw.scale = a.scale + c0; aw.shift = a.shift + c0; }
  if (a.residual) aw.residual = (const void*)((const T*)a.residual + c0);
  aw.cout = a.cout - c0;

  if (tid < P::PPR) *(u32x4*)(L + P::ZROW * P::RB + tid * 16) = u32x4{0u, 0u, 0u, 0u};
  uint16_t* const slot = (uint16_t*)(L + P::OFF_SLOT);
  const unsigned row_bytes = (unsigned)a.feat_stride * 2u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.feat, 0, (unsigned)a.n_in * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)sa.slots, 0, sa.slot_bytes, 0x00020000);
  const unsigned lr = (unsigned)(lane / P::PPR), lp = (unsigned)(lane % P::PPR) * 16u;
  // this lane's share of a reduction chunk: tap (ch * TPC + tsub), 16-byte piece `piece` of the neighbour's row
  const int tsub = g4 / P::GPT;
  const unsigned piece = (unsigned)(g4 % P::GPT) * 16u;

  for (; blk < bend; blk += gx) {
    __syncthreads();   // every wave is done with the previous block's rows and slots
    int2 hl = make_int2(blk * P::BM > 300 ? blk * P::BM - 300 : 0, 300);
    if constexpr (!(SMALL_ABL & 32)) hl = sa.hdr[(size_t)blk * PLANES + (lane < PLANES ? lane : 0)];
    int lo[PLANES], cnt[PLANES];
#pragma unroll
    for (int j = 0; j < PLANES; ++j) {
      lo[j] = __builtin_amdgcn_readlane(hl.x, j);
      cnt[j] = __builtin_amdgcn_readlane(hl.y, j);
    }
    const bool big = cnt[0] > CAP || cnt[1] > CAP || cnt[2] > CAP;   // workgroup-uniform
    // slot table of the block -> LDS, by DMA as well (whole KiB pieces: the last one may run into the epilogue scratch, which
    // nobody uses before the stores at the end of the block; past the end of the buffer the descriptor returns zeros)
    if constexpr (!(SMALL_ABL & 2))
    for (int i = w; i < P::SLOT_KIB; i += NW)
      dma16(rs_s, (unsigned)lane * 16u, (unsigned)blk * (unsigned)(27 * P::BM * 2) + (unsigned)i * 1024u, (char*)slot + i * 1024);
    if (!big && !(SMALL_ABL & 1)) {
#pragma unroll
      for (int j = 0; j < PLANES; ++j) {
        const int np = (cnt[j] + P::RPI - 1) / P::RPI;
        const unsigned soff = (unsigned)lo[j] * row_bytes;
        for (int i = w; i < np; i += NW) {
          unsigned r = (unsigned)(i * P::RPI) + lr;
          r = r < (unsigned)cnt[j] ? r : (unsigned)cnt[j] - 1u;
          dma16(rs_x, r * row_bytes + lp, soff, L + (j * CAP + i * P::RPI) * P::RB);
        }
      }
    }
    wait_dma<0>();
    __syncthreads();

    WT wt;
    wt.init(aw, blk * P::BM + wr * 16 * MT, m, nullptr, (u32x4*)(L + P::OFF_EPI) + w * EpiScratch<NTW>::U4);
    const uint16_t* sl = slot + wr * 16 * MT + c;
    // The reduction, once per operand source (compile-time: no branch inside the chunk loop).  BIG = false: rows are resident,
    // an operand is one ds_read_b128 at (plane * CAP + slot) * RB; BIG = true: a buffer load at (lo[plane] + slot) * row_bytes.
    auto reduce = [&](auto big_tag) {
      constexpr bool BIG = decltype(big_tag)::value;
      auto offsets = [&](int ch, unsigned (&xo)[MT]) {
        const int tap = ch * P::TPC + tsub;
        const bool tap_ok = tap < 27;
        const int tp = tap_ok ? tap : 26;
        const int j = tp / TAPS;
        const unsigned pbase = BIG ? (unsigned)(j == 0 ? lo[0] : j == 1 ? lo[1] : lo[2]) : (unsigned)(j * CAP);
        unsigned sv[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) sv[mt] = (unsigned)sl[tp * P::BM + mt * 16];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const bool hit = tap_ok && sv[mt] != NO_SLOT;
          if constexpr (BIG) xo[mt] = hit ? (pbase + sv[mt]) * row_bytes + piece : OOB;
          else xo[mt] = (hit ? pbase + sv[mt] : (unsigned)P::ZROW) * P::RB + piece;
        }
      };
      auto fetch = [&](const unsigned (&xo)[MT], u32x4 (&x)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          if constexpr (BIG) x[mt] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xo[mt], 0, 0);
          else x[mt] = *(const u32x4*)(L + xo[mt]);
        }
      };
      unsigned xo[2][MT];
      u32x4 xa[2][MT];
      offsets(0, xo[0]);
      fetch(xo[0], xa[0]);
      if (P::NCH > 1) offsets(1, xo[1]);
#pragma unroll
      for (int ch = 0; ch < P::NCH; ++ch) {
        if (ch + 1 < P::NCH) fetch(xo[(ch + 1) & 1], xa[(ch + 1) & 1]);
        if (ch + 2 < P::NCH) offsets(ch + 2, xo[ch & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if constexpr (SMALL_ABL & 16) asm volatile("" ::"v"(xa[ch & 1][mt]), "v"(wf[ch][nt]));
            else wt.acc[mt][nt] = mfma<DT>(wf[ch][nt], xa[ch & 1][mt], wt.acc[mt][nt]);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if constexpr (!(SMALL_ABL & 4)) {
      if (big) reduce(std::true_type{});
      else reduce(std::false_type{});
    }
    if constexpr (!(SMALL_ABL & 8)) wt.store(aw);
    else if (blk < 0) wt.store(aw);   // (keeps the accumulators alive)
  }
}

}  // namespace slab
}  // namespace bevamd
