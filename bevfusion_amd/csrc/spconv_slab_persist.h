// Staged-rows submanifold convolution, third cut: the register-filter kernel of spconv_slab_regw.h made PERSISTENT.
//
// What the second cut still pays per block (profiles/r02_slab_ablation.txt: prologue + epilogue = 20-35 % of a layer): two
// dependent memory round trips before the first MFMA (block header -> row DMA) with only 1-3 workgroups per CU to hide
// them, and an epilogue during which the workgroup's share of the CU idles.  Launches are sized by row CAPACITY too (the row
// count lives on the device), so at 8 frames 80 % of the launched workgroups only find out that they have no block.
//
// Here a workgroup walks every gx-th block of its XCD's contiguous block range (gx = resident workgroups per XCD).  The unit
// of the pipeline is the PIECE (block, channel pass, kernel plane, <= CAP rows of that plane's range), and a block boundary
// is just another piece boundary: while the last piece of block b multiplies, the rows and the 9-tap slot sub-table of the
// first piece of the workgroup's next block are already in flight (its header was fetched a block earlier), the filter
// ring simply continues, and the epilogue of block b overlaps that DMA.  The slot table is staged per plane and per wave row
// (each wave gather-DMAs the [9 taps][its rows] sub-table it reads: no barrier between its arrival and its use), double
// buffered with the rows, and the epilogue has its own scratch, so nothing aliases a buffer with a request in flight.
//
// Same metadata (slab_build_kernel), same summation order, bit-identical results to spconv_slab_regw.h / spconv_slab.h.
#pragma once
#include "spconv_slab_regw.h"

namespace bevamd {
namespace slab {

template <int KC, int CIN, int NT, int MT, int RW, int CW, int CAP>
struct PlanP {
  typedef PlanR<KC, CIN, NT, MT, RW, CW, CAP> R;
  static constexpr int NW = R::NW, BM = R::BM, NTW = R::NTW, RB = R::RB, PPR = R::PPR, RPI = R::RPI, CH = R::CH, CPB = R::CPB,
                       NH = R::NH, NB = R::NB, WD = R::WD, NXB = R::NXB, XB = R::XB, PX = R::PX, NX = R::NX;
  static constexpr int G = 16 * MT;                                 // rows of a wave
  static constexpr int SE = TAPS * G * 2 / 16;                      // 16-byte entries of a wave's slot sub-table [9 taps][G rows]
  static constexpr int NS = (SE + 63) / 64;                         // gather-DMA instructions per wave
  static constexpr int SW = NS * 1024;                              // LDS bytes of one such sub-table
  static constexpr int SB = RW * SW;                                // ... of all wave rows
  static constexpr int OFF_X = 0;
  static constexpr int OFF_SLOT = NXB * XB;
  static constexpr int OFF_EPI = OFF_SLOT + NXB * SB;
  static constexpr int OFF_DUMP = OFF_EPI + NW * EpiScratch<NTW>::U4 * 16;
  static constexpr int BYTES = OFF_DUMP + 1024;
  static_assert(WD * NB + NX + NS < 60, "vmcnt is a 6-bit counter");
  static_assert(RW * 16 * MT != BAKED_ROWS, "64-row blocks carry baked slot metadata (spconv_slab_meta.h): not decoded by the persistent kernel");
  // register budget handed to hipcc (waves per SIMD): what the LDS plan admits, but no more than the accumulators + filter
  // ring leave room for.  Without it the persistent kernels came out 50-80 registers above their one-block twins — one wave per
  // SIMD less — for no use.
  static constexpr int WGS = 160 * 1024 / BYTES;
  static constexpr int WPE_LDS = WGS * NW / 4 < 1 ? 1 : WGS * NW / 4;
  static constexpr int WPE_REG = 512 / (MT * NTW * 4 + (WD + 1) * NB * 4 + 88);
  static constexpr int WPE = WPE_LDS < WPE_REG ? WPE_LDS : (WPE_REG < 1 ? 1 : WPE_REG);
};

struct Piece {
  int h, j, q;   // channel pass, kernel plane, piece of the plane's range
  bool nb;       // belongs to the workgroup's NEXT block (only ever true for the piece after the current one)
  bool done;
};

template <int DT, int KC, int CIN, int NT, int MT, int RW, int CW, int CAP>
__global__ __launch_bounds__(RW * CW * 64) __attribute__((amdgpu_waves_per_eu(PlanP<KC, CIN, NT, MT, RW, CW, CAP>::WPE)))
void spconv_slabp_kernel(SlabArgs sa) {
  typedef PlanP<KC, CIN, NT, MT, RW, CW, CAP> P;
  typedef WaveTile<DT, (CIN > 64 ? 64 : CIN), P::NTW, MT, (CIN > 64 ? 64 : CIN) / 32> WT;   // accumulators + epilogue only
  typedef typename Num<DT>::T T;
  extern __shared__ u32x4 lds[];
  char* const L = (char*)lds;
  const Args& a = sa.a;
  const int m = a.m_dev ? (*a.m_dev < a.m_cap ? *a.m_dev : a.m_cap) : a.m_cap;
  const int nblk = (m + P::BM - 1) / P::BM;
  // XCD x owns the contiguous block range [x*per, (x+1)*per); its gx workgroups walk it round-robin, so the blocks in flight
  // on an XCD at any time are neighbours (they share staged rows in its L2)
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int per = (nblk + 7) >> 3;
  const int blk_end = (xcd + 1) * per < nblk ? (xcd + 1) * per : nblk;
  int blk = xcd * per + bix;
  if (blk >= blk_end) return;   // the whole workgroup leaves together
  int blk_n = blk + gx;         // this workgroup's next block (>= blk_end: none)
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for hipcc too (no waterfall loops around the DMA)
  const int wr = w / CW, wc = w % CW;
  const int c = lane & 15, g4 = lane >> 4;

  // block headers: lane j < 3 holds (first row, row count) of plane j; current and next block
  auto load_hdr = [&](int b) { return sa.hdr[(size_t)b * PLANES + (lane < PLANES ? lane : 0)]; };
  int2 hc = load_hdr(blk);
  int2 hn = blk_n < blk_end ? load_hdr(blk_n) : make_int2(0, 0);
  if (tid < P::NXB * P::PPR) {   // zero rows behind both X buffers
    const int b = tid / P::PPR, p = tid % P::PPR;
    *(u32x4*)(L + P::OFF_X + b * P::XB + CAP * P::RB + p * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  auto live_of = [&](const int2& h) { return (unsigned)__builtin_amdgcn_readfirstlane((int)__ballot(lane < PLANES && h.y > 0)); };
  unsigned live_c = live_of(hc);
  auto lo_of = [&](const Piece& u) { return u.nb ? __builtin_amdgcn_readlane(hn.x, u.j) : __builtin_amdgcn_readlane(hc.x, u.j); };
  auto cnt_of = [&](const Piece& u) { return u.nb ? __builtin_amdgcn_readlane(hn.y, u.j) : __builtin_amdgcn_readlane(hc.y, u.j); };
  auto next_plane = [&](unsigned live, int from) {   // first plane >= from with rows, PLANES if none
    const unsigned rest = from < PLANES ? live >> from : 0u;
    return rest ? from + (int)__builtin_ctz(rest) : PLANES;
  };
  // the piece after `u` (u lies in the current block)
  auto advance = [&](Piece u) {
    if (u.done) return u;
    if ((u.q + 1) * CAP < cnt_of(u)) { ++u.q; return u; }
    u.q = 0;
    const int j2 = next_plane(live_c, u.j + 1);
    if (j2 < PLANES) { u.j = j2; return u; }
    if (u.h + 1 < P::NH) { ++u.h; u.j = next_plane(live_c, 0); return u; }
    if (blk_n < blk_end) return Piece{0, next_plane(live_of(hn), 0), 0, true, false};   // a live block always has its centre plane
    u.done = true;
    return u;
  };

  const unsigned row_bytes = (unsigned)a.feat_stride * 2u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.feat, 0, (unsigned)a.n_in * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wimg, 0, sa.wimg_bytes, 0x00020000);
  const unsigned slot_bytes = (unsigned)((a.m_cap + P::BM - 1) / P::BM) * (unsigned)(27 * P::BM * 2);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)sa.slots, 0, slot_bytes, 0x00020000);
  char* const dump = L + P::OFF_DUMP;
  const unsigned lr = (unsigned)(lane / P::PPR), sp = (unsigned)(lane % P::PPR);
  const unsigned lane_piece_off = (sp ^ RowSwz<KC>::of(lr)) * 16u;
  static_assert(P::RPI % 8 == 0, "swizzle must not depend on the instruction index");
  // rows and slot sub-table of piece u -> buffer pair pb: exactly NX + NS 1-KiB requests per wave (row pieces past the range
  // re-read its last row into rows no slot refers to; a finished `u` sends everything to the dump)
  auto stage = [&](const Piece& u, int pb) {
    const bool go = !u.done;
    const int n = go ? cnt_of(u) - u.q * CAP : 1;
    const unsigned rows = (unsigned)(n < CAP ? n : CAP);
    const unsigned soff = go ? (unsigned)(lo_of(u) + u.q * CAP) * row_bytes + (unsigned)(u.h * KC * 2) : 0u;
    char* dst = L + P::OFF_X + pb * P::XB;
#pragma unroll
    for (int t = 0; t < P::NX; ++t) {
      const int i = w + t * P::NW;
      unsigned r = (unsigned)(i * P::RPI) + lr;
      r = r < rows ? r : rows - 1u;
      dma16(rs_x, r * row_bytes + lane_piece_off, soff, (i < P::PX && go) ? dst + i * 1024 : dump);
    }
    // slots: every wave gathers the [9 taps][its G rows] sub-table it reads itself (both waves of a wave row write the same
    // bytes to the same place), so reading it needs this wave's own requests only — no barrier
    const unsigned sbase = go ? ((unsigned)(u.nb ? blk_n : blk) * 27u + (unsigned)(u.j * TAPS)) * (unsigned)(P::BM * 2) : 0u;
    char* sdst = L + P::OFF_SLOT + pb * P::SB + wr * P::SW;
#pragma unroll
    for (int t = 0; t < P::NS; ++t) {
      int e = t * 64 + lane;
      e = e < P::SE ? e : P::SE - 1;
      const unsigned tap = (unsigned)e / (unsigned)(P::G / 8), seg = (unsigned)e % (unsigned)(P::G / 8);
      dma16(rs_s, (tap * P::BM + (unsigned)(wr * P::G)) * 2u + seg * 16u, sbase, go ? sdst + t * 1024 : dump);
    }
  };
  // this wave's filter fragments of tap d (0..8) of piece u: NB coalesced 16-byte loads
  auto load_w = [&](const Piece& u, int d, u32x4 (&wf)[P::NB]) {
    const int k = u.j * TAPS + d;
#pragma unroll
    for (int cc = 0; cc < P::CH; ++cc)
#pragma unroll
      for (int nt = 0; nt < P::NTW; ++nt)
        wf[cc * P::NTW + nt] = __builtin_amdgcn_raw_buffer_load_b128(
            rs_w, (unsigned)lane * 16u, (unsigned)(((k * P::CPB + u.h * P::CH + cc) * NT + wc * P::NTW + nt) * 1024), 0);
  };
  // 16-bit slots of tap d of the plane staged in slot buffer pb (read two taps ahead), and their LDS row addresses
  auto load_slots = [&](int pb, int d, unsigned (&raw)[MT]) {
    const uint16_t* sl = (const uint16_t*)(L + P::OFF_SLOT + pb * P::SB + wr * P::SW) + d * P::G + c;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) raw[mt] = (unsigned)sl[mt * 16];
  };
  auto to_offsets = [&](const Piece& u, const unsigned (&raw)[MT], unsigned (&xo)[MT]) {
    const unsigned pbase = (unsigned)(u.q * CAP);
    const unsigned plive = (unsigned)cnt_of(u) - pbase;
    const unsigned prow = plive < (unsigned)CAP ? plive : (unsigned)CAP;   // rows of this piece
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      unsigned e = raw[mt] - pbase;        // NO_SLOT - pbase stays >= prow
      e = e < prow ? e : (unsigned)CAP;    // outside the piece: the zero row
      xo[mt] = e * P::RB + (RowSwz<KC>::of(e) << 20);
    }
  };

  // output channels [c0, c0 + 16*NTW) of the rows: the epilogue sees a narrower convolution
  Args aw = a;
  const int c0 = wc * P::NTW * 16;
  aw.out = (void*)((T*)a.out + c0);
  if (a.bias) aw.bias = (const void*)((const T*)a.bias + c0);
  if (a.scale) { aw.scale = a.scale + c0; aw.shift = a.shift + c0; }
  if (a.residual) aw.residual = (const void*)((const T*)a.residual + c0);
  aw.cout = a.cout - c0;
  u32x4* const eps = (u32x4*)(L + P::OFF_EPI) + w * EpiScratch<P::NTW>::U4;
  WT wt;
  wt.init(aw, blk * P::BM + wr * 16 * MT, m, nullptr, eps);

  Piece sub[2];   // this piece, the next
  sub[0] = Piece{0, next_plane(live_c, 0), 0, false, false};
  sub[1] = advance(sub[0]);
  u32x4 wf[P::WD + 1][P::NB];    // filter ring: tap d of a piece lives in set d % 3
  unsigned raw[P::WD + 1][MT];   // slot ring, same indexing
  stage(sub[0], 0);
  load_w(sub[0], 0, wf[0]);
  load_w(sub[0], 1, wf[1]);
  wait_dma<0>();
  __syncthreads();   // first piece landed; also publishes the zero rows
  load_slots(0, 0, raw[0]);
  load_slots(0, 1, raw[1]);
  int pb = 0;
  // Two nested loops — blocks outside, the pieces of a block inside — so that the accumulators are loop-carried by the piece
  // loop only (MFMA -> MFMA, they stay in AGPRs); with the epilogue inside one flat loop hipcc moved all of them AGPR -> VGPR
  // -> AGPR at every piece.
  for (;;) {
    for (;;) {
      const char* X = L + P::OFF_X + pb * P::XB;
      constexpr int U = TAPS * P::CH;   // reduction units of a piece: (tap, 32-channel chunk)
      unsigned xo[2][MT];
      u32x4 xa[2][MT];
      auto fetch = [&](int i) {
        const int d = i / P::CH, cc = i % P::CH;
  #pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          xa[i & 1][mt] = *(const u32x4*)(X + (xo[d & 1][mt] & 0xFFFFFu) + (((unsigned)(cc * 4 + g4)) ^ (xo[d & 1][mt] >> 20)) * 16);
      };
      to_offsets(sub[0], raw[0], xo[0]);
      fetch(0);
  #pragma unroll
      for (int d = 0; d < TAPS; ++d) {
        // requests: the filter of the tap WD ahead (the next piece's first taps at the end of this one); at the first tap the
        // rows + slots of the next piece, AFTER the filter request — a filter load issued behind them could not complete first
        {
          const int dn = d + P::WD;
          const Piece& un = dn < TAPS ? sub[0] : (sub[1].done ? sub[0] : sub[1]);   // past the last piece: a valid, unused load
          load_w(un, dn % TAPS, wf[dn % (P::WD + 1)]);
          load_slots(dn < TAPS ? pb : pb ^ 1, dn % TAPS, raw[dn % (P::WD + 1)]);   // next piece: requested at tap 0, landed since tap 3
        }
        if (d == 0) stage(sub[1], pb ^ 1);
  #pragma unroll
        for (int cc = 0; cc < P::CH; ++cc) {
          const int i = d * P::CH + cc;
          if (i + 1 < U) {
            if ((i + 1) % P::CH == 0) to_offsets(sub[0], raw[(d + 1) % (P::WD + 1)], xo[(d + 1) & 1]);
            fetch(i + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
          for (int nt = 0; nt < P::NTW; ++nt)
  #pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              wt.acc[mt][nt] = mfma<DT>(wf[d % (P::WD + 1)][cc * P::NTW + nt], xa[i & 1][mt], wt.acc[mt][nt]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // the next piece's rows and slots (requested nine taps ago, in front of all but the two newest filter sets) have
      // landed; every wave is done reading this piece
      wait_dma<P::WD * P::NB>();
      barrier_keep_dma();
      if (sub[1].done || sub[1].nb) break;
      sub[0] = sub[1];
      sub[1] = advance(sub[0]);
      pb ^= 1;
    }
    // Block boundary: finish block `blk` while the first piece of the next one is landing.  The header of the block after the
    // next is requested BEFORE the epilogue and the two filter sets in flight are requested again AFTER it: both ways into
    // the piece loop then leave the same requests pending (two filter sets, nothing younger), and hipcc keeps its counted
    // vmcnt at the loop head instead of draining the queue at every piece.
    const int blk_nn = blk_n + gx;
    const int2 hnn = blk_nn < blk_end ? load_hdr(blk_nn) : make_int2(0, 0);
    wt.store(aw);
    if (sub[1].done) break;
    blk = blk_n;
    blk_n = blk_nn;
    hc = hn;
    hn = hnn;
    live_c = live_of(hc);
    wt.init(aw, blk * P::BM + wr * 16 * MT, m, nullptr, eps);
    sub[1].nb = false;
    sub[0] = sub[1];
    sub[1] = advance(sub[0]);
    pb ^= 1;
    load_w(sub[0], 0, wf[0]);
    load_w(sub[0], 1, wf[1]);
  }
}

}  // namespace slab
}  // namespace bevamd
