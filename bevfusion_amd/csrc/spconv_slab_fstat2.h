// Staged-rows submanifold convolution, fifth cut: filter-stationary WAVE PAIRS (32 -> 32 channels), two waves per SIMD.
//
// The filter-stationary kernel (spconv_slab_fstat.h) is one in-order wave per SIMD holding the whole 27 x 32 x 32 filter in 216
// registers.  Taken apart in round 6 (BEVAMD_SLABF_EXP, EXPERIMENTS D.11: 161 us; 121 without the epilogue, 133 without the row
// requests, 119 with a quarter of the MFMAs, 55 with none of it) every part is paid in full: nothing overlaps.  Here a workgroup is a
// PAIR of waves that split the INPUT channels: wave w keeps W[k][16 w .. 16 w + 15][:] — 27 A fragments of v_mfma_f32_32x32x16 =
// 108 registers, so two waves fit a SIMD (four pairs per CU) — stages only ITS 32-byte half of every row, wave-private (half the
// requests per wave, no barrier anywhere in the tap loop), and multiplies all 64 rows of the block by its half: one MFMA per
// (tap, 32-row tile).  At the end of a block the two partial tiles are exchanged through LDS (each wave finishes 32 of the 64
// rows: partial of the partner's tile out, partner's partial of its own tile in: two barriers per block) and each wave runs the
// epilogue of its 32 rows.  The slot tables are shared by the pair (each wave requests half of the next block's table; the
// block-end barrier publishes them), everything else is wave-private, and every s_waitcnt vmcnt in the loop is a counted wait as
// in the one-wave kernel.  Same baked 64-row metadata (spconv_slab_meta.h): an entry is (row + 1) * 64 | swizzle, the 32-byte
// row offset of this layout is (entry >> 6) << 5 — rows of a tap are near-consecutive, so the unswizzled half rows read
// conflict-free.
//
// Summation order: each wave sums its 16 input channels over the 27 taps, then the two halves are added — not the tap-major
// order of the other kernels: equal to fp32 rounding before the single 16-bit rounding, like the one-wave kernel.
#pragma once
#include "spconv_slab_fstat.h"

namespace bevamd {
namespace slab {

template <int CAP>
struct PlanF2 {
  static constexpr int CIN = 32, COUT = 32;
  static constexpr int BM = BAKED_ROWS;             // 64 rows per block = two 32-row MFMA tiles
  static constexpr int HB = 32;                     // staged bytes per row: this wave's 16 input channels
  static constexpr int RPI = 32;                    // rows per DMA instruction (2 lanes x 16 B per row)
  static constexpr int PX = CAP / RPI;              // row requests of a full piece
  static constexpr int XB = (CAP + 1) * HB;         // zero row first, then CAP staged half rows
  static constexpr int NXB = 3;                     // buffer j holds kernel plane j: two pieces of lookahead
  static constexpr int SLB = 27 * BM * 2;           // slot table of a block
  static constexpr int NSL = (SLB + 1023) / 1024;   // its DMA requests (4: two per wave); the last runs 640 bytes into the next table
  static constexpr int SLL = NSL * 1024;
  static constexpr int SRB = COUT * 2 + 16;         // padded row pitch of the epilogue scratch (32 rows per wave)
  static constexpr int XCH = 16 * 64 * 4;           // one partial tile: 16 floats per lane
  static constexpr int NRES = 32 * COUT * 2 / 1024; // residual requests of a wave's 32 rows (2)
  // per wave
  static constexpr int W_X = 0;
  // exchange area (the epilogue scratch aliases it: 32 * SRB <= XCH).  Buffer 2 is idle between the last tap of plane 2 and the first
  // request of the next block's plane 0 — exactly when a block is finished — so with CAP >= 128 the area lives INSIDE buffer 2,
  // behind its zero row (4 pairs per CU instead of 3); a smaller buffer cannot hold it and the area gets its own 4 KiB
  static constexpr bool XCH_IN_X = XB - HB >= XCH;
  static constexpr int W_XCH = XCH_IN_X ? 2 * XB + HB : NXB * XB;
  static constexpr int W_RES = XCH_IN_X ? NXB * XB : W_XCH + XCH;
  static constexpr int W_HDR = W_RES + NRES * 1024;
  static constexpr int WB = W_HDR + 32;
  // per workgroup
  static constexpr int OFF_SLOT = 2 * WB;           // two slot tables: this block's, the next one's
  static constexpr int OFF_CONST = OFF_SLOT + 2 * SLL;   // scale [32] f32, shift [32] f32, bias [32] 16-bit
  static constexpr int BYTES = OFF_CONST + 32 * 4 * 2 + 64;
  static_assert(CAP % RPI == 0, "CAP must be a whole number of DMA instructions");
  static_assert(32 * SRB <= XCH, "the epilogue scratch aliases the exchange area");
  static_assert(PX <= TAPS - 2, "one row request per tap, the last two taps carry the slot / residual requests");
  static_assert(NSL == 4 && NRES == 2, "two taps x one request per wave");
  static_assert(3 * PX + NSL / 2 + NRES + 1 < 60, "vmcnt is a 6-bit counter");
  static_assert(PX + NSL / 2 + 1 <= 15 && PX + NRES <= 15, "the run-time counted wait covers 0..15");
  static_assert(XB % 16 == 0 && WB % 16 == 0 && OFF_SLOT % 16 == 0 && OFF_CONST % 16 == 0, "16-byte aligned regions");
};

template <int DT, int CAP>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
void spconv_slabf2_kernel(SlabArgs sa) {
  typedef PlanF2<CAP> P;
  typedef typename Num<DT>::T T;
  extern __shared__ u32x4 lds[];
  char* const L = (char*)lds;
  lds_char* const L3 = (lds_char*)(void*)lds;
  const Args& a = sa.a;
  const int m = a.m_dev ? (*a.m_dev < a.m_cap ? *a.m_dev : a.m_cap) : a.m_cap;
  const int nblk = (m + P::BM - 1) / P::BM;
  // XCD x owns the contiguous block range [x*per, (x+1)*per); its gx wave pairs walk it round-robin
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int per = (nblk + 7) >> 3;
  const int blk_end = (xcd + 1) * per < nblk ? (xcd + 1) * per : nblk;
  int blk = xcd * per + bix;
  if (blk >= blk_end) return;   // the pair leaves together
  int blk_n = blk + gx;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // input-channel half of this wave
  const unsigned lane16 = (unsigned)lane * 16u;
  const int r32 = lane & 31, h = lane >> 5;   // MFMA operand layout: row (or output channel) of the 32-tile, 8-channel group of the 16
  const int WO = w * P::WB, PO = (w ^ 1) * P::WB;   // this wave's region, the partner's

  const unsigned row_bytes = (unsigned)a.feat_stride * 2u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.feat, 0, (unsigned)a.n_in * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wimg, 0, sa.wimg_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)sa.slots, 0, sa.slot_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)sa.hdr, 0, (unsigned)((a.m_cap + P::BM - 1) / P::BM) * (unsigned)(PLANES * 8), 0x00020000);
  const unsigned res_pitch = (unsigned)a.res_stride * 2u;
  const __amdgpu_buffer_rsrc_t rs_r =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual ? a.residual : a.feat), 0, a.residual ? (unsigned)a.m_cap * res_pitch : 0u, 0x00020000);

  // ---- once per kernel: this wave's half of the filter, its zero rows, the per-channel epilogue operands -------------------
  // A fragment (tap k) of lane (r32 = output channel, h): W[k][16 w + 8 h .. + 7][r32] = the 16 bytes lane (r32 % 16) + 16 (2 w + h)
  // holds in the 16x16x32 filter image's fragment (k, output tile r32 / 16)
  u32x4 wf[27];
#pragma unroll
  for (int k = 0; k < 27; ++k)
    wf[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (unsigned)((k * 2 + (r32 >> 4)) * 1024 + ((r32 & 15) + 16 * (2 * w + h)) * 16), 0u, 0);
  if (lane < P::NXB * 2) {
    const int b = lane >> 1, p = lane & 1;
    *(u32x4*)(L + WO + P::W_X + b * P::XB + p * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  if (w == 0 && lane < 32) {
    ((float*)(L + P::OFF_CONST))[lane] = a.scale ? a.scale[lane] : 1.f;
    ((float*)(L + P::OFF_CONST))[32 + lane] = a.scale ? a.shift[lane] : 0.f;
    ((T*)(L + P::OFF_CONST + 256))[lane] = a.bias ? ((const T*)a.bias)[lane] : Num<DT>::from_f32(0.f);
  }

  // block headers (scalar): (first row, row count | HDR_RAW) of the three planes, current and next block
  struct Hdr { int lo[PLANES], cnt[PLANES]; };
  auto load_hdr = [&](int b, bool ok) {
    Hdr hd;
    const int2* hp = sa.hdr + (size_t)(ok ? b : 0) * PLANES;
#pragma unroll
    for (int j = 0; j < PLANES; ++j) {
      const int2 v = hp[j];
      hd.lo[j] = __builtin_amdgcn_readfirstlane(v.x);
      hd.cnt[j] = ok ? __builtin_amdgcn_readfirstlane(v.y) : 0;
    }
    return hd;
  };
  Hdr hc = load_hdr(blk, true);
  Hdr hn = load_hdr(blk_n, blk_n < blk_end);
  auto issue_hdr = [&](bool go, int b) {
    if (go && lane < 2) dma16_l(rs_h, lane16, (unsigned)b * (unsigned)(PLANES * 8), L3 + (WO + P::W_HDR));
  };
  auto read_hdr = [&](bool ok) {
    Hdr hd;
#pragma unroll
    for (int j = 0; j < PLANES; ++j) {
      const int2 v = *(const int2*)(L + WO + P::W_HDR + j * 8);
      hd.lo[j] = __builtin_amdgcn_readfirstlane(v.x);
      hd.cnt[j] = ok ? __builtin_amdgcn_readfirstlane(v.y) : 0;
    }
    return hd;
  };

  // ---- requests -------------------------------------------------------------------------------------------------------
  // Row request i of a piece: this wave's halves of source rows [lo + q*CAP + 32 i, + 32) -> LDS rows 1 + 32 i .. of buffer `buf`:
  // lane l -> row l / 2, 16-byte piece l % 2 of the half; the DMA writes lanes linearly, which IS that layout.
  const unsigned lane_row_off = (unsigned)(lane >> 1) * row_bytes + (unsigned)w * 32u + (unsigned)(lane & 1) * 16u;
  struct RowReq { int n; unsigned soff; lds_char* dst; };
  auto row_req = [&](bool go, const Hdr& hd, int j, int q) {
    const int cnt = (int)((unsigned)hd.cnt[j] & ~HDR_RAW);
    int n = go ? cnt - q * CAP : 0;
    n = n < 0 ? 0 : (n < CAP ? n : CAP);
    RowReq rq;
    rq.n = (n + P::RPI - 1) / P::RPI;   // the last request may run past the range: rows no slot refers to (past the tensor: zeros)
    rq.soff = (unsigned)(hd.lo[j] + q * CAP) * row_bytes;
    rq.dst = L3 + (WO + P::W_X + j * P::XB + P::HB);
    return rq;
  };
  auto issue_row = [&](const RowReq& rq, int i) {
    if (i < rq.n) dma16_l(rs_x, lane_row_off, rq.soff + (unsigned)(i * P::RPI) * row_bytes, rq.dst + i * 1024);
  };
  // slot table of block b -> slot buffer sb: request i of NSL; wave w issues requests 2 w and 2 w + 1
  auto issue_slots = [&](bool go, int b, int sb, int i) {
    if (go) dma16_l(rs_s, lane16, (unsigned)b * (unsigned)P::SLB + (unsigned)(i * 1024), L3 + (P::OFF_SLOT + sb * P::SLL + i * 1024));
  };
  // residual pieces of this wave's 32 rows of block b, pass i: lane -> row 16 i + lane / 4, 16-byte piece lane % 4
  const int j4 = lane & 3, rsub = lane >> 2, col0 = j4 * 8;
  const unsigned lane_res_off = (unsigned)rsub * res_pitch + (unsigned)j4 * 16u;
  const bool has_res = a.residual != nullptr;
  auto issue_residual = [&](int b, int i) {
    if (has_res) dma16_l(rs_r, lane_res_off, (unsigned)(b * P::BM + w * 32 + i * 16) * res_pitch, L3 + (WO + P::W_RES + i * 1024));
  };

  f32x16 acc[2];   // acc[0]: the tile this wave FINISHES (rows 32 w ..), acc[1]: the partner's
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // ---- the nine taps of plane J --------------------------------------------------------------------------------------------
  auto plane = [&](auto jc, auto fastc, const Hdr& hd, int sb, auto&& tapwork) {
    constexpr int J = decltype(jc)::value;
    constexpr bool FAST = decltype(fastc)::value;
    const char* X = L + WO + P::W_X + J * P::XB + h * 16;   // this lane's piece of LDS row 0 of buffer J
    const uint16_t* sl = (const uint16_t*)(L + P::OFF_SLOT + sb * P::SLL) + J * TAPS * P::BM + r32;
    const uint16_t* sl0 = sl + w * 32;           // slots of the tile this wave finishes
    const uint16_t* sl1 = sl + (w ^ 1) * 32;     // ... of the partner's
    const int cnt = (int)((unsigned)hd.cnt[J] & ~HDR_RAW);
    const bool rawslots = ((unsigned)hd.cnt[J] & HDR_RAW) != 0;
    const int pieces = FAST ? 1 : (cnt + CAP - 1) / CAP;
    for (int q = 0; q < pieces; ++q) {
      if (!FAST && q > 0) {   // the next piece of a long range, synchronously
        const RowReq rq = row_req(true, hd, J, q);
#pragma unroll
        for (int i = 0; i < P::PX; ++i) issue_row(rq, i);
        wait_dma<0>();
      }
      unsigned raw[4][2];
      u32x4 xa[3][2];   // [tap % 3][tile]: fragments are requested TWO taps ahead, slots three
      auto load_slots = [&](int d) {
        raw[d % 4][0] = (unsigned)sl0[d * P::BM];
        raw[d % 4][1] = (unsigned)sl1[d * P::BM];
      };
      auto fetch = [&](int d, int t) {
        unsigned row;   // LDS row of the neighbour: 0 = the zero row
        if constexpr (FAST) {
          row = raw[d % 4][t] >> 6;
        } else {
          const unsigned s = rawslots ? raw[d % 4][t] : (raw[d % 4][t] >> 6) - 1u;   // baked 0 -> 0xFFFFFFFF: outside any piece
          const unsigned pbase = (unsigned)(q * CAP), plive = (unsigned)cnt - pbase;
          const unsigned prow = plive < (unsigned)CAP ? plive : (unsigned)CAP;
          const unsigned e = s - pbase;
          row = e < prow ? e + 1u : 0u;
        }
        xa[d % 3][t] = *(const u32x4*)(X + (row << 5));   // v_lshl_add_u32 + the buffer's offset as the instruction's immediate
      };
      load_slots(0);
      load_slots(1);
      load_slots(2);
      fetch(0, 0);
      fetch(0, 1);
      fetch(1, 0);
      fetch(1, 1);
#pragma unroll
      for (int d = 0; d < TAPS; ++d) {
        const int k = J * TAPS + d;
        acc[0] = mfma32<DT>(wf[k], xa[d % 3][0], acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (d + 3 < TAPS) load_slots(d + 3);
        if (FAST || q == pieces - 1) tapwork(d);
        __builtin_amdgcn_sched_barrier(0);
        acc[1] = mfma32<DT>(wf[k], xa[d % 3][1], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        if (d + 2 < TAPS) {
          fetch(d + 2, 0);
          fetch(d + 2, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  auto run_plane = [&](auto jc, const Hdr& hd, int sb, auto&& tapwork) {
    constexpr int J = decltype(jc)::value;
    const unsigned hv = (unsigned)hd.cnt[J];
    const int cnt = (int)(hv & ~HDR_RAW);
    if (cnt == 0) {   // nothing to multiply, but the requests of the pieces ahead still go out
#pragma unroll
      for (int d = 0; d < TAPS; ++d) tapwork(d);
      return;
    }
    if (!(hv & HDR_RAW) && cnt <= CAP) plane(jc, std::true_type{}, hd, sb, tapwork);
    else plane(jc, std::false_type{}, hd, sb, tapwork);
  };

  // ---- end of a block: exchange the partial tiles, then the epilogue of this wave's 32 rows ---------------------------------
  char* const scr = L + WO + P::W_XCH;
  const unsigned xch_mine = (unsigned)(uintptr_t)(L3 + (WO + P::W_XCH)) + lane16;
  const unsigned xch_partner = (unsigned)(uintptr_t)(L3 + (PO + P::W_XCH)) + lane16;
  auto finish_block = [&](int b) {
    // the partner's tile, as this wave's partial of it: lane-linear, conflict-free; inline asm — hipcc would put vmcnt(0) in front of
    // an LDS store it can see while LDS-DMA requests are in flight (the exchange area is nobody's DMA target)
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      const u32x4 v = {__float_as_uint(acc[1][qd * 4 + 0]), __float_as_uint(acc[1][qd * 4 + 1]), __float_as_uint(acc[1][qd * 4 + 2]),
                       __float_as_uint(acc[1][qd * 4 + 3])};
      asm volatile("ds_write_b128 %0, %1" ::"v"(xch_mine + (unsigned)(qd * 1024)), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // both partials are in LDS; also: the partner's half of the next block's slot table has landed
    {
      u32x4 p[4];
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) asm volatile("ds_read_b128 %0, %1" : "=v"(p[qd]) : "v"(xch_partner + (unsigned)(qd * 1024)) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[0][qd * 4 + e] += __uint_as_float(p[qd][e]);
    }
    __builtin_amdgcn_s_barrier();   // the partner has read this wave's area: the epilogue scratch may overwrite it
    // D[channel i][row j]: lane (j = r32, h) holds channels 8 (e/4) + 4 h + e%4, e = 0..15, of row r32 of this wave's tile
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) {
      T p[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) p[e] = Num<DT>::from_f32(acc[0][qd * 4 + e]);
      *(uint2*)(scr + r32 * P::SRB + (8 * qd + 4 * h) * 2) = *(const uint2*)p;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const float* cs = (const float*)(L + P::OFF_CONST);
    float sv[8], hv[8];
    {
      const f32x4 s0 = *(const f32x4*)(cs + col0), s1 = *(const f32x4*)(cs + col0 + 4);
      const f32x4 h0 = *(const f32x4*)(cs + 32 + col0), h1 = *(const f32x4*)(cs + 32 + col0 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { sv[e] = s0[e]; sv[4 + e] = s1[e]; hv[e] = h0[e]; hv[4 + e] = h1[e]; }
    }
    const u32x4 bias16 = *(const u32x4*)(L + P::OFF_CONST + 256 + col0 * 2);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = pass * 16 + rsub;
      const int row = b * P::BM + w * 32 + r;
      const u32x4 rawv = *(const u32x4*)(scr + r * P::SRB + j4 * 16);
      const u32x4 resv = *(const u32x4*)(L + WO + P::W_RES + pass * 1024 + lane * 16);
      if (row < m) {
        const u32x4 o = finish8<DT>(rawv, a.bias != nullptr, bias16, a.scale != nullptr, sv, hv, has_res, resv, a.relu != 0);
        *(u32x4*)((T*)a.out + (size_t)row * a.out_stride + col0) = o;
      }
    }
  };

  // ---- pipeline (per wave, as in the one-wave kernel: counted waits on run-time counts) -------------------------------------
  auto wait_pending = [&](int n) {
    switch (n < 15 ? n : 15) {
#define BEVAMD_W(N) case N: wait_dma<N>(); break;
      BEVAMD_W(0) BEVAMD_W(1) BEVAMD_W(2) BEVAMD_W(3) BEVAMD_W(4) BEVAMD_W(5) BEVAMD_W(6) BEVAMD_W(7)
      BEVAMD_W(8) BEVAMD_W(9) BEVAMD_W(10) BEVAMD_W(11) BEVAMD_W(12) BEVAMD_W(13) BEVAMD_W(14) BEVAMD_W(15)
#undef BEVAMD_W
    }
  };
  RowReq r_a = row_req(true, hc, 0, 0), r_b = row_req(true, hc, 1, 0);   // the two pieces in flight: a = next to be read, b = the one after
  {
#pragma unroll
    for (int i = 0; i < P::PX; ++i) issue_row(r_a, i);
    issue_slots(true, blk, 0, 2 * w);
    issue_slots(true, blk, 0, 2 * w + 1);
#pragma unroll
    for (int i = 0; i < P::PX; ++i) issue_row(r_b, i);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), through the builtin: hipcc then knows the filter has arrived and puts no waits for it into the loop
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // zero rows and epilogue operands are in LDS
    __builtin_amdgcn_s_barrier();         // ... and so are the partner's half of the slot table and wave 0's epilogue operands
  }
  int sb = 0;
  int extra_prev = 0;   // slot / residual requests of the previous piece (issued after its row requests)
  for (;;) {
    const bool has_n = blk_n < blk_end;
    const bool has_nn = blk_n + gx < blk_end;
    const int n_slots = (has_n ? P::NSL / 2 : 0) + (has_nn ? 1 : 0), n_res = has_res ? P::NRES : 0;
    {   // plane 0 reads r_a
      const RowReq rq = row_req(true, hc, 2, 0);
      wait_pending(r_b.n + extra_prev);
      run_plane(std::integral_constant<int, 0>{}, hc, sb, [&](int d) {
        if (d < P::PX) issue_row(rq, d);
        if (d >= TAPS - 2) issue_slots(has_n, blk_n, sb ^ 1, 2 * w + (d - (TAPS - 2)));
        if (d == TAPS - 1) issue_hdr(has_nn, blk_n + gx);
      });
      r_a = r_b; r_b = rq; extra_prev = n_slots;
    }
    {   // plane 1
      const RowReq rq = row_req(has_n, hn, 0, 0);
      wait_pending(r_b.n + extra_prev);
      run_plane(std::integral_constant<int, 1>{}, hc, sb, [&](int d) {
        if (d < P::PX) issue_row(rq, d);
        if (d >= TAPS - 2) issue_residual(blk, d - (TAPS - 2));
      });
      r_a = r_b; r_b = rq; extra_prev = n_res;
    }
    {   // plane 2
      const RowReq rq = row_req(has_n, hn, 1, 0);
      wait_pending(r_b.n + extra_prev);
      run_plane(std::integral_constant<int, 2>{}, hc, sb, [&](int d) {
        if (d < P::PX) issue_row(rq, d);
      });
      r_a = r_b; r_b = rq; extra_prev = 0;
    }
    wait_pending(r_b.n);   // the residual pieces (requested before this plane's row requests) and this wave's half of the next slot table
    finish_block(blk);
    if (!has_n) break;
    blk = blk_n;
    blk_n += gx;
    sb ^= 1;
    hc = hn;
    hn = read_hdr(has_nn);   // landed: requested a block ago, older than everything the waits of this block have covered
  }
  wait_dma<0>();
}

}  // namespace slab
}  // namespace bevamd
