// Staged-rows submanifold convolution, sixth cut: filter-stationary WAVE OCTETS (64 -> 64 channels).
//
// The register-filter kernel of the 64-channel layers (spconv_slab_regw.h) re-reads its filter fragments for every block: 4 KiB per
// wave and tap, 432 KiB per 128-row block and workgroup through the vector-memory pipe against 64 KiB of staged rows — the
// "operand part" that adds to the matrix part instead of hiding under it (EXPERIMENTS D.9: 85 + 95 us).  Here the filter never moves:
// the 27 x 64 x 64 filter (221 KB) is spread over the EIGHT waves of a workgroup, one workgroup per compute unit, two waves per SIMD:
// wave (ci, co) keeps W[k][16 ci .. 16 ci + 15][32 co .. 32 co + 31] as 27 A fragments of v_mfma_f32_32x32x16 (108 registers — the
// same register picture as the 32-channel wave pairs of spconv_slab_fstat2.h), stages only ITS 32-byte quarter of every row,
// wave-private (no barrier anywhere in the tap loop; the two waves with one ci stage the same quarter twice), and multiplies both
// 32-row tiles of a 64-row block by its slice: one MFMA per (tap, tile).  At the end of a block the four input-channel partials of
// every output element meet: in the MFMA's result layout register group q of a lane holds output channels 8 q + 4 h .. + 3 of the
// wave's 32, so wave ci FINISHES group ci — channels 32 co + 8 ci .. + 7 of all 64 rows — sends its three other groups to their
// owners through LDS (6 KiB out, 6 KiB in per wave and block: the outgoing tiles lie in the wave's idle third row buffer) and adds
// the four partials in ascending ci order.  Two workgroup barriers per block, as in the pair kernel.  Each lane then owns four
// consecutive channels of two rows: epilogue in registers, 8-byte stores (the eight waves of the workgroup cover a 128-byte row
// between them).  Same baked 64-row metadata as the other filter-stationary kernels (spconv_slab_meta.h).
//
// Summation order: 16 input channels x 27 taps per wave, then the four quarters in ascending order — equal to the other kernels
// to fp32 rounding before the single 16-bit rounding, not bit for bit.
#pragma once
#include "spconv_slab_fstat2.h"

namespace bevamd {
namespace slab {

template <int CAP>
struct PlanF4 {
  static constexpr int CIN = 64, COUT = 64, NWAVE = 8;
  static constexpr int BM = BAKED_ROWS;             // 64 rows per block = two 32-row MFMA tiles
  static constexpr int HB = 32;                     // staged bytes per row: this wave's 16 input channels
  static constexpr int RPI = 32;                    // rows per DMA instruction (2 lanes x 16 B per row)
  static constexpr int PX = CAP / RPI;              // row requests of a full piece
  static constexpr int XB = (CAP + 1) * HB;         // zero row first, then CAP staged quarter rows
  static constexpr int NXB = 3;                     // buffer j holds kernel plane j: two pieces of lookahead
  static constexpr int SLB = 27 * BM * 2;           // slot table of a block
  static constexpr int NSL = (SLB + 1023) / 1024;   // its DMA requests (4: waves 0-3 issue one each)
  static constexpr int SLL = NSL * 1024;
  // per wave
  static constexpr int W_X = 0;
  // outgoing partial tiles [receiver slot 0..2][tile 0..1] x 1 KiB: the first four inside row buffer 2 behind its zero row (idle
  // between the last tap of plane 2 and the first request of the next block's plane 2), the last two behind the buffers
  static constexpr int W_OUT_A = 2 * XB + HB;
  static constexpr int W_OUT_B = NXB * XB;
  static constexpr int W_RES = W_OUT_B + 2 * 1024;  // residual pieces of the 64 rows: 16 bytes (this wave's 8 output channels) per row
  static constexpr int W_HDR = W_RES + 1024;
  static constexpr int WB = W_HDR + 32;
  // per workgroup
  static constexpr int OFF_SLOT = NWAVE * WB;            // two slot tables: this block's, the next one's
  static constexpr int OFF_CONST = OFF_SLOT + 2 * SLL;   // scale [64] f32, shift [64] f32, bias [64] 16-bit
  static constexpr int BYTES = OFF_CONST + 64 * 4 * 2 + 64 * 2 + 64;
  static_assert(CAP % RPI == 0, "CAP must be a whole number of DMA instructions");
  static_assert(XB - HB >= 4 * 1024, "four outgoing tiles live in row buffer 2");
  static_assert(PX <= TAPS - 1, "one row request per tap, the last tap carries the slot / header / residual requests");
  static_assert(NSL == 4, "one request each for waves 0-3");
  static_assert(3 * PX + 3 < 60, "vmcnt is a 6-bit counter");
  static_assert(PX + 2 <= 15, "the run-time counted wait covers 0..15");
  static_assert(XB % 16 == 0 && WB % 16 == 0 && OFF_SLOT % 16 == 0 && OFF_CONST % 16 == 0, "16-byte aligned regions");
  static_assert(BYTES <= 160 * 1024, "one workgroup per compute unit");
};

template <int DT, int CAP>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void spconv_slabf4_kernel(SlabArgs sa) {
  typedef PlanF4<CAP> P;
  typedef typename Num<DT>::T T;
  extern __shared__ u32x4 lds[];
  char* const L = (char*)lds;
  lds_char* const L3 = (lds_char*)(void*)lds;
  const Args& a = sa.a;
  const int m = a.m_dev ? (*a.m_dev < a.m_cap ? *a.m_dev : a.m_cap) : a.m_cap;
  const int nblk = (m + P::BM - 1) / P::BM;
  // XCD x owns the contiguous block range [x*per, (x+1)*per); its gx workgroups walk it round-robin
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int per = (nblk + 7) >> 3;
  const int blk_end = (xcd + 1) * per < nblk ? (xcd + 1) * per : nblk;
  int blk = xcd * per + bix;
  if (blk >= blk_end) return;   // the eight waves leave together
  int blk_n = blk + gx;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int ci = w & 3, co = w >> 2;   // input-channel quarter, output-channel half
  const unsigned lane16 = (unsigned)lane * 16u;
  const int r32 = lane & 31, h = lane >> 5;   // MFMA operand layout: row (or output channel) of the 32-tile, 8-channel group of the 16
  const int WO = w * P::WB;

  const unsigned row_bytes = (unsigned)a.feat_stride * 2u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.feat, 0, (unsigned)a.n_in * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wimg, 0, sa.wimg_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)sa.slots, 0, sa.slot_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)sa.hdr, 0, (unsigned)((a.m_cap + P::BM - 1) / P::BM) * (unsigned)(PLANES * 8), 0x00020000);
  const unsigned res_pitch = (unsigned)a.res_stride * 2u;
  const __amdgpu_buffer_rsrc_t rs_r =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual ? a.residual : a.feat), 0, a.residual ? (unsigned)a.m_cap * res_pitch : 0u, 0x00020000);

  // ---- once per kernel: this wave's slice of the filter, its zero rows, the per-channel epilogue operands --------------------
  // A fragment (tap k) of lane (r32 = output channel of the wave's 32, h): W[k][16 ci + 8 h .. + 7][32 co + r32] = the 16 bytes lane
  // (r32 % 16) + 16 (2 (ci % 2) + h) holds in the 16x16x32 filter image's fragment (k, chunk ci / 2, output tile 2 co + r32 / 16)
  u32x4 wf[27];
#pragma unroll
  for (int k = 0; k < 27; ++k)
    wf[k] = __builtin_amdgcn_raw_buffer_load_b128(
        rs_w, (unsigned)((((k * 2 + (ci >> 1)) * 4 + 2 * co + (r32 >> 4)) * 1024) + ((r32 & 15) + 16 * (2 * (ci & 1) + h)) * 16), 0u, 0);
  if (lane < P::NXB * 2) {
    const int b = lane >> 1, p = lane & 1;
    *(u32x4*)(L + WO + P::W_X + b * P::XB + p * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  if (w == 0) {
    ((float*)(L + P::OFF_CONST))[lane] = a.scale ? a.scale[lane] : 1.f;
    ((float*)(L + P::OFF_CONST))[64 + lane] = a.scale ? a.shift[lane] : 0.f;
    ((T*)(L + P::OFF_CONST + 512))[lane] = a.bias ? ((const T*)a.bias)[lane] : Num<DT>::from_f32(0.f);
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
  // Row request i of a piece: this wave's quarters of source rows [lo + q*CAP + 32 i, + 32) -> LDS rows 1 + 32 i .. of buffer j:
  // lane l -> row l / 2, 16-byte piece l % 2 of the quarter; the DMA writes lanes linearly, which IS that layout.
  const unsigned lane_row_off = (unsigned)(lane >> 1) * row_bytes + (unsigned)ci * 32u + (unsigned)(lane & 1) * 16u;
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
  // slot table of block b -> slot buffer sb: request i of NSL, issued by wave i
  auto issue_slots = [&](bool go, int b, int sb) {
    if (go && w < P::NSL) dma16_l(rs_s, lane16, (unsigned)b * (unsigned)P::SLB + (unsigned)(w * 1024), L3 + (P::OFF_SLOT + sb * P::SLL + w * 1024));
  };
  // residual pieces of this wave's 8 output channels of the block's 64 rows: lane -> row
  const unsigned lane_res_off = (unsigned)lane * res_pitch + (unsigned)(32 * co + 8 * ci) * 2u;
  const bool has_res = a.residual != nullptr;
  auto issue_residual = [&](int b) {
    if (has_res) dma16_l(rs_r, lane_res_off, (unsigned)(b * P::BM) * res_pitch, L3 + (WO + P::W_RES));
  };

  f32x16 acc[2];   // the two 32-row tiles of the block x this wave's 32 output channels, over its 16 input channels
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // ---- the nine taps of plane J --------------------------------------------------------------------------------------------
  auto plane = [&](auto jc, auto fastc, const Hdr& hd, int sb, auto&& tapwork) {
    constexpr int J = decltype(jc)::value;
    constexpr bool FAST = decltype(fastc)::value;
    const char* X = L + WO + P::W_X + J * P::XB + h * 16;   // this lane's piece of LDS row 0 of buffer J
    const uint16_t* sl0 = (const uint16_t*)(L + P::OFF_SLOT + sb * P::SLL) + J * TAPS * P::BM + r32;
    const uint16_t* sl1 = sl0 + 32;
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
        xa[d % 3][t] = *(const u32x4*)(X + (row << 5));
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

  // ---- end of a block: the four input-channel partials meet, then the epilogue of this wave's 8 channels x 64 rows --------------
  // outgoing tile i = 2 * (receiver slot) + tile of a wave: LDS offset inside its region
  auto out_off = [&](int i) { return i < 4 ? P::W_OUT_A + i * 1024 : P::W_OUT_B + (i - 4) * 1024; };
  const unsigned lds_base = (unsigned)(uintptr_t)L3;
  auto finish_block = [&](int b) {
    // partials of the register groups this wave does not own -> its outgoing area (lane-linear, conflict-free); inline asm: hipcc
    // would put vmcnt(0) in front of an LDS store it can see while LDS-DMA requests are in flight (nobody's DMA target right now)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (r != ci) {   // wave-uniform
        const int s = r < ci ? r : r - 1;   // receiver r's slot among this wave's three receivers
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const u32x4 v = {__float_as_uint(acc[t][r * 4 + 0]), __float_as_uint(acc[t][r * 4 + 1]), __float_as_uint(acc[t][r * 4 + 2]),
                           __float_as_uint(acc[t][r * 4 + 3])};
          asm volatile("ds_write_b128 %0, %1" ::"v"(lds_base + (unsigned)(WO + out_off(s * 2 + t)) + lane16), "v"(v) : "memory");
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every partial is in LDS; also: the next block's slot table has landed
    u32x4 p[3][2];   // [sender in ascending ci order, this wave left out][tile]
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int s = j < ci ? j : j + 1;          // the sender
      const int slot = ci < s ? ci : ci - 1;     // this wave's slot among ITS receivers
#pragma unroll
      for (int t = 0; t < 2; ++t)
        asm volatile("ds_read_b128 %0, %1" : "=v"(p[j][t]) : "v"(lds_base + (unsigned)((co * 4 + s) * P::WB + out_off(slot * 2 + t)) + lane16) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every outgoing area has been read: row buffer 2 may take the next block's rows
    const float* cs = (const float*)(L + P::OFF_CONST);
    const int c0 = 32 * co + 8 * ci + 4 * h;   // this lane's four consecutive output channels
    const f32x4 sv = *(const f32x4*)(cs + c0), hv = *(const f32x4*)(cs + 64 + c0);
    const uint2 bias16 = *(const uint2*)(L + P::OFF_CONST + 512 + c0 * 2);
    auto tail = [&](auto cic) {
      constexpr int CI = decltype(cic)::value;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float f[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float s = 0.f;
#pragma unroll
          for (int c = 0; c < 4; ++c) {   // the four quarters in ascending order, whoever owns the element
            const float pc = c == CI ? acc[t][CI * 4 + e] : __uint_as_float(p[c < CI ? c : c - 1][t][e]);
            s = c == 0 ? pc : s + pc;
          }
          f[e] = s;
        }
        T q16[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) q16[e] = Num<DT>::from_f32(f[e]);
        const uint2 rawv = *(const uint2*)q16;
        const int r = t * 32 + r32, row = b * P::BM + r;
        const uint2 resv = *(const uint2*)(L + WO + P::W_RES + r * 16 + h * 8);
        if (row < m) {
          uint2 o;
          o.x = finish_pair<DT>(rawv.x, a.bias != nullptr, bias16.x, a.scale != nullptr, sv[0], sv[1], hv[0], hv[1], has_res, resv.x, a.relu != 0);
          o.y = finish_pair<DT>(rawv.y, a.bias != nullptr, bias16.y, a.scale != nullptr, sv[2], sv[3], hv[2], hv[3], has_res, resv.y, a.relu != 0);
          *(uint2*)((T*)a.out + (size_t)row * a.out_stride + c0) = o;
        }
      }
    };
    switch (ci) {   // the owned register group is a compile-time index
      case 0: tail(std::integral_constant<int, 0>{}); break;
      case 1: tail(std::integral_constant<int, 1>{}); break;
      case 2: tail(std::integral_constant<int, 2>{}); break;
      default: tail(std::integral_constant<int, 3>{}); break;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  };

  // ---- pipeline (per wave, as in the pair kernel: counted waits on run-time counts) -----------------------------------------
  auto wait_pending = [&](int n) {
    switch (n < 15 ? n : 15) {
#define BEVAMD_W(N) case N: wait_dma<N>(); break;
      BEVAMD_W(0) BEVAMD_W(1) BEVAMD_W(2) BEVAMD_W(3) BEVAMD_W(4) BEVAMD_W(5) BEVAMD_W(6) BEVAMD_W(7)
      BEVAMD_W(8) BEVAMD_W(9) BEVAMD_W(10) BEVAMD_W(11) BEVAMD_W(12) BEVAMD_W(13) BEVAMD_W(14) BEVAMD_W(15)
#undef BEVAMD_W
    }
  };
  const int slot_req = w < P::NSL ? 1 : 0;   // this wave's share of a slot table
  RowReq r_a = row_req(true, hc, 0, 0), r_b = row_req(true, hc, 1, 0);   // the two pieces in flight: a = next to be read, b = the one after
  {
#pragma unroll
    for (int i = 0; i < P::PX; ++i) issue_row(r_a, i);
    issue_slots(true, blk, 0);
#pragma unroll
    for (int i = 0; i < P::PX; ++i) issue_row(r_b, i);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), through the builtin: hipcc then knows the filter has arrived and puts no waits for it into the loop
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // zero rows and epilogue operands are in LDS
    __builtin_amdgcn_s_barrier();         // ... and so are the other waves' shares of the slot table and wave 0's epilogue operands
  }
  int sb = 0;
  int extra_prev = 0;   // slot / header / residual requests of the previous piece (issued after its row requests)
  for (;;) {
    const bool has_n = blk_n < blk_end;
    const bool has_nn = blk_n + gx < blk_end;
    const int n_slots = (has_n ? slot_req : 0) + (has_nn ? 1 : 0), n_res = has_res ? 1 : 0;
    {   // plane 0 reads r_a
      const RowReq rq = row_req(true, hc, 2, 0);
      wait_pending(r_b.n + extra_prev);
      run_plane(std::integral_constant<int, 0>{}, hc, sb, [&](int d) {
        if (d < P::PX) issue_row(rq, d);
        if (d == TAPS - 1) {
          issue_slots(has_n, blk_n, sb ^ 1);
          issue_hdr(has_nn, blk_n + gx);
        }
      });
      r_a = r_b; r_b = rq; extra_prev = n_slots;
    }
    {   // plane 1
      const RowReq rq = row_req(has_n, hn, 0, 0);
      wait_pending(r_b.n + extra_prev);
      run_plane(std::integral_constant<int, 1>{}, hc, sb, [&](int d) {
        if (d < P::PX) issue_row(rq, d);
        if (d == TAPS - 1) issue_residual(blk);
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
    wait_pending(r_b.n);   // the residual pieces (requested before this plane's row requests) and this wave's share of the next slot table
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
