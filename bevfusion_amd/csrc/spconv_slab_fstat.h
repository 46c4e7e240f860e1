// Staged-rows submanifold convolution, fourth cut: FILTER-STATIONARY waves (32 -> 32 channels).
//
// What bounds the register-ring kernels (spconv_slab_regw.h / _persist.h) on the 32-channel layers, measured on experiment
// builds (tools/time_slab_variant.py, 8 frames, 2.08 M rows, 191 us; the MFMA work is 46 us): slot -> LDS address arithmetic
// 23 us, the per-tap filter loads 40 us, row + slot staging 60 us, the epilogue 50 us; the ds_read_b128 fragment reads and the
// MFMAs themselves are free (compiling either out changes nothing).  Counters (profiles/r03_slab32_pmc.txt, r03_slab32_experiments.txt): a wave issues
// 9 instructions per MFMA and a SIMD can start one instruction of a wave every ~4 cycles, so with 2 waves per SIMD the
// 16-cycle MFMA can be at most ~1/3 busy — the kernels are ISSUE-bound, and 2/3 of what goes through the vector-memory pipe is
// the filter (27 taps x 2 KiB x 4 waves per 256 rows).
//
// Here the filter never moves and an MFMA is four times the work.  A workgroup is ONE wave; it keeps the whole 27 x 32 x 32
// filter in 216 registers (one wave per SIMD, 512-register budget) as A-operand fragments of v_mfma_f32_32x32x16, and walks
// 64-row blocks of its XCD's contiguous range: per tap and 64 rows 2 slot reads, 4 address VALU, 4 ds_read_b128 and 4 MFMAs of
// 32 cycles — 3.5 instructions per 32 MFMA-cycles instead of 9 per 16.  The slots arrive BAKED (spconv_slab_meta.h: the LDS
// byte offset of the staged row with the bank swizzle folded in, 0 = the zero row): the address of a fragment is one v_xor.
// Rows are staged per (block, kernel plane) piece into a ring of three wave-private buffers by LDS-DMA, two pieces ahead, ONE
// request per tap slot so that no MFMA group waits behind a burst of requests; the slot table of the next block and the
// residual rows of this one come the same way.  Nothing is shared between waves: no barrier anywhere, every s_waitcnt vmcnt is
// one of the counted waits below (block headers are scalar loads).  The epilogue packs the tile through wave-private LDS and
// finishes whole 64-byte rows (16-byte residual pieces, 16-byte stores).
//
// A plane whose range does not fit the staging buffer (CAP rows) or whose slots are not baked (HDR_RAW) takes the general
// path: the same taps with explicit slot arithmetic, the range staged piece by piece synchronously (rare; correctness, not
// speed).
//
// Same summation order (kernel offset ascending, then the two 16-channel halves of the 32 input channels inside one fp32
// accumulator chain) as the 16x16x32 kernels?  No: v_mfma_f32_32x32x16 reduces 16 channels per instruction, the 16x16x32
// kernels 32 — the fp32 sums associate differently, so results agree with the other kernels to fp32 rounding before the single
// 16-bit rounding (tests: <= 1 ulp of the 16-bit result in a handful of elements), not bit for bit.
#pragma once
#include "spconv_slab.h"

#ifndef BEVAMD_SLABF_EXP
#define BEVAMD_SLABF_EXP 0   // experiment builds (tools/exp_build.sh; wrong results by design): 1 no row requests, 2 no fragment reads after
                             // a plane's first two taps, 4 one MFMA per tap instead of four, 8 no epilogue, 16 no slot / residual requests
#endif

namespace bevamd {
namespace slab {

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int DT>
__device__ __forceinline__ f32x16 mfma32(const u32x4& w, const u32x4& x, f32x16 acc) {
  if constexpr (DT == T_F16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
}

// LDS-DMA with the destination as an address-space-3 pointer: built from the kernel's one LDS base + integer offsets, there is
// no generic -> local cast per request (hipcc guards each such cast with a null check: 8 scalar instructions)
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ void dma16_l(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, lds_char* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, (int)voff, (int)soff, 0, 0);
}

template <int CAP>
struct PlanF {
  static constexpr int CIN = 32, COUT = 32;
  static constexpr int BM = BAKED_ROWS;             // 64 rows per block = two 32-row MFMA tiles
  static constexpr int RB = BAKED_ROW_BYTES;        // staged bytes per row
  static constexpr int PPR = RB / 16;               // 16-byte pieces per row
  static constexpr int RPI = 64 / PPR;              // rows per DMA instruction (1 KiB)
  static constexpr int PX = CAP / RPI;              // row requests of a full piece (a piece issues exactly ceil(rows / 16))
  static constexpr int XB = (CAP + 1) * RB;         // zero row first, then CAP staged rows
  static constexpr int NXB = 3;                     // buffer j holds kernel plane j: two pieces of lookahead
  static constexpr int SLB = 27 * BM * 2;           // slot table of a block
  static constexpr int NSL = (SLB + 1023) / 1024;   // its DMA requests; the last one runs 640 bytes into the next block's table
  static constexpr int SLL = NSL * 1024;            // ... so a slot buffer is 4 KiB in LDS
  static constexpr int SRB = COUT * 2 + 16;         // padded row pitch of the epilogue scratch
  static constexpr int NRES = BM * COUT * 2 / 1024; // residual requests of a block
  static constexpr int OFF_X = 0;
  static constexpr int OFF_SLOT = NXB * XB;                   // two slot tables: this block's, the next one's
  static constexpr int OFF_EPI = OFF_SLOT + 2 * SLL;
  static constexpr int OFF_RES = OFF_EPI + BM * SRB;          // residual pieces: [pass][64 lanes] x 16 B
  static constexpr int OFF_CONST = OFF_RES + NRES * 1024;     // scale [32] f32, shift [32] f32, bias [32] 16-bit
  static constexpr int OFF_HDR = OFF_CONST + 32 * 4 * 2 + 64;    // the header of the block after next (24 bytes, fetched as 32)
  static constexpr int BYTES = OFF_HDR + 32;
  static_assert(CAP % RPI == 0, "CAP must be a whole number of DMA instructions");
  static_assert((CAP + 1) * RB <= 0xFFFF, "baked offsets are 16-bit");
  static_assert(PX <= TAPS - 2, "one row request per tap, the last two taps carry the slot / residual requests");
  static_assert(NSL == 4 && NRES == 4, "two taps x two requests");
  static_assert(3 * PX + NSL + NRES < 60, "vmcnt is a 6-bit counter");
  static_assert(PX + NSL + 1 <= 15 && PX + NRES <= 15, "the run-time counted wait covers 0..15 (rows of one piece + the extras of one piece)");
  static_assert(OFF_SLOT % 16 == 0 && OFF_EPI % 16 == 0 && OFF_RES % 16 == 0 && OFF_CONST % 16 == 0 && OFF_HDR % 16 == 0, "16-byte aligned regions");
};

template <int DT, int CAP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 1)))
void spconv_slabf_kernel(SlabArgs sa) {
  typedef PlanF<CAP> P;
  typedef typename Num<DT>::T T;
  extern __shared__ u32x4 lds[];
  char* const L = (char*)lds;
  lds_char* const L3 = (lds_char*)(void*)lds;
  const Args& a = sa.a;
  const int m = a.m_dev ? (*a.m_dev < a.m_cap ? *a.m_dev : a.m_cap) : a.m_cap;
  const int nblk = (m + P::BM - 1) / P::BM;
  // XCD x owns the contiguous block range [x*per, (x+1)*per); its gx waves walk it round-robin
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int per = (nblk + 7) >> 3;
  const int blk_end = (xcd + 1) * per < nblk ? (xcd + 1) * per : nblk;
  int blk = xcd * per + bix;
  if (blk >= blk_end) return;
  int blk_n = blk + gx;
  const int lane = threadIdx.x;
  const unsigned lane16_ = (unsigned)lane * 16u;
  const int r32 = lane & 31, h = lane >> 5;   // MFMA operand layout: row (or output channel) of the 32-tile, 8-channel group of the 16
  const unsigned G0 = (unsigned)h * 16u, G1 = G0 ^ 32u;   // the lane's 16-byte piece of a staged row, channel half 0 / 1

  const unsigned row_bytes = (unsigned)a.feat_stride * 2u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.feat, 0, (unsigned)a.n_in * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wimg, 0, sa.wimg_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)sa.slots, 0, sa.slot_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_h = __builtin_amdgcn_make_buffer_rsrc((void*)sa.hdr, 0, (unsigned)((a.m_cap + P::BM - 1) / P::BM) * (unsigned)(PLANES * 8), 0x00020000);
  const unsigned res_pitch = (unsigned)a.res_stride * 2u;
  const __amdgpu_buffer_rsrc_t rs_r =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.residual ? a.residual : a.feat), 0, a.residual ? (unsigned)a.m_cap * res_pitch : 0u, 0x00020000);

  // ---- once per kernel: the filter, the zero rows, the per-channel epilogue operands -------------------------------------
  // A fragment (tap k, channel half kk) of lane (r32 = output channel, h): W[k][16 kk + 8 h .. + 7][r32] = the 16 bytes lane
  // (r32 % 16) + 16 (2 kk + h) holds in the 16x16x32 filter image's fragment (k, output tile r32 / 16)
  u32x4 wf[27][2];
#pragma unroll
  for (int k = 0; k < 27; ++k)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      wf[k][kk] = __builtin_amdgcn_raw_buffer_load_b128(
          rs_w, (unsigned)((k * 2 + (r32 >> 4)) * 1024 + ((r32 & 15) + 16 * (2 * kk + h)) * 16), 0u, 0);
  if (lane < P::NXB * P::PPR) {
    const int b = lane / P::PPR, p = lane % P::PPR;
    *(u32x4*)(L + P::OFF_X + b * P::XB + p * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  if (lane < 32) {
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
  // Inside the loop a header comes by LDS-DMA a whole block before it is needed: a scalar load of it sat in front of the first
  // LDS wait of every block with its full latency (measured: 22 % of the kernel).
  auto issue_hdr = [&](bool go, int b) {
    if (go && lane < 2) dma16_l(rs_h, lane16_, (unsigned)b * (unsigned)(PLANES * 8), L3 + P::OFF_HDR);
  };
  auto read_hdr = [&](bool ok) {
    Hdr hd;
#pragma unroll
    for (int j = 0; j < PLANES; ++j) {
      const int2 v = *(const int2*)(L + P::OFF_HDR + j * 8);
      hd.lo[j] = __builtin_amdgcn_readfirstlane(v.x);
      hd.cnt[j] = ok ? __builtin_amdgcn_readfirstlane(v.y) : 0;
    }
    return hd;
  };

  // ---- requests -------------------------------------------------------------------------------------------------------
  // Row request i of a piece: source rows [lo + q*CAP + 16 i, + 16) -> LDS rows 1 + 16 i .. of buffer `buf`.  The DMA writes
  // lane l to LDS row e = 1 + 16 i + l/4, piece l%4, which must hold source piece (l%4) ^ ((e >> 2) & 3): the involution is
  // applied to the SOURCE address (it does not depend on i).  A piece issues exactly the requests its range needs; the counted
  // waits below take the counts at run time.
  const unsigned lr = (unsigned)(lane / P::PPR), sp = (unsigned)(lane % P::PPR);
  const unsigned lane_row_off = lr * row_bytes + (sp ^ (((1u + lr) >> 2) & 3u)) * 16u;   // loop-invariant: a request costs no VALU
  struct RowReq { int n; unsigned soff; lds_char* dst; };   // n requests: source rows [first, first + 16 n) -> LDS rows 1 .. of buffer j
  auto row_req = [&](bool go, const Hdr& hd, int j, int q) {
    const int cnt = (int)((unsigned)hd.cnt[j] & ~HDR_RAW);
    int n = go ? cnt - q * CAP : 0;
    n = n < 0 ? 0 : (n < CAP ? n : CAP);
    RowReq rq;
    rq.n = (n + P::RPI - 1) / P::RPI;   // the last request may run past the range: rows no slot refers to (past the tensor: zeros)
    rq.soff = (unsigned)(hd.lo[j] + q * CAP) * row_bytes;
    rq.dst = L3 + (P::OFF_X + j * P::XB + P::RB);
    return rq;
  };
  auto issue_row = [&](const RowReq& rq, int i) {
    if constexpr (BEVAMD_SLABF_EXP & 1) return;
    if (i < rq.n) dma16_l(rs_x, lane_row_off, rq.soff + (unsigned)(i * P::RPI) * row_bytes, rq.dst + i * 1024);
  };
  // slot table of block b -> slot buffer sb, request i of NSL
  const unsigned lane16 = (unsigned)lane * 16u;
  auto issue_slots = [&](bool go, int b, int sb, int i) {
    if constexpr (BEVAMD_SLABF_EXP & 16) return;
    if (go) dma16_l(rs_s, lane16, (unsigned)b * (unsigned)P::SLB + (unsigned)(i * 1024), L3 + (P::OFF_SLOT + sb * P::SLL + i * 1024));
  };
  // residual pieces of block b, pass i: lane -> row 16 i + lane/4, 16-byte piece lane%4
  const int j4 = lane & 3, rsub = lane >> 2, col0 = j4 * 8;
  const unsigned lane_res_off = (unsigned)rsub * res_pitch + (unsigned)j4 * 16u;
  const bool has_res = a.residual != nullptr;
  auto issue_residual = [&](int b, int i) {
    if constexpr (BEVAMD_SLABF_EXP & 16) return;
    if (has_res) dma16_l(rs_r, lane_res_off, (unsigned)(b * P::BM + i * 16) * res_pitch, L3 + (P::OFF_RES + i * 1024));
  };

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // ---- the nine taps of plane J.  FAST: baked slots, the whole range staged by the pipeline.  `tapwork(d)` is called once per
  // tap in front of its MFMAs: the requests of the pieces ahead, spread one per tap. ------------------------------------------
  auto plane = [&](auto jc, auto fastc, const Hdr& hd, int sb, auto&& tapwork) {
    constexpr int J = decltype(jc)::value;
    constexpr bool FAST = decltype(fastc)::value;
    const char* X = L + P::OFF_X + J * P::XB;
    const uint16_t* sl = (const uint16_t*)(L + P::OFF_SLOT + sb * P::SLL) + J * TAPS * P::BM + r32;
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
      u32x4 xa[3][2][2];   // [tap % 3][tile][channel half]: fragments are requested TWO taps ahead (an LDS round trip under load is
                           // longer than the two MFMAs that separate a request from its use one tap ahead), slots three
      auto load_slots = [&](int d) {
#pragma unroll
        for (int t = 0; t < 2; ++t) raw[d % 4][t] = (unsigned)sl[d * P::BM + t * 32];
      };
      auto fetch = [&](int d, int t) {
        unsigned off;
        if constexpr (FAST) {
          off = raw[d % 4][t];
        } else {
          const unsigned s = rawslots ? raw[d % 4][t] : (raw[d % 4][t] >> 6) - 1u;   // baked 0 -> 0xFFFFFFFF: outside any piece
          const unsigned pbase = (unsigned)(q * CAP), plive = (unsigned)cnt - pbase;
          const unsigned prow = plive < (unsigned)CAP ? plive : (unsigned)CAP;
          const unsigned e = s - pbase;
          off = e < prow ? baked_entry(e + 1u) : 0u;
        }
        xa[d % 3][t][0] = *(const u32x4*)(X + (off ^ G0));   // one v_xad_u32 each: (entry ^ piece) + buffer
        xa[d % 3][t][1] = *(const u32x4*)(X + (off ^ G1));
      };
      load_slots(0);
      load_slots(1);
      load_slots(2);
      fetch(0, 0);
      fetch(0, 1);
      fetch(1, 0);
      fetch(1, 1);
      // One wave per SIMD issues in order: an MFMA that has to wait for the pipe holds back everything behind it.  So the work
      // of the NEXT tap is cut in four and one part goes behind each of this tap's four MFMAs (<= 5 single-issue instructions
      // fit under a 32-cycle MFMA: MI355X_MICROARCH.md); with "all fetches, then four MFMAs" the same loop ran at half the rate.
#pragma unroll
      for (int d = 0; d < TAPS; ++d) {
        const int k = J * TAPS + d;
        constexpr bool ONE = (BEVAMD_SLABF_EXP & 4) != 0, NOFETCH = (BEVAMD_SLABF_EXP & 2) != 0;
        acc[0] = mfma32<DT>(wf[k][0], xa[d % 3][0][0], acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (d + 3 < TAPS) load_slots(d + 3);
        if (FAST || q == pieces - 1) tapwork(d);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!ONE) acc[1] = mfma32<DT>(wf[k][0], xa[d % 3][1][0], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!NOFETCH) if (d + 2 < TAPS) fetch(d + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!ONE) acc[0] = mfma32<DT>(wf[k][1], xa[d % 3][0][1], acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!NOFETCH) if (d + 2 < TAPS) fetch(d + 2, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!ONE) acc[1] = mfma32<DT>(wf[k][1], xa[d % 3][1][1], acc[1]);
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

  // ---- epilogue: the tile, rounded to 16 bits, through wave-private LDS; then 64 whole rows, 16 rows x 4 pieces per pass ----
  char* const scr = L + P::OFF_EPI;
  auto finish_block = [&](int b) {
    if constexpr (BEVAMD_SLABF_EXP & 8) {   // keep the accumulators alive with (practically) no store
      float t = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) t += acc[0][e] + acc[1][e];
      if (t == 12345.678f) ((float*)a.out)[lane] = t;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tt][e] = 0.f;
      return;
    }
    // D[channel i][row j]: lane (j = r32, h) holds channels 8 (e/4) + 4 h + e%4, e = 0..15, of rows r32 and 32 + r32
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) {
        T p[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) p[e] = Num<DT>::from_f32(acc[t][qd * 4 + e]);
        *(uint2*)(scr + (t * 32 + r32) * P::SRB + (8 * qd + 4 * h) * 2) = *(const uint2*)p;
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
    for (int pass = 0; pass < 4; ++pass) {
      const int r = pass * 16 + rsub;
      const int row = b * P::BM + r;
      const u32x4 rawv = *(const u32x4*)(scr + r * P::SRB + j4 * 16);
      const u32x4 resv = *(const u32x4*)(L + P::OFF_RES + pass * 1024 + lane * 16);
      if (row < m) {
        const u32x4 o = finish8<DT>(rawv, a.bias != nullptr, bias16, a.scale != nullptr, sv, hv, has_res, resv, a.relu != 0);
        *(u32x4*)((T*)a.out + (size_t)row * a.out_stride + col0) = o;
      }
    }
  };

  // ---- pipeline ---------------------------------------------------------------------------------------------------------
  // During piece (b, j) the rows of the piece two ahead are requested, one request behind the first MFMA of taps 0, 1, ...;
  // on taps 7-8 the slot table of block b+1 (j = 0) or the residual rows of block b (j = 1).  Loads complete in order, so "at
  // most N requests pending", N = what was requested AFTER the rows this piece reads, says that they have landed (stores in
  // flight can only make the wait longer).  The counts are run-time values: the wait is a 16-way switch of immediates.
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
#pragma unroll
    for (int i = 0; i < P::NSL; ++i) issue_slots(true, blk, 0, i);
#pragma unroll
    for (int i = 0; i < P::PX; ++i) issue_row(r_b, i);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), through the builtin: hipcc then knows the filter has arrived and puts no waits for it into the loop
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // zero rows and epilogue operands are in LDS
  }
  int sb = 0;
  int extra_prev = 0;   // slot / residual requests of the previous piece (issued after its row requests)
  for (;;) {
    const bool has_n = blk_n < blk_end;
    const bool has_nn = blk_n + gx < blk_end;
    const int n_slots = (has_n ? P::NSL : 0) + (has_nn ? 1 : 0), n_res = has_res ? P::NRES : 0;
    {   // plane 0 reads r_a (in flight behind it: extras of the piece before last = none [plane 1's residual came BEFORE... see order], r_b, extras of the last piece)
      const RowReq rq = row_req(true, hc, 2, 0);
      wait_pending(r_b.n + extra_prev);
      run_plane(std::integral_constant<int, 0>{}, hc, sb, [&](int d) {
        if (d < P::PX) issue_row(rq, d);
        if (d >= TAPS - 2) { issue_slots(has_n, blk_n, sb ^ 1, (d - (TAPS - 2)) * 2); issue_slots(has_n, blk_n, sb ^ 1, (d - (TAPS - 2)) * 2 + 1); }
        if (d == TAPS - 1) issue_hdr(has_nn, blk_n + gx);
      });
      r_a = r_b; r_b = rq; extra_prev = n_slots;
    }
    {   // plane 1
      const RowReq rq = row_req(has_n, hn, 0, 0);
      wait_pending(r_b.n + extra_prev);
      run_plane(std::integral_constant<int, 1>{}, hc, sb, [&](int d) {
        if (d < P::PX) issue_row(rq, d);
        if (d >= TAPS - 2) { issue_residual(blk, (d - (TAPS - 2)) * 2); issue_residual(blk, (d - (TAPS - 2)) * 2 + 1); }
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
    wait_pending(r_b.n);   // the residual pieces (requested before this plane's row requests)
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
