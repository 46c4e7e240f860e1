// Staged-rows submanifold convolution, second cut: filter fragments in REGISTERS, no barrier inside a kernel plane.
//
// spconv_slab.h stages the rows of a kernel plane AND the filter of every tap in LDS; each tap then costs a filter DMA, a
// counted wait and a workgroup barrier, and 8 of a wave's 12 fragment reads per tap are filter fragments.  The compile-time
// ablation (profiles/r02_slab_ablation.txt) shows a latency-serial kernel: with a barrier every 16 MFMAs nothing overlaps.
//
// Here the workgroup's waves form an RW x CW grid: wave (r, c) owns 16*MT rows x (Cout / CW) output channels.  The filter
// image is already in MFMA-fragment order (1 KiB per (tap, 32-channel chunk, 16 output channels)), so a wave loads ITS
// fragments of a tap with NTW*CH coalesced 16-byte buffer loads straight into registers, two taps ahead (a three-set register
// ring, static indices after unrolling the 9 taps of a plane); every filter byte crosses the L1 RW times per block
// instead of going through LDS.  Only the staged rows live in LDS (same LDS-DMA, same swizzle, same slot table and block
// headers as spconv_slab.h — slab_build_kernel is shared), so the ONLY workgroup barrier left is the one that switches
// planes (3 per block, + pieces).  Row fragments are software-pipelined across the taps of a plane: the ds_read_b128 of
// unit i+1 are issued before the MFMAs of unit i, so one LDS round trip is exposed per plane, not per tap.
//
// Summation order per output element is unchanged (kernel offset ascending, then 32-channel chunk): results are
// bit-identical to spconv_slab.h and, for Cin <= 64, to the gather kernels.
#pragma once
#include "spconv_slab.h"

#ifndef BEVAMD_SLABR_EXP
#define BEVAMD_SLABR_EXP 0   // experiment builds (tools/exp_build.sh): 1 = s_setprio around the MFMA groups, 2 = fragment reads interleaved with the MFMAs
#endif

namespace bevamd {
namespace slab {

// KC  channels staged per row (32 | 64; CIN / KC passes)      MT  16-row tiles per wave
// RW  wave rows, CW wave columns (block = RW*16*MT rows)       CAP rows of one X buffer
template <int KC, int CIN, int NT, int MT, int RW, int CW, int CAP>
struct PlanR {
  static_assert(KC == 32 || KC == 64, "staged row = 32 or 64 channels");
  static_assert(CIN % KC == 0 && NT % CW == 0, "bad split");
  static constexpr int NW = RW * CW;
  static constexpr int BM = RW * 16 * MT;
  static constexpr int NTW = NT / CW;               // 16-channel output tiles per wave
  static constexpr int RB = KC * 2;                 // staged bytes per row
  static constexpr int PPR = RB / 16;               // 16-byte pieces per row
  static constexpr int RPI = 64 / PPR;              // rows per DMA instruction (1 KiB)
  static constexpr int CH = KC / 32;                // 32-channel chunks per staged row
  static constexpr int CPB = CIN / 32;              // chunks per kernel offset in the filter image
  static constexpr int NH = CIN / KC;               // channel passes
  static constexpr int NB = NTW * CH;               // filter fragments (16 B per lane) per tap per wave
  static constexpr int WD = 2;                      // taps of filter lookahead (ring of WD + 1 register sets; 9 % (WD + 1) == 0)
  static constexpr int NXB = 2;                     // X buffers: this piece and the next
  static constexpr int XB = ((CAP + 1) * RB + 1023) / 1024 * 1024;   // one X buffer incl. the zero row, KiB-aligned
  static constexpr int PX = CAP / RPI;              // 1 KiB DMA pieces of a full X buffer
  static constexpr int NX = (PX + NW - 1) / NW;     // ... per wave: a FIXED count, so that s_waitcnt can count
  static constexpr int OFF_X = 0;
  static constexpr int OFF_SLOT = NXB * XB;
  static constexpr int OFF_DUMP = OFF_SLOT + 27 * BM * 2;   // landing zone of the dummy pieces
  static constexpr int BYTES = OFF_DUMP + 1024;
  static_assert(NW * EpiScratch<NTW>::U4 * 16 <= NXB * XB, "epilogue scratch must fit the X buffers it aliases");
  static_assert(CAP % RPI == 0, "CAP must be a whole number of DMA instructions");
  static_assert((CAP + 1) * RB < 65536, "row offsets are 16-bit");
  static_assert(WD * NB + NX < 60, "vmcnt is a 6-bit counter");
  static_assert(TAPS % (WD + 1) == 0, "register ring must close over a plane");
};

// FLAGS (the bits of a shape's ID above 3; IDs 0-3 only tell equal shapes with different CAP apart):
//   8  F_BAKED: the slot metadata arrives as LDS byte offsets with the bank swizzle folded in (spconv_slab_meta.h, FMT_BAKED128;
//      the zero row is LDS row 0, staged row s is row s + 1).  The kernel issues ~47 instructions per 8 MFMAs without it, 22 of
//      them the VALU arithmetic slot -> (row, swizzle) -> fragment address, and a SIMD starts one instruction of a wave about every
//      4 cycles: the loop is ISSUE-bound (round 5: predicating the zero-row reads away removed 57 % of the bank-conflict cycles and
//      made the kernel SLOWER, EXPERIMENTS.md C.2 — it is not the LDS).  With baked entries a fragment address is one v_xad_u32.
//      A plane whose range does not fit one piece (> CAP rows) or whose header says raw takes the arithmetic path.
constexpr int F_BAKED = 8;
template <int DT, int KC, int CIN, int NT, int MT, int RW, int CW, int CAP, int FLAGS = 0>
__global__ __launch_bounds__(RW * CW * 64, ((FLAGS & 8) && CIN > 64 && MT <= 4) ? 2 : 1) void spconv_slabr_kernel(SlabArgs sa) {
  typedef PlanR<KC, CIN, NT, MT, RW, CW, CAP> P;
  constexpr bool BK = (FLAGS & F_BAKED) != 0;   // baked 128-byte-row metadata
  static_assert(!BK || KC == 64, "the baked format describes 128-byte staged rows");
  constexpr bool BAKED = !BK && P::BM == BAKED_ROWS;   // 64-row blocks: the metadata is in the filter-stationary kernels' baked format
  typedef WaveTile<DT, (CIN > 64 ? 64 : CIN), P::NTW, MT, (CIN > 64 ? 64 : CIN) / 32> WT;   // accumulators + epilogue only
  typedef typename Num<DT>::T T;
  extern __shared__ u32x4 lds[];
  char* const L = (char*)lds;
  const Args& a = sa.a;
  const int m = a.m_dev ? (*a.m_dev < a.m_cap ? *a.m_dev : a.m_cap) : a.m_cap;
  const int nblk = (m + P::BM - 1) / P::BM;
  // XCD-aware block map: XCD x walks a contiguous range of blocks (neighbouring blocks share staged rows in its L2)
  const int xcd = blockIdx.x & 7, bix = blockIdx.x >> 3;
  const int per = (nblk + 7) >> 3;
  const int blk = xcd * per + bix;
  if (bix >= per || blk >= nblk) return;   // the whole workgroup leaves together
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform for hipcc too (no waterfall loops around the DMA)
  const int wr = w / CW, wc = w % CW;
  const int c = lane & 15, g4 = lane >> 4;

  // ---- slot table -> LDS; zero rows; block header -> one VGPR pair (lane j = plane j) ------------------------------
  uint16_t* slot = (uint16_t*)(L + P::OFF_SLOT);
  {
    const u32x4* src = (const u32x4*)(sa.slots + (size_t)blk * 27 * P::BM);
    constexpr int N16 = 27 * P::BM * 2 / 16;
    for (int i = tid; i < N16; i += P::NW * 64) ((u32x4*)slot)[i] = src[i];
    if (tid < P::NXB * P::PPR) {
      const int b = tid / P::PPR, p = tid % P::PPR;
      *(u32x4*)(L + P::OFF_X + b * P::XB + (BK ? 0 : CAP * P::RB) + p * 16) = u32x4{0u, 0u, 0u, 0u};   // the zero row: last, or first (baked)
    }
  }
  const int2 hl = sa.hdr[(size_t)blk * PLANES + (lane < PLANES ? lane : 0)];
  const int vlo = hl.x, vcnt = lane < PLANES ? ((BAKED || BK) ? (int)((unsigned)hl.y & ~HDR_RAW) : hl.y) : 0;
  // baked metadata (spconv_slab_meta.h): a slot is (row + 1) * row bytes | swizzle bits, 0 = none — unless the plane's header says raw
  const unsigned raw_planes = (BAKED || BK) ? (unsigned)__builtin_amdgcn_readfirstlane((int)__ballot(lane < PLANES && ((unsigned)hl.y & HDR_RAW))) : ~0u;
  const unsigned live_planes = (unsigned)__builtin_amdgcn_readfirstlane((int)__ballot(vcnt > 0));   // bit j: plane j has rows
  auto plane_lo = [&](int j) { return __builtin_amdgcn_readlane(vlo, j); };
  auto plane_cnt = [&](int j) { return __builtin_amdgcn_readlane(vcnt, j); };
  auto next_plane = [&](int from) {   // first plane >= from with rows, PLANES if none
    const unsigned rest = from < PLANES ? live_planes >> from : 0u;
    return rest ? from + (int)__builtin_ctz(rest) : PLANES;
  };
  auto next_sub = [&](Sub u) {
    if (u.done) return u;
    if ((u.q + 1) * CAP < plane_cnt(u.j)) { ++u.q; return u; }
    u.q = 0;
    u.j = next_plane(u.j + 1);
    if (u.j < PLANES) return u;
    u.j = next_plane(0);
    if (++u.h >= P::NH) u.done = true;
    return u;
  };

  const unsigned row_bytes = (unsigned)a.feat_stride * 2u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.feat, 0, (unsigned)a.n_in * row_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wimg, 0, sa.wimg_bytes, 0x00020000);
  char* const dump = L + P::OFF_DUMP;
  const unsigned lr = (unsigned)(lane / P::PPR), sp = (unsigned)(lane % P::PPR);
  const unsigned lane_piece_off = (sp ^ RowSwz<KC>::of(lr + (BK ? 1u : 0u))) * 16u;   // baked: staged row s lives in LDS row s + 1
  static_assert(P::RPI % 8 == 0, "swizzle must not depend on the instruction index");
  // rows of piece (j, q), channel pass h -> X buffer xb: exactly NX 1-KiB requests per wave (pieces past the range re-read
  // its last row into rows no slot refers to, or into the dump); a finished `u` sends all of them to the dump
  auto stage_x = [&](const Sub& u, int xb) {
    const int j = u.done ? next_plane(0) : u.j;
    const int n = plane_cnt(j) - (u.done ? 0 : u.q * CAP);
    const unsigned rows = (unsigned)(n < CAP ? n : CAP);
    const unsigned soff = (unsigned)(plane_lo(j) + (u.done ? 0 : u.q * CAP)) * row_bytes + (unsigned)((u.done ? 0 : u.h) * KC * 2);
    char* dst = L + P::OFF_X + xb * P::XB + (BK ? P::RB : 0);
#pragma unroll
    for (int t = 0; t < P::NX; ++t) {
      const int i = w + t * P::NW;
      unsigned r = (unsigned)(i * P::RPI) + lr;
      r = r < rows ? r : rows - 1u;
      dma16(rs_x, r * row_bytes + lane_piece_off, soff, (i < P::PX && !u.done) ? dst + i * 1024 : dump);
    }
  };
  // this wave's filter fragments of tap d (0..8) of piece u: NB coalesced 16-byte loads
  auto load_w = [&](const Sub& u, int d, u32x4 (&wf)[P::NB]) {
    const int k = u.j * TAPS + d;
#pragma unroll
    for (int cc = 0; cc < P::CH; ++cc)
#pragma unroll
      for (int nt = 0; nt < P::NTW; ++nt)
        wf[cc * P::NTW + nt] = __builtin_amdgcn_raw_buffer_load_b128(
            rs_w, (unsigned)lane * 16u, (unsigned)(((k * P::CPB + u.h * P::CH + cc) * NT + wc * P::NTW + nt) * 1024), 0);
  };
  // Row addressing is split in two so that no LDS round trip sits in front of an MFMA group: the 16-bit slots of tap d are
  // READ two taps ahead (raw ring, like the filter), and turned into LDS addresses (byte address of piece 0 | swizzle << 20)
  // just before the tap's first fragment read, one reduction unit later at the earliest.
  auto load_slots = [&](const Sub& u, int d, unsigned (&raw)[MT]) {
    const uint16_t* sl = slot + (u.j * TAPS + d) * P::BM + wr * 16 * MT + c;
    if constexpr (BK) {
      // sign-extending 16-bit loads (ds_read_i16: hipcc puts a v_and 0xffff behind every ds_read_u16): baked entries are < 0x8000
      // ((CAP + 1) * 128 bytes), so they come out as they are; the arithmetic path masks what it reads
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) raw[mt] = (unsigned)(int)((const int16_t*)sl)[mt * 16];
      return;
    }
    const bool israw = !BAKED || ((raw_planes >> u.j) & 1u);   // wave-uniform
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const unsigned r = (unsigned)sl[mt * 16];
      raw[mt] = israw ? r : (r >> 6) - 1u;   // baked 0 -> 0xFFFFFFFF: outside any piece, like NO_SLOT
    }
  };
  static_assert(!BK || (CAP + 1) * P::RB < 0x8000, "baked entries are read as signed 16-bit values");
  // wave-uniform, once per piece: the whole range of the plane is resident and its slots are baked — the entries ARE the offsets
  auto piece_is_fast = [&](const Sub& u) {
    return BK && !((raw_planes >> u.j) & 1u) && u.q == 0 && plane_cnt(u.j) <= CAP;
  };
  auto to_offsets = [&](const Sub& u, bool fast, const unsigned (&raw)[MT], unsigned (&xo)[MT]) {
    const unsigned pbase = (unsigned)(u.q * CAP);
    const unsigned plive = (unsigned)plane_cnt(u.j) - pbase;
    const unsigned prow = plive < (unsigned)CAP ? plive : (unsigned)CAP;   // rows of this piece
    if constexpr (BK) {
      if (fast) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xo[mt] = raw[mt];
      } else {
        const bool planeraw = (raw_planes >> u.j) & 1u;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const unsigned r16 = raw[mt] & 0xFFFFu;                           // undo the sign extension of the load
          const unsigned sidx = planeraw ? r16 : (r16 >> 7) - 1u;           // baked 0 -> 0xFFFFFFFF; raw NO_SLOT stays 0xFFFF
          const unsigned e = sidx - pbase;
          const unsigned r = e < prow ? e + 1u : 0u;                        // outside the piece: the zero row (row 0)
          xo[mt] = baked128_entry(r);
        }
      }
      return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      unsigned e = raw[mt] - pbase;        // NO_SLOT - pbase stays >= prow
      e = e < prow ? e : (unsigned)CAP;    // outside the piece: the zero row
      xo[mt] = e * P::RB + (RowSwz<KC>::of(e) << 20);
    }
  };
  const unsigned piece_xor[2] = {(unsigned)g4 << 4, (unsigned)(4 + g4) << 4};   // baked: the lane's 16-byte piece of chunk 0 / 1

  // output channels [c0, c0 + 16*NTW) of the rows: the epilogue sees a narrower convolution
  Args aw = a;
  const int c0 = wc * P::NTW * 16;
  aw.out = (void*)((T*)a.out + c0);
  if (a.bias) aw.bias = (const void*)((const T*)a.bias + c0);
  if (a.scale) { aw.scale = a.scale + c0; aw.shift = a.shift + c0; }
  if (a.residual) aw.residual = (const void*)((const T*)a.residual + c0);
  aw.cout = a.cout - c0;
  WT wt;
  wt.init(aw, blk * P::BM + wr * 16 * MT, m, nullptr, (u32x4*)(L + P::OFF_X) + w * EpiScratch<P::NTW>::U4);

  Sub sub[2];   // this piece, the next
  sub[0] = Sub{0, next_plane(0), 0, false};   // the centre plane always has rows in a live block
  sub[1] = next_sub(sub[0]);
  u32x4 wf[P::WD + 1][P::NB];   // filter ring: tap d of a piece lives in set d % 3
  stage_x(sub[0], 0);
  load_w(sub[0], 0, wf[0]);
  load_w(sub[0], 1, wf[1]);
  wait_dma<0>();
  __syncthreads();   // rows of the first piece landed; also publishes the slot table and the zero rows
  int xb = 0;
  unsigned raw[P::WD + 1][MT];   // slot ring: tap d of a piece lives in set d % 3
  load_slots(sub[0], 0, raw[0]);
  load_slots(sub[0], 1, raw[1]);
  for (;;) {
    const char* X = L + P::OFF_X + xb * P::XB;
    constexpr int U = TAPS * P::CH;   // reduction units of a piece: (tap, 32-channel chunk)
    unsigned xo[2][MT];
    u32x4 xa[2][MT];
    auto fetch = [&](int i) {
      const int d = i / P::CH, cc = i % P::CH;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const unsigned o = xo[d & 1][mt];
        if constexpr (BK) xa[i & 1][mt] = *(const u32x4*)(X + (o ^ piece_xor[cc]));
        else xa[i & 1][mt] = *(const u32x4*)(X + (o & 0xFFFFFu) + (((unsigned)(cc * 4 + g4)) ^ (o >> 20)) * 16);
      }
    };
    const bool fast = piece_is_fast(sub[0]);
    to_offsets(sub[0], fast, raw[0], xo[0]);
    fetch(0);
#pragma unroll
    for (int d = 0; d < TAPS; ++d) {
      // requests: the filter of the tap WD ahead (the next piece's first taps at the end of this one); at the first tap the
      // rows of the next piece, AFTER the filter request — a filter load issued behind them could not complete before them
      {
        const int dn = d + P::WD;
        const Sub& un = dn < TAPS ? sub[0] : (sub[1].done ? sub[0] : sub[1]);   // past the last piece: a valid, unused load
        if constexpr (!(BEVAMD_SLABR_EXP & 4)) load_w(un, dn % TAPS, wf[dn % (P::WD + 1)]);
        else if (d == 0) load_w(un, dn % TAPS, wf[dn % (P::WD + 1)]);   // ablation: one filter set per plane instead of nine
        load_slots(un, dn % TAPS, raw[dn % (P::WD + 1)]);
      }
      if constexpr (!(BEVAMD_SLABR_EXP & 8)) { if (d == 0) stage_x(sub[1], xb ^ 1); }   // ablation 8: no row staging after the first piece
#pragma unroll
      for (int cc = 0; cc < P::CH; ++cc) {
        const int i = d * P::CH + cc;
#if BEVAMD_SLABR_EXP & 2   // experiment: one fragment read of unit i + 1 behind every NTW MFMAs of unit i instead of all reads in front
        if (i + 1 < U && (i + 1) % P::CH == 0) to_offsets(sub[0], fast, raw[(d + 1) % (P::WD + 1)], xo[(d + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
          for (int nt = 0; nt < P::NTW; ++nt)
            wt.acc[mt][nt] = mfma<DT>(wf[d % (P::WD + 1)][cc * P::NTW + nt], xa[i & 1][mt], wt.acc[mt][nt]);
          __builtin_amdgcn_sched_barrier(0);
          if (i + 1 < U) {
            const int d1 = (i + 1) / P::CH, c1 = (i + 1) % P::CH;
            const unsigned o = xo[d1 & 1][mt];
            if constexpr (BK) xa[(i + 1) & 1][mt] = *(const u32x4*)(X + (o ^ piece_xor[c1]));
            else xa[(i + 1) & 1][mt] = *(const u32x4*)(X + (o & 0xFFFFFu) + (((unsigned)(c1 * 4 + g4)) ^ (o >> 20)) * 16);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#else
        if (i + 1 < U) {
          if ((i + 1) % P::CH == 0) to_offsets(sub[0], fast, raw[(d + 1) % (P::WD + 1)], xo[(d + 1) & 1]);
          if constexpr (!(BEVAMD_SLABR_EXP & 16)) fetch(i + 1);   // ablation 16: no fragment reads after the first unit
          else if (i == 0) fetch(1);
        }
        __builtin_amdgcn_sched_barrier(0);
#if BEVAMD_SLABR_EXP & 1   // experiment: the wave that has its operands goes first
        __builtin_amdgcn_s_setprio(2);
#endif
#pragma unroll
        for (int nt = 0; nt < P::NTW; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
            if constexpr (BEVAMD_SLABR_EXP & 32) { if (nt + mt) continue; }   // ablation 32: one MFMA per unit instead of NTW * MT
            wt.acc[mt][nt] = mfma<DT>(wf[d % (P::WD + 1)][cc * P::NTW + nt], xa[i & 1][mt], wt.acc[mt][nt]);
          }
#if BEVAMD_SLABR_EXP & 1
        __builtin_amdgcn_s_setprio(0);
#endif
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
    // the next piece's rows (requested nine taps ago, in front of all but the two newest filter sets) have landed; every
    // wave is done reading this piece
    wait_dma<P::WD * P::NB>();
    barrier_keep_dma();
    if (sub[1].done) break;
    sub[0] = sub[1];
    sub[1] = next_sub(sub[1]);
    xb ^= 1;
  }
  // Epilogue scratch aliases the X buffers: every wave passed the last barrier, nobody reads X any more.  (Requesting the
  // residual rows at the start of the last plane instead of here was measured: +16 live registers, one wave per SIMD less.)
  if constexpr (BEVAMD_SLABR_EXP & 64) {   // ablation 64: no epilogue (one store per wave keeps the accumulators alive)
    float t = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < P::NTW; ++nt) t += wt.acc[mt][nt][0] + wt.acc[mt][nt][1] + wt.acc[mt][nt][2] + wt.acc[mt][nt][3];
    if (t == 12345.678f) ((float*)a.out)[tid] = t;
    return;
  }
  wt.store(aw);
}

}  // namespace slab
}  // namespace bevamd
