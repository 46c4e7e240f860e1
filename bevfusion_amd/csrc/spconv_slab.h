// Staged-rows ("slab") submanifold convolution for gfx950, 16-bit features, fp32 accumulate.
//
// Same operator and same results as the gather kernels of spconv_tile.h (reference: spconv_ops.h:260-361 indiceConv<T> with
// subM = 1, 3x3x3), for voxel sets whose rows are in ASCENDING LINEAR INDEX (every level a strided convolution produced:
// the row order of the reference's CUDA path).  What it changes is where the operand rows come from.
//
// The gather kernels fetch, for every output row and every one of the 27 kernel offsets, the neighbour's feature row from
// L2 through the texture path (19 of 27 exist on average): 64-byte segments, one address cycle each, then a ds_bpermute
// relayout — profiles/r01_spconv_pmc.txt: MFMA busy 19 %, LDS bank conflicts 55-83 % of LDS-active cycles.  But in linear-
// index order the neighbour of row o at offset (kx, ky, kz) sits at linear index L(o) + const, so for a block of BM
// consecutive output rows the inputs read through ONE kernel line (kx, ky) — its three kz taps — are a CONTIGUOUS range of
// input rows, on average BM + 1 rows long (measured: 9.05 staged rows per output row instead of 19-27 gathered ones).
//
//   slab_build_kernel (once per voxel set, shared by the 4 SubM convolutions of a level): per block of BM rows and kernel
//     line j: (lo, cnt) of that range, and the neighbour table rewritten as 16-bit slots relative to lo (0xFFFF = no
//     neighbour): 54 B per row instead of 108, no range reduction inside the convolution.
//   spconv_slab_kernel: a workgroup owns a block.  Per line, the range is copied global -> LDS with LDS-DMA
//     (global_load_lds_dwordx4: fully coalesced 1 KiB pieces, no VGPR staging, no ds_write), double-buffered one line
//     ahead; every MFMA operand is then ONE ds_read_b128 straight in MFMA layout from `slot * row_bytes`, XOR-swizzled so that
//     the 16 rows of a fragment hit 16 distinct bank groups (the DMA applies the same involution on its SOURCE address, the
//     LDS image stays lane-linear as the DMA requires).  A missing neighbour reads a zero row.  No texture-path gather, no
//     ds_bpermute.  The filter image of the step is DMA'd into a 2-slot ring exactly as fragment order needs it.
//     Ranges longer than the staging buffer (CAP rows; 1-2 % of lines) are processed in pieces: slots outside the piece read
//     the zero row, i.e. add exact zeros.
//   Summation order per output element is the gather kernels' (kernel offset ascending, then 32-channel chunk) for
//   Cin <= 64 -> bit-identical results; Cin = 128 runs as two 64-channel halves (half-major), equal within fp32 rounding.
//
// Roofline bookkeeping: per output row and offset the MFMA work is Cin*Cout MACs; staged X traffic is
// 9.05 * Cin * 2 B per row (L2 -> LDS), filter traffic 27 * Cin * Cout * 2 B per block.
#pragma once
#include "spconv_tile.h"
#include "spconv_slab_meta.h"

namespace bevamd {
namespace slab {

using namespace tile;

#ifndef BEVAMD_SLAB_ABL
#define BEVAMD_SLAB_ABL 0   // profiling builds: -DBEVAMD_SLAB_ABL=mask compiles parts of the kernel out (wrong results by design)
#endif
constexpr int SLAB_ABL = BEVAMD_SLAB_ABL;   // 1 no filter DMA, 2 no row DMA, 4 no MFMA, 8 no fragment reads, 16 no barrier, 32 no epilogue stores,
                                            // 64 no slot-table load, 128 no main loop, 256 launch only

struct SlabArgs {
  Args a;                    // features, filter image, epilogue operands (nbr is unused)
  const int2* hdr;           // [nblocks][PLANES] (lo, cnt)
  const uint16_t* slots;     // [nblocks][27][BM]
  unsigned wimg_bytes;       // size of the filter image (buffer descriptor bound)
  unsigned slot_bytes;       // size of the slot table (buffer descriptor bound; narrow-row kernels)
  unsigned long long* prof;  // -DBEVAMD_PROFILING builds: [4] cycle sums over all waves (issue, multiply, dma wait, barrier) + [4] = waves
};

// ---- LDS plan ------------------------------------------------------------------------------------------------------
template <int KC> struct RowSwz;   // XOR applied to the 16-byte piece index of LDS row r (rows are KC*2 bytes)
template <> struct RowSwz<64> {    // 128-byte rows: 8 pieces, consecutive rows rotate through all 8 bank groups
  __device__ __forceinline__ static unsigned of(unsigned r) { return r & 7u; }
};
template <> struct RowSwz<32> {    // 64-byte rows: 4 pieces; rows r and r+4 share banks -> flip the upper pair
  __device__ __forceinline__ static unsigned of(unsigned r) { return (r >> 1) & 2u; }
};

// KC   channels staged per row (32 | 64; CIN / KC passes over the kernel)       SPS  taps per barrier (1 | 3 | 9)
// MT   16-row tiles per wave, NW waves per workgroup (block = NW*16*MT rows)     WR   filter ring slots (2 | 3)
// CAP  rows of one X buffer (ranges longer than that run in pieces)
template <int KC, int CIN, int NT, int MT, int NW, int SPS, int WR, int CAP>
struct Plan {
  static_assert(KC == 32 || KC == 64, "staged row = 32 or 64 channels");
  static_assert(CIN % KC == 0 && TAPS % SPS == 0 && (WR == 2 || WR == 3), "bad split");
  static constexpr int BM = NW * 16 * MT;
  static constexpr int RB = KC * 2;                 // staged bytes per row
  static constexpr int PPR = RB / 16;               // 16-byte pieces per row
  static constexpr int RPI = 64 / PPR;              // rows per DMA instruction (1 KiB)
  static constexpr int CH = KC / 32;                // 32-channel chunks per staged row
  static constexpr int CPB = CIN / 32;              // chunks per kernel offset in the filter image
  static constexpr int NH = CIN / KC;               // channel passes
  static constexpr int GROUPS = TAPS / SPS;         // sync steps per plane
  static constexpr int WD = WR - 1;                 // steps of lookahead of the filter ring
  static constexpr int DX = GROUPS >= 2 ? 1 : WD;   // planes of lookahead of the row staging (never less lead than the filter)
  static constexpr int NXB = DX + 1;                // X buffers
  static constexpr int XB = ((CAP + 1) * RB + 1023) / 1024 * 1024;   // one X buffer incl. the zero row, KiB-aligned
  static constexpr int WS = SPS * CH * NT * 1024;   // filter bytes per sync step
  // DMA roles: the lower half of the waves requests the filter pieces, the upper half the row pieces.  A wave's requests
  // complete in issue order, so a row request (far memory, needed a whole plane later) issued in front of a filter request
  // (L2, needed next step) would have to be waited for first; with separate issuers the rows keep their full lead.
  static constexpr int NWI = NW / 2;                // issuing waves per role
  static constexpr int PW = SPS * CH * NT;          // 1 KiB DMA pieces of one step's filter
  static constexpr int NWS = (PW + NWI - 1) / NWI;  // ... per filter wave (waves past the end issue a dummy piece)
  static constexpr int PX = CAP / RPI;              // 1 KiB DMA pieces of a full X buffer
  static constexpr int NX = (PX + NWI - 1) / NWI;   // ... per row wave: a FIXED count, so that s_waitcnt can count
  static constexpr int OFF_X = 0;
  static constexpr int OFF_W = NXB * XB;
  static constexpr int OFF_SLOT = OFF_W + WR * WS;
  static constexpr int OFF_DUMP = OFF_SLOT + 27 * BM * 2;   // landing zone of the dummy pieces
  static constexpr int BYTES = OFF_DUMP + 1024;
  static_assert(NW * EpiScratch<NT>::U4 * 16 <= NXB * XB, "epilogue scratch must fit the X buffers it aliases");
  static_assert(CAP % RPI == 0, "CAP must be a whole number of DMA instructions");
  static_assert((CAP + 1) * RB < 65536, "row offsets are 16-bit");
  static_assert(NW >= 2 && NW % 2 == 0, "two DMA roles");
  // EXPERIMENTS.md B.17, root cause (round 5): metadata for 64-row blocks is written in the filter-stationary kernels' BAKED format
  // (spconv_slab_meta.h); a kernel that reads raw slots from it is wrong by O(10).  spconv_slabr_kernel decodes it, this one does not.
  static_assert(BM != BAKED_ROWS, "64-row blocks carry baked slot metadata: not decoded by the LDS-filter kernel");
  static_assert(DX * NX < 60 && WD * NWS < 60, "vmcnt is a 6-bit counter");
};

// LDS-DMA, buffer form: 16 bytes per lane from rsrc[voff + soff] to lds + lane*16 (wave-uniform lds / soff)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, void* l) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)l, 16, (int)voff, (int)soff, 0, 0);
}
// at most N of this wave's DMA requests still in flight (they complete in issue order)
template <int N>
__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// workgroup barrier that does NOT drain the DMA queue (hipcc's __syncthreads puts vmcnt(0) in front of s_barrier while an
// LDS-DMA is in flight): LDS reads of this wave are complete (their MFMAs were issued), requests stay in flight across it
__device__ __forceinline__ void barrier_keep_dma() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// one staged range: channel pass h, kernel plane j, piece q of that plane's rows
struct Sub {
  int h, j, q;
  bool done;
};

template <int DT, int KC, int CIN, int NT, int MT, int NW, int SPS, int WR, int CAP>
__global__ __launch_bounds__(NW * 64) void spconv_slab_kernel(SlabArgs sa) {
  typedef Plan<KC, CIN, NT, MT, NW, SPS, WR, CAP> P;
  typedef WaveTile<DT, (CIN > 64 ? 64 : CIN), NT, MT, (CIN > 64 ? 64 : CIN) / 32> WT;   // accumulators + epilogue only
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
  if constexpr (SLAB_ABL & 256) return;   // launch + block map only
  const int tid = threadIdx.x, lane = tid & 63;
  // the wave id IS wave-uniform, but anything derived from threadIdx is divergent to hipcc: without the readfirstlane every
  // LDS-DMA below (uniform LDS base in M0, uniform soffset) is wrapped in a waterfall loop (cdna_hip_programming.md T20)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 15, g4 = lane >> 4;

  // ---- slot table -> LDS; zero rows; block header -> one VGPR pair (lane j = plane j) ------------------------------
  uint16_t* slot = (uint16_t*)(L + P::OFF_SLOT);
  {
    const u32x4* src = (const u32x4*)(sa.slots + (size_t)blk * 27 * P::BM);
    constexpr int N16 = 27 * P::BM * 2 / 16;
    if constexpr (SLAB_ABL & 64) { for (int i = tid; i < N16; i += NW * 64) ((u32x4*)slot)[i] = u32x4{0x00010000u, 0x00030002u, 0x00050004u, 0x00070006u}; }
    else for (int i = tid; i < N16; i += NW * 64) ((u32x4*)slot)[i] = src[i];
    if (tid < P::NXB * P::PPR) {
      const int b = tid / P::PPR, p = tid % P::PPR;
      *(u32x4*)(L + P::OFF_X + b * P::XB + CAP * P::RB + p * 16) = u32x4{0u, 0u, 0u, 0u};
    }
  }
  const int2 hl = sa.hdr[(size_t)blk * PLANES + (lane < PLANES ? lane : 0)];
  const int vlo = hl.x, vcnt = lane < PLANES ? hl.y : 0;
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
  // Lane constants of the row staging: lane -> (row lr of the instruction's RPI rows, physical 16-byte position sp).  The
  // logical piece stored there is sp ^ swz(row) and swz only looks at row bits below RPI, so it is the same for every instruction.
  const unsigned lr = (unsigned)(lane / P::PPR), sp = (unsigned)(lane % P::PPR);
  const unsigned lane_piece_off = (sp ^ RowSwz<KC>::of(lr)) * 16u;
  static_assert(P::RPI % 8 == 0, "swizzle must not depend on the instruction index");
  // rows of piece (j, q), channel pass h -> X buffer xb.  Every wave issues exactly NX 1-KiB requests: pieces past the
  // range re-read its last row (into rows no slot refers to, or into the dump)
  auto stage_x = [&](const Sub& u, int xb) {
    const int n = plane_cnt(u.j) - u.q * CAP;
    const unsigned rows = (unsigned)(n < CAP ? n : CAP);
    const unsigned soff = (unsigned)(plane_lo(u.j) + u.q * CAP) * row_bytes + (unsigned)(u.h * KC * 2);
    char* dst = L + P::OFF_X + xb * P::XB;
#pragma unroll
    for (int t = 0; t < P::NX; ++t) {
      const int i = (w - P::NWI) + t * P::NWI;
      unsigned r = (unsigned)(i * P::RPI) + lr;
      r = r < rows ? r : rows - 1u;
      dma16(rs_x, r * row_bytes + lane_piece_off, soff, i < P::PX ? dst + i * 1024 : dump);
    }
  };
  // filter fragments of the SPS taps of group g of plane j, pass h -> ring slot ws: exactly NWS requests per wave
  auto stage_w = [&](const Sub& u, int g, int ws) {
    char* dst = L + P::OFF_W + ws * P::WS;
    constexpr int PER_TAP = P::CH * NT;   // KiB per tap
#pragma unroll
    for (int t = 0; t < P::NWS; ++t) {
      const int i0 = w + t * P::NWI;
      const int i = i0 < P::PW ? i0 : P::PW - 1;
      const int s = i / PER_TAP, e = i - s * PER_TAP;
      const int k = u.j * TAPS + g * SPS + s;
      dma16(rs_w, (unsigned)lane * 16u, (unsigned)(((k * P::CPB + u.h * P::CH) * NT + e) * 1024), i0 < P::PW ? dst + i * 1024 : dump);
    }
  };
  const bool w_role = w < P::NWI;   // wave-uniform: this wave requests filter pieces (else row pieces)
  // filter waves: all but the filter requests of the `n_w` newest steps have landed
  auto wait_w = [&](int n_w) {
    if (n_w >= 2) wait_dma<2 * P::NWS>();
    else if (n_w == 1) wait_dma<P::NWS>();
    else wait_dma<0>();
  };
  // row waves: all but the row requests of the `n_x` newest pieces have landed
  auto wait_x = [&](int n_x) {
    if (n_x >= 2) wait_dma<2 * P::NX>();
    else if (n_x == 1) wait_dma<P::NX>();
    else wait_dma<0>();
  };

  WT wt;
  wt.init(a, blk * P::BM + w * 16 * MT, m, nullptr, (u32x4*)(L + P::OFF_X) + w * EpiScratch<NT>::U4);

  // multiply the SPS taps of group g of piece `u` from X buffer xb and ring slot ws.  The reduction is a flat list of
  // SPS * CH units (tap, 32-channel chunk); the fragments of unit i+1 (NT filter + MT row fragments, all ds_read_b128) are
  // requested BEFORE the MT*NT MFMAs of unit i are issued (two register sets, sched_barrier keeps hipcc from re-serialising
  // them behind one s_waitcnt): the LDS round trip of one unit hides under the MFMAs of the previous one.
  auto multiply = [&](const Sub& u, int g, int xb, int ws) {
    const char* X = L + P::OFF_X + xb * P::XB;
    const u32x4* Wl = (const u32x4*)(L + P::OFF_W + ws * P::WS);
    const unsigned pbase = (unsigned)(u.q * CAP);
    const unsigned plive = (unsigned)plane_cnt(u.j) - pbase;
    const unsigned prow = plive < (unsigned)CAP ? plive : (unsigned)CAP;   // rows of this piece
    constexpr int U = SPS * P::CH;
    const uint16_t* sl = slot + (u.j * TAPS + g * SPS) * P::BM + w * 16 * MT + c;
    unsigned xo[SPS][MT];   // byte address of piece 0 of the row | swizzle in the top bits
#pragma unroll
    for (int s = 0; s < SPS; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        unsigned e = (unsigned)sl[s * P::BM + mt * 16] - pbase;   // NO_SLOT - pbase stays >= prow
        e = e < prow ? e : (unsigned)CAP;                         // outside the piece: the zero row
        xo[s][mt] = e * P::RB + (RowSwz<KC>::of(e) << 20);
      }
    auto fetch = [&](int i, u32x4 (&b)[NT], u32x4 (&x)[MT]) {
      const int s = i / P::CH, cc = i % P::CH;
      if constexpr (SLAB_ABL & 8) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = u32x4{(unsigned)(i + nt), 0x3C003C00u, 0x3C003C00u, (unsigned)lane};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) x[mt] = u32x4{xo[s][mt], 0x3C003C00u, (unsigned)cc, (unsigned)lane};
        return;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) b[nt] = Wl[(i * NT + nt) * 64 + lane];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        x[mt] = *(const u32x4*)(X + (xo[s][mt] & 0xFFFFFu) + (((unsigned)(cc * 4 + g4)) ^ (xo[s][mt] >> 20)) * 16);
    };
    auto mma = [&](const u32x4 (&b)[NT], const u32x4 (&x)[MT]) {
      if constexpr (SLAB_ABL & 4) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) wt.acc[mt][nt][0] += __uint_as_float(b[nt].x ^ x[mt].x);
        return;
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) wt.acc[mt][nt] = mfma<DT>(b[nt], x[mt], wt.acc[mt][nt]);
    };
    u32x4 b0[NT], x0[MT], b1[NT], x1[MT];
    fetch(0, b0, x0);
#pragma unroll
    for (int i = 0; i < U; i += 2) {
      if (i + 1 < U) fetch(i + 1, b1, x1);
      __builtin_amdgcn_sched_barrier(0);
      mma(b0, x0);
      __builtin_amdgcn_sched_barrier(0);
      if (i + 1 < U) {
        if (i + 2 < U) fetch(i + 2, b0, x0);
        __builtin_amdgcn_sched_barrier(0);
        mma(b1, x1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  // Pipeline.  Steps are numbered t = 0, 1, ...; step t reads filter ring slot t % WR and the X buffer of its piece.
  // While step t multiplies, the filter of step t+WD and (at the first step of a piece) the rows of the piece DX planes ahead
  // are in flight; each wave's requests complete in issue order and every wave issues the same fixed number per step, so
  // "everything step t+1 needs has landed" is a COUNTED wait at the end of step t, never a full drain.
  Sub sub[3];   // this piece, the next, the one after
  sub[0] = Sub{0, next_plane(0), 0, false};   // the centre plane always has rows in a live block
  sub[1] = next_sub(sub[0]);
  sub[2] = next_sub(sub[1]);
  auto step_after = [&](int g, int d, int& off, int& g2) { const int gg = g + d; off = gg / P::GROUPS; g2 = gg % P::GROUPS; };
  if (!w_role) {
    stage_x(sub[0], 0);
    if (P::DX == 2 && !sub[1].done) stage_x(sub[1], 1);
  } else {
#pragma unroll
    for (int d = 0; d < P::WD; ++d) {
      int off, gd;
      step_after(0, d, off, gd);
      if (!sub[off].done) stage_w(sub[off], gd, d);
    }
  }
  wait_dma<0>();
  __syncthreads();   // also publishes the slot table and the zero rows
  int xb = 0, ws = 0;
#ifdef BEVAMD_PROFILING
  unsigned long long t_issue = 0, t_mul = 0, t_wait = 0, t_bar = 0, t0, t1;
  const bool timers = sa.prof != nullptr;   // wave-uniform: without a buffer the ablation runs are not perturbed by s_memtime
#define BEVAMD_TICK(acc) do { if (timers) { t1 = __builtin_readcyclecounter(); acc += t1 - t0; t0 = t1; } } while (0)
  t0 = __builtin_readcyclecounter();
#else
#define BEVAMD_TICK(acc) do { } while (0)
#endif
  for (;;) {
    if constexpr (SLAB_ABL & 128) break;   // prologue + epilogue only
#pragma unroll
    for (int g = 0; g < P::GROUPS; ++g) {
      // requests: filter waves ask for the filter of step t + WD; row waves ask — at the first step of a piece — for the rows
      // of the piece DX planes ahead
      int off, gd;
      step_after(g, P::WD, off, gd);
      constexpr bool do_w = !(SLAB_ABL & 1), do_x = !(SLAB_ABL & 2);
      if (w_role) {
        if (do_w && !sub[off].done) stage_w(sub[off], gd, (ws + P::WD) % WR);
      } else if (do_x && g == 0 && !sub[P::DX].done) {
        stage_x(sub[P::DX], (xb + P::DX) % P::NXB);
      }
      BEVAMD_TICK(t_issue);
      multiply(sub[0], g, xb, ws);
      BEVAMD_TICK(t_mul);
      // Needed by step t+1: its filter (filter waves: everything but the requests of steps t+2 .. t+WD) and, if step t+1 opens a
      // piece, that piece's rows (row waves: everything but the requests for the pieces after it).
      if (w_role) {
        int n_w = 0;
#pragma unroll
        for (int d = 2; d <= P::WD; ++d) {
          int o2, g2;
          step_after(g, d, o2, g2);
          n_w += !sub[o2].done;
        }
        wait_w(n_w);
      } else if (g == P::GROUPS - 1) {
        int n_x = 0;
#pragma unroll
        for (int d = 2; d <= P::DX; ++d) n_x += !sub[d].done;
        wait_x(n_x);
      }
      BEVAMD_TICK(t_wait);
      if constexpr (!(SLAB_ABL & 16)) barrier_keep_dma();
      BEVAMD_TICK(t_bar);
      ws = ws + 1 == WR ? 0 : ws + 1;
    }
    if (sub[1].done) break;
    sub[0] = sub[1];
    sub[1] = sub[2];
    sub[2] = next_sub(sub[2]);
    xb = xb + 1 == P::NXB ? 0 : xb + 1;
  }
#ifdef BEVAMD_PROFILING
  if (sa.prof && lane == 0) {
    atomicAdd(sa.prof + 0, t_issue); atomicAdd(sa.prof + 1, t_mul); atomicAdd(sa.prof + 2, t_wait); atomicAdd(sa.prof + 3, t_bar);
    atomicAdd(sa.prof + 4, 1ull);
  }
#endif
#undef BEVAMD_TICK
  if constexpr (SLAB_ABL & 32) {   // keep the accumulators alive with (at most) one store per wave
    float t = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) t += wt.acc[mt][nt][0] + wt.acc[mt][nt][1] + wt.acc[mt][nt][2] + wt.acc[mt][nt][3];
    if (t == 12345.678f) ((float*)a.out)[0] = t;
    return;
  }
  wt.store(a);   // epilogue scratch aliases the X buffers: every wave passed the last barrier, nobody reads X any more
}

// ---- host side -------------------------------------------------------------------------------------------------------
int launch_f16(const SlabArgs& sa, int cin, int nt, int variant, hipStream_t stream);
int launch_bf16(const SlabArgs& sa, int cin, int nt, int variant, hipStream_t stream);

}  // namespace slab
}  // namespace bevamd
