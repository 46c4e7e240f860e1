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

namespace bevamd {
namespace slab {

using namespace tile;

constexpr int LINES = 9;          // (kx, ky) pairs of the 3x3x3 kernel
constexpr int TAPS = 3;           // kz taps per line
constexpr unsigned NO_SLOT = 0xFFFFu;

struct SlabArgs {
  Args a;                    // features, filter image, epilogue operands (nbr is unused)
  const int2* hdr;           // [nblocks][LINES] (lo, cnt)
  const uint16_t* slots;     // [nblocks][27][BM]
};

// ---- metadata -----------------------------------------------------------------------------------------------------
// One workgroup (BM threads) per block of BM output rows.
template <int BM>
__global__ __launch_bounds__(BM) void slab_build_kernel(const int* __restrict__ nbr, int nbr_stride, int m_cap,
                                                        const int* __restrict__ m_dev, int2* __restrict__ hdr,
                                                        uint16_t* __restrict__ slots, int* __restrict__ status) {
  __shared__ int s_lo[BM / 64][LINES], s_hi[BM / 64][LINES];
  int m = m_dev ? *m_dev : m_cap;
  if (m > m_cap) m = m_cap;
  const int blk = blockIdx.x, t = threadIdx.x, row = blk * BM + t;
  const bool live = row < m;
  int v[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) v[k] = live ? nbr[(size_t)k * nbr_stride + row] : -1;
  const int w = t >> 6;
#pragma unroll
  for (int j = 0; j < LINES; ++j) {
    int lo = 0x7FFFFFFF, hi = -1;
#pragma unroll
    for (int d = 0; d < TAPS; ++d) {
      const int x = v[j * TAPS + d];
      if (x >= 0) { lo = x < lo ? x : lo; hi = x > hi ? x : hi; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const int l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
    }
    if ((t & 63) == 0) { s_lo[w][j] = lo; s_hi[w][j] = hi; }
  }
  __syncthreads();
  bool overflow = false;
#pragma unroll
  for (int j = 0; j < LINES; ++j) {
    int lo = 0x7FFFFFFF, hi = -1;
#pragma unroll
    for (int i = 0; i < BM / 64; ++i) { lo = s_lo[i][j] < lo ? s_lo[i][j] : lo; hi = s_hi[i][j] > hi ? s_hi[i][j] : hi; }
    int cnt = hi >= 0 ? hi - lo + 1 : 0;
    if (hi < 0) lo = 0;
    if (cnt > 0xFFFE) { cnt = 0xFFFE; overflow = true; }   // cannot happen for rows in linear-index order on grids the host admits
    if (t == 0) hdr[(size_t)blk * LINES + j] = make_int2(lo, cnt);
#pragma unroll
    for (int d = 0; d < TAPS; ++d) {
      const int k = j * TAPS + d, x = v[k];
      unsigned s = NO_SLOT;
      if (x >= 0 && x - lo < cnt) s = (unsigned)(x - lo);
      slots[((size_t)blk * 27 + k) * BM + t] = (uint16_t)s;
    }
  }
  if (overflow && t == 0 && status) atomicOr(status, 1);
}

// ---- LDS plan ------------------------------------------------------------------------------------------------------
template <int KC> struct RowSwz;   // XOR applied to the 16-byte piece index of LDS row r (rows are KC*2 bytes)
template <> struct RowSwz<64> {    // 128-byte rows: 8 pieces, consecutive rows rotate through all 8 bank groups
  __device__ __forceinline__ static unsigned of(unsigned r) { return r & 7u; }
};
template <> struct RowSwz<32> {    // 64-byte rows: 4 pieces; rows r and r+4 share banks -> flip the upper pair
  __device__ __forceinline__ static unsigned of(unsigned r) { return (r >> 1) & 2u; }
};

template <int KC, int CIN, int NT, int MT, int NW, int SPS, int CAP>
struct Plan {
  static_assert(KC == 32 || KC == 64, "staged row = 32 or 64 channels");
  static_assert(CIN % KC == 0 && TAPS % SPS == 0, "bad split");
  static constexpr int BM = NW * 16 * MT;
  static constexpr int RB = KC * 2;                 // staged bytes per row
  static constexpr int PPR = RB / 16;               // 16-byte pieces per row
  static constexpr int RPI = 64 / PPR;              // rows per DMA instruction (1 KiB)
  static constexpr int CH = KC / 32;                // 32-channel chunks per staged row
  static constexpr int CPB = CIN / 32;              // chunks per kernel offset in the filter image
  static constexpr int NH = CIN / KC;               // channel halves
  static constexpr int GROUPS = TAPS / SPS;         // sync steps per line
  static constexpr int XB = ((CAP + 1) * RB + 1023) / 1024 * 1024;   // one X buffer incl. the zero row, KiB-aligned
  static constexpr int WS = SPS * CH * NT * 1024;   // filter bytes per sync step
  static constexpr int OFF_X = 0;
  static constexpr int OFF_W = 2 * XB;
  static constexpr int OFF_SLOT = OFF_W + 2 * WS;
  static constexpr int BYTES = OFF_SLOT + 27 * BM * 2;
  static_assert(NW * EpiScratch<NT>::U4 * 16 <= 2 * XB, "epilogue scratch must fit the X buffers it aliases");
  static_assert(CAP % RPI == 0, "CAP must be a whole number of DMA instructions");
};

template <typename T>
__device__ __forceinline__ void glds16(const T* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// one staged range: channel half h, kernel line j, piece q of that line's rows
struct Sub {
  int h, j, q;
  bool done;
};

template <int DT, int KC, int CIN, int NT, int MT, int NW, int SPS, int CAP>
__global__ __launch_bounds__(NW * 64) void spconv_slab_kernel(SlabArgs sa) {
  typedef Plan<KC, CIN, NT, MT, NW, SPS, CAP> P;
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
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, g4 = lane >> 4;

  // ---- slot table -> LDS; zero rows; block header -> one VGPR pair (lane j = line j) -------------------------------
  uint16_t* slot = (uint16_t*)(L + P::OFF_SLOT);
  {
    const u32x4* src = (const u32x4*)(sa.slots + (size_t)blk * 27 * P::BM);
    constexpr int N16 = 27 * P::BM * 2 / 16;
    for (int i = tid; i < N16; i += NW * 64) ((u32x4*)slot)[i] = src[i];
    if (tid < 2 * P::PPR) {
      const int b = tid / P::PPR, p = tid % P::PPR;
      *(u32x4*)(L + P::OFF_X + b * P::XB + CAP * P::RB + p * 16) = u32x4{0u, 0u, 0u, 0u};
    }
  }
  const int2 hl = sa.hdr[(size_t)blk * LINES + (lane < LINES ? lane : 0)];
  const int vlo = hl.x, vcnt = lane < LINES ? hl.y : 0;
  const unsigned live_lines = (unsigned)__builtin_amdgcn_readfirstlane((int)__ballot(vcnt > 0));   // bit j: line j has rows
  auto line_lo = [&](int j) { return __builtin_amdgcn_readlane(vlo, j); };
  auto line_cnt = [&](int j) { return __builtin_amdgcn_readlane(vcnt, j); };
  auto next_line = [&](int from) {   // first line >= from with rows, LINES if none
    const unsigned rest = from < LINES ? live_lines >> from : 0u;
    return rest ? from + (int)__builtin_ctz(rest) : LINES;
  };
  auto next_sub = [&](Sub u) {
    if ((u.q + 1) * CAP < line_cnt(u.j)) { ++u.q; return u; }
    u.q = 0;
    u.j = next_line(u.j + 1);
    if (u.j < LINES) return u;
    u.j = next_line(0);
    if (++u.h >= P::NH) u.done = true;
    return u;
  };

  const typename Num<DT>::T* feat = (const typename Num<DT>::T*)a.feat;
  const size_t row_elems = (size_t)a.feat_stride;
  // rows of piece (j, q), channel half h -> X buffer xb: 1 KiB DMA instructions dealt round-robin to the waves
  auto stage_x = [&](const Sub& u, int xb) {
    const int n = line_cnt(u.j) - u.q * CAP;
    const int rows = n < CAP ? n : CAP;
    const int base = line_lo(u.j) + u.q * CAP;
    const int ninstr = (rows + P::RPI - 1) / P::RPI;
    char* dst = L + P::OFF_X + xb * P::XB;
    for (int i = w; i < ninstr; i += NW) {
      unsigned r = (unsigned)(i * P::RPI + lane / P::PPR);
      const unsigned sp = (unsigned)(lane % P::PPR);
      const unsigned p = sp ^ RowSwz<KC>::of(r);          // logical piece stored at physical position sp of LDS row r
      r = r < (unsigned)rows ? r : (unsigned)(rows - 1);   // tail lanes re-read the last row (never referenced by a slot)
      glds16(feat + (size_t)(base + (int)r) * row_elems + u.h * KC + p * 8, dst + i * 1024);
    }
  };
  // filter fragments of the SPS taps of group g of line j, half h -> ring slot wb (contiguous CH * NT KiB per tap)
  auto stage_w = [&](const Sub& u, int g, int wb) {
    char* dst = L + P::OFF_W + wb * P::WS;
    constexpr int PER_TAP = P::CH * NT;   // KiB per tap
    for (int i = w; i < SPS * PER_TAP; i += NW) {
      const int s = i / PER_TAP, e = i - s * PER_TAP;
      const int k = u.j * TAPS + g * SPS + s;
      const char* src = (const char*)a.wimg + ((size_t)(k * P::CPB + u.h * P::CH) * NT + e) * 1024;
      glds16(src + lane * 16, dst + i * 1024);
    }
  };

  WT wt;
  wt.init(a, blk * P::BM + w * 16 * MT, m, nullptr, (u32x4*)(L + P::OFF_X) + w * EpiScratch<NT>::U4);

  // multiply the SPS taps of group g of piece `u` from X buffer xb and ring slot wb
  auto multiply = [&](const Sub& u, int g, int xb, int wb) {
    const char* X = L + P::OFF_X + xb * P::XB;
    const u32x4* Wl = (const u32x4*)(L + P::OFF_W + wb * P::WS);
    const unsigned pbase = (unsigned)(u.q * CAP);
    const unsigned plive = (unsigned)line_cnt(u.j) - pbase;
    const unsigned prow = plive < (unsigned)CAP ? plive : (unsigned)CAP;   // rows of this piece
#pragma unroll
    for (int s = 0; s < SPS; ++s) {
      const int k = u.j * TAPS + g * SPS + s;
      unsigned xo[MT], xs[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        unsigned e = (unsigned)slot[k * P::BM + w * 16 * MT + mt * 16 + c] - pbase;   // NO_SLOT - pbase stays >= prow
        e = e < prow ? e : (unsigned)CAP;                                             // outside the piece: the zero row
        xo[mt] = e * P::RB;
        xs[mt] = RowSwz<KC>::of(e);
      }
#pragma unroll
      for (int cc = 0; cc < P::CH; ++cc) {
        u32x4 b[NT], x[MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = Wl[((s * P::CH + cc) * NT + nt) * 64 + lane];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          x[mt] = *(const u32x4*)(X + xo[mt] + (((unsigned)(cc * 4 + g4)) ^ xs[mt]) * 16);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) wt.acc[mt][nt] = mfma<DT>(b[nt], x[mt], wt.acc[mt][nt]);
      }
    }
  };

  Sub cur{0, next_line(0), 0, false};   // the centre line always has rows in a live block
  Sub nxt = next_sub(cur);
  stage_x(cur, 0);
  stage_w(cur, 0, 0);
  int xb = 0, wb = 0;
  __syncthreads();   // (hipcc drains the DMA queue — vmcnt(0) — in front of the barrier; also publishes the slot table)
  for (;;) {
#pragma unroll
    for (int g = 0; g < P::GROUPS; ++g) {
      // requests for the next step go out first: the filter of the next group (or of the next piece's first group), and —
      // at the first group of a piece — the rows of the next piece; they land while this step multiplies
      if (g + 1 < P::GROUPS) stage_w(cur, g + 1, wb ^ 1);
      else if (!nxt.done) stage_w(nxt, 0, wb ^ 1);
      if (g == 0 && !nxt.done) stage_x(nxt, xb ^ 1);
      multiply(cur, g, xb, wb);
      __syncthreads();   // the requests landed (vmcnt(0) in front of the barrier); everyone is done with xb / wb
      wb ^= 1;
    }
    if (nxt.done) break;
    cur = nxt;
    nxt = next_sub(cur);
    xb ^= 1;
  }
  wt.store(a);   // epilogue scratch aliases the X buffers: every wave passed the last barrier, nobody reads X any more
}

// ---- host side -------------------------------------------------------------------------------------------------------
int launch_f16(const SlabArgs& sa, int cin, int nt, int variant, hipStream_t stream);
int launch_bf16(const SlabArgs& sa, int cin, int nt, int variant, hipStream_t stream);

}  // namespace slab
}  // namespace bevamd
