// Block metadata of the staged-rows ("slab") submanifold convolution (spconv_slab.h): per block of BM consecutive output rows
// and kernel plane kx, the contiguous range of input rows its nine (ky, kz) taps read, and the neighbour table as 16-bit
// slots relative to the range start.  Two producers share the emitter: from a neighbour table (any row order the caller
// vouches for), or straight from the rank index of a set in ascending linear order (spconv_indice.hip) — no table at all.
#pragma once
#include "common.h"

namespace bevamd {
namespace slab {

constexpr int PLANES = 3;         // kx planes of the 3x3x3 kernel: the 9 (ky, kz) taps of a plane share one staged range
constexpr int TAPS = 9;           // taps per plane
constexpr unsigned NO_SLOT = 0xFFFFu;

// BAKED metadata (64-row blocks, the filter-stationary kernels of spconv_slab_fstat.h): a slot is stored as the LDS byte
// offset the kernel reads it from — staged row s of the range lives in LDS row e = s + 1 of 64 bytes (row 0 is the zero row a
// missing neighbour reads), with the kernel's bank swizzle already folded in: entry = e*64 | ((e >> 2) & 3) << 4, 0 = none.  The
// address of a fragment is then ONE v_xor of the entry with the lane's 16-byte piece (measured on the 32-channel layers:
// 12 VALU instructions per slot of index arithmetic before, 12 % of the layer).  A range too long for 16 bits of bytes
// (> 1022 rows) keeps raw slots and says so in the header (HDR_RAW in the row count); the kernel then takes its general path.
constexpr unsigned HDR_RAW = 0x40000000u;
constexpr int BAKED_ROWS = 64;        // block size that implies the baked format
constexpr int BAKED_ROW_BYTES = 64;   // 32 channels of 16 bits
__host__ __device__ __forceinline__ unsigned baked_entry(unsigned e) { return e * BAKED_ROW_BYTES | (((e >> 2) & 3u) << 4); }

// Round 5: the same idea for the 128-byte staged rows (64 channels) of the register-filter kernels (spconv_slab_regw.h, flag
// F_BAKED): entry = r*128 | (r & 7) << 4 with r = slot + 1 the LDS row (row 0 = the zero row), 0 = none; the address of a fragment
// is (entry ^ piece*16) + buffer base = one v_xad_u32.  Requested with a FORMAT CODE in the upper half of `block_rows`
// (block_rows = rows | FMT_BAKED128 << 16: what bevamd_spconv_slab_block_rows returns for such variants); ranges of more than 510
// rows keep raw slots and say so in the header (HDR_RAW), like the 64-byte format.
constexpr int FMT_SHIFT = 16;
constexpr int FMT_BAKED128 = 1;
__host__ __device__ __forceinline__ int rows_of_code(int code) { return code & 0xFFFF; }
__host__ __device__ __forceinline__ int fmt_of_code(int code) { return (code >> FMT_SHIFT) & 0xFF; }
__host__ __device__ __forceinline__ unsigned baked128_entry(unsigned r) { return r * 128u | ((r & 7u) << 4); }

// Round 6: the formats of the staged-rows FILTER GRADIENT (spconv_wgrad_slab.h; 128-row blocks only).  The reduction index of that
// GEMM is the row, and a lane of its transposing LDS reads wants the staged rows of neighbours {32 q + 8 g + cc + 4 h} (q < 4, h < 2)
// of an offset: the table of a block is stored TRANSPOSED — entry (offset k, row r) at ((k*4 + g)*4 + cc)*8 + q*2 + h with
// r = 32 q + 8 g + 4 h + cc — so that those eight entries are one 16-byte read, and an entry is the LDS byte offset of the staged
// row inside a stage of the kernel: plane * WG_CAP * row bytes + slot * row bytes (+ the half swap of 64-byte rows: bit 3 of the
// slot -> bit 5), or WG_ZERO (the zero row behind the three planes) for a missing neighbour.  FMT_WG64: 64-byte staged rows
// (32 input channels per workgroup: 32 channels and wider), 192 rows per plane; FMT_WG32: 32-byte rows (16 channels), 256 rows per
// plane.  A plane whose range is longer keeps RAW slots — in the same transposed positions — and says so in its header (HDR_RAW):
// the kernel converts those itself, piece by piece.
constexpr int FMT_WG64 = 2, FMT_WG32 = 3;
__host__ __device__ constexpr bool fmt_is_wg(int fmt) { return fmt == FMT_WG64 || fmt == FMT_WG32; }
__host__ __device__ constexpr int wg_row_bytes(int fmt) { return fmt == FMT_WG64 ? 64 : 32; }
__host__ __device__ constexpr int wg_cap(int fmt) { return fmt == FMT_WG64 ? 192 : 256; }
__host__ __device__ constexpr unsigned wg_zero(int fmt) { return 3u * (unsigned)wg_cap(fmt) * (unsigned)wg_row_bytes(fmt); }
__host__ __device__ __forceinline__ unsigned wg_entry(int fmt, int plane, unsigned slot) {
  const unsigned rb = (unsigned)wg_row_bytes(fmt);
  return (unsigned)plane * (unsigned)wg_cap(fmt) * rb + slot * rb + (fmt == FMT_WG64 ? ((slot >> 3) & 1u) << 5 : 0u);
}
__host__ __device__ __forceinline__ int wg_index(int k, int r) {   // position of (offset k, row r) inside a block's table
  return ((k * 4 + ((r >> 3) & 3)) * 4 + (r & 3)) * 8 + (r >> 5) * 2 + ((r >> 2) & 1);
}

// One workgroup of BM threads per block; thread t holds the 27 neighbour rows v[] of output row blk*BM + t (-1 = none).
template <int BM>
__device__ __forceinline__ void slab_emit(const int (&v)[27], int blk, int t, int2* __restrict__ hdr,
                                          uint16_t* __restrict__ slots, int* __restrict__ status, int fmt = 0) {
  __shared__ int s_lo[BM / 64][PLANES], s_hi[BM / 64][PLANES];
  const int w = t >> 6;
#pragma unroll
  for (int j = 0; j < PLANES; ++j) {
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
  for (int j = 0; j < PLANES; ++j) {
    int lo = 0x7FFFFFFF, hi = -1;
#pragma unroll
    for (int i = 0; i < BM / 64; ++i) { lo = s_lo[i][j] < lo ? s_lo[i][j] : lo; hi = s_hi[i][j] > hi ? s_hi[i][j] : hi; }
    int cnt = hi >= 0 ? hi - lo + 1 : 0;
    if (hi < 0) lo = 0;
    if (cnt > 0xFFFE) { cnt = 0xFFFE; overflow = true; }   // cannot happen for rows in linear-index order on grids the host admits
    const bool wants128 = fmt == FMT_BAKED128;                 // wave-uniform (kernel argument)
    const bool wg = BM == 128 && fmt_is_wg(fmt);
    const bool wants = wants128 || wg || BM == BAKED_ROWS;
    const bool baked = wg ? cnt <= wg_cap(fmt) : wants128 ? (cnt + 1) * 128 <= 0xFFFF : (BM == BAKED_ROWS && (cnt + 1) * BAKED_ROW_BYTES <= 0xFFFF);
    if (t == 0) hdr[(size_t)blk * PLANES + j] = make_int2(lo, wants && !baked ? (int)((unsigned)cnt | HDR_RAW) : cnt);
#pragma unroll
    for (int d = 0; d < TAPS; ++d) {
      const int k = j * TAPS + d, x = v[k];
      unsigned s = !baked ? NO_SLOT : wg ? wg_zero(fmt) : 0u;
      if (x >= 0 && x - lo < cnt) {
        const unsigned r = (unsigned)(x - lo) + 1u;
        s = !baked ? r - 1u : wg ? wg_entry(fmt, j, r - 1u) : wants128 ? baked128_entry(r) : baked_entry(r);
      }
      slots[(size_t)blk * 27 * BM + (wg ? wg_index(k, t) : k * BM + t)] = (uint16_t)s;
    }
  }
  if (overflow && t == 0 && status) atomicOr(status, 1);
}

// from a neighbour table nbr [27, nbr_stride]
template <int BM>
__global__ __launch_bounds__(BM) void slab_build_kernel(const int* __restrict__ nbr, int nbr_stride, int m_cap,
                                                        const int* __restrict__ m_dev, int2* __restrict__ hdr,
                                                        uint16_t* __restrict__ slots, int* __restrict__ status, int fmt) {
  int m = m_dev ? *m_dev : m_cap;
  if (m > m_cap) m = m_cap;
  const int blk = blockIdx.x, t = threadIdx.x, row = blk * BM + t;
  if (blk * BM >= m) return;   // capacity-sized launch: nobody reads a dead block's metadata
  const bool live = row < m;
  int v[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) v[k] = live ? nbr[(size_t)k * nbr_stride + row] : -1;
  slab_emit<BM>(v, blk, t, hdr, slots, status, fmt);
}

}  // namespace slab
}  // namespace bevamd
