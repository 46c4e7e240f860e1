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
    const bool wants = wants128 || BM == BAKED_ROWS;
    const bool baked = wants128 ? (cnt + 1) * 128 <= 0xFFFF : (BM == BAKED_ROWS && (cnt + 1) * BAKED_ROW_BYTES <= 0xFFFF);
    if (t == 0) hdr[(size_t)blk * PLANES + j] = make_int2(lo, wants && !baked ? (int)((unsigned)cnt | HDR_RAW) : cnt);
#pragma unroll
    for (int d = 0; d < TAPS; ++d) {
      const int k = j * TAPS + d, x = v[k];
      unsigned s = baked ? 0u : NO_SLOT;
      if (x >= 0 && x - lo < cnt) {
        const unsigned r = (unsigned)(x - lo) + 1u;
        s = !baked ? r - 1u : wants128 ? baked128_entry(r) : baked_entry(r);
      }
      slots[((size_t)blk * 27 + k) * BM + t] = (uint16_t)s;
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
