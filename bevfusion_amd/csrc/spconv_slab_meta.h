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

// One workgroup of BM threads per block; thread t holds the 27 neighbour rows v[] of output row blk*BM + t (-1 = none).
template <int BM>
__device__ __forceinline__ void slab_emit(const int (&v)[27], int blk, int t, int2* __restrict__ hdr,
                                          uint16_t* __restrict__ slots, int* __restrict__ status) {
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
    if (t == 0) hdr[(size_t)blk * PLANES + j] = make_int2(lo, cnt);
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

// from a neighbour table nbr [27, nbr_stride]
template <int BM>
__global__ __launch_bounds__(BM) void slab_build_kernel(const int* __restrict__ nbr, int nbr_stride, int m_cap,
                                                        const int* __restrict__ m_dev, int2* __restrict__ hdr,
                                                        uint16_t* __restrict__ slots, int* __restrict__ status) {
  int m = m_dev ? *m_dev : m_cap;
  if (m > m_cap) m = m_cap;
  const int blk = blockIdx.x, t = threadIdx.x, row = blk * BM + t;
  if (blk * BM >= m) return;   // capacity-sized launch: nobody reads a dead block's metadata
  const bool live = row < m;
  int v[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) v[k] = live ? nbr[(size_t)k * nbr_stride + row] : -1;
  slab_emit<BM>(v, blk, t, hdr, slots, status);
}

}  // namespace slab
}  // namespace bevamd
