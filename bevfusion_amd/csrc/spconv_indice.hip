// spconv rulebook ("indice pairs") construction for gfx950.
//
// Replaces (reference, /root/reference/mmdet3d/ops/spconv):
//   include/spconv/spconv_ops.h:27-141      getIndicePair<3>  (allocates a dense int32 grid of
//                                            B*X*Y*Z entries per call: 340 MB at 1440x1440x41)
//   include/spconv/indice.cu.h:147-203      prepareSubMGridKernel / getSubMIndicePairsKernel
//   include/spconv/indice.cu.h:22-65,112-145 prepareIndicePairsKernel / assignGridAndIndiceOut /
//                                            assignIndicePairs (+ torch::_unique, spconv_ops.h:130)
//   include/spconv/geometry.h:24-85         getValidOutPos (offset numbering)
//
// Native formulation (integer only; hash probes hit L2, nothing here is HBM- or MFMA-bound):
//   * active input voxels go into an open-addressing hash table (key = linear index, 2N slots,
//     ~2.5 MB for 160 k voxels) instead of a dense grid that must be filled with -1 every call;
//   * the rulebook is OUTPUT-STATIONARY: nbr[k][o] = input row feeding output row o through kernel
//     offset k (or -1).  The fused convolution walks it without scatter-add or atomics;
//   * strided conv: every input sets the bits of the <= prod(ceil(k/s)) output cells it touches in a
//     bitmap over the output grid (1 bit per cell: 1.4 MB at 720x720x21); a popcount prefix over the
//     bitmap words numbers the active outputs in ascending linear index — exactly the row order of the
//     reference's CUDA path (torch::_unique, spconv_ops.h:130) — with no sort.  The (bits, prefix)
//     words are kept as a RANK INDEX of the output set: "which row sits at cell c" is one 8-byte load
//     + popcount, which the next layers use instead of building a hash;
//   * every count may stay on the device (n_dev / num_out_dev): nothing here needs a host sync;
//   * the reference's (indice_pairs [K,2,N], indice_num [K]) arrays are derived from nbr by a
//     per-offset stream compaction (deterministic: pairs ordered by output row; the reference's CUDA
//     order is atomicAdd order, i.e. unspecified).
// Offset numbering: offset = (kx*Ky + ky)*Kz + kz with k = in - out*stride + pad (geometry.h:67-69).
#include "common.h"
#include "spconv_slab_meta.h"

namespace bevamd {

constexpr uint32_t HASH_EMPTY = 0xFFFFFFFFu;

struct ConvGeom {
  int in_shape[3], out_shape[3], ksize[3], stride[3], pad[3], dil[3];
  int batch, K;
  int transpose;  // 1: out = in*stride - pad + k*dil (geometry.h:86-141) instead of in = out*stride - pad + k*dil
};

// The two coordinate maps of one axis.  `scatter` runs the relation  dst*stride - pad + k*dil == src  backwards (dst from
// src: the division must be exact), `reach` runs it forwards.  A regular convolution reaches inputs from outputs and
// scatters inputs to outputs; a transposed one does the opposite.
__device__ __forceinline__ bool axis_scatter(const ConvGeom& g, int a, int src, int k, int dst_size, int& dst) {
  const int t = src + g.pad[a] - k * g.dil[a];
  if (t < 0 || t % g.stride[a]) return false;
  dst = t / g.stride[a];
  return dst < dst_size;
}
__device__ __forceinline__ bool axis_reach(const ConvGeom& g, int a, int src, int k, int dst_size, int& dst) {
  dst = src * g.stride[a] - g.pad[a] + k * g.dil[a];
  return dst >= 0 && dst < dst_size;
}
// output cell fed by input coordinate `in` through kernel tap k, on axis a.  GENERAL = false is the undilated regular
// convolution (every layer of the SparseEncoder): no dilation multiply, no transposed branch.
template <bool GENERAL>
__device__ __forceinline__ bool axis_in_to_out(const ConvGeom& g, int a, int in, int k, int& out) {
  if constexpr (!GENERAL) {
    const int t = in + g.pad[a] - k;
    if (t < 0 || t % g.stride[a]) return false;
    out = t / g.stride[a];
    return out < g.out_shape[a];
  } else {
    return g.transpose ? axis_reach(g, a, in, k, g.out_shape[a], out) : axis_scatter(g, a, in, k, g.out_shape[a], out);
  }
}
// input cell read by output coordinate `out` through kernel tap k, on axis a
__device__ __forceinline__ bool axis_out_to_in(const ConvGeom& g, int a, int out, int k, int& in) {
  return g.transpose ? axis_scatter(g, a, out, k, g.in_shape[a], in) : axis_reach(g, a, out, k, g.in_shape[a], in);
}

__device__ __forceinline__ uint32_t hash_u32(uint32_t k) {
  k ^= k >> 16; k *= 0x7feb352dU; k ^= k >> 15; k *= 0x846ca68bU; k ^= k >> 16;
  return k;
}

// Home slot of a key: sample b's keys start in region b of the table (`region` slots each, a power of two; probing still
// wraps over the whole table, so an over-full region spills into the next one instead of failing).  The rows of one
// sample are consecutive, so the workgroups looking up its neighbours touch 1/B of the table at a time — it stays in L2
// when a step carries several frames (a 32 MB table at 8 frames otherwise misses on most probes).
__device__ __forceinline__ uint32_t hash_home(uint32_t key, uint32_t b, uint32_t mask, uint32_t region) {
  return (b * region + (hash_u32(key) & (region - 1u))) & mask;
}

// Rows [lo, hi) of this workgroup when n rows are dealt out as gridDim.x contiguous chunks and XCD x (= blockIdx.x % 8:
// workgroups go to the XCDs round-robin) takes the x-th eighth of them.  A batch is packed sample after sample, so each XCD
// then works on about one sample at a time and that sample's slice of the index (a 4 MB hash region, 2.7 MB of rank words) stays in
// ITS 4 MB L2 — with a grid-stride loop every XCD touched every sample and the probes went to the Infinity Cache.
__device__ __forceinline__ void xcd_chunk(int n, int& lo, int& hi) {
  const int nb = (int)gridDim.x;
  int chunk = (int)blockIdx.x;
  if ((nb & 7) == 0) chunk = ((int)blockIdx.x & 7) * (nb >> 3) + ((int)blockIdx.x >> 3);
  const int per = (n + nb - 1) / nb;
  lo = chunk * per;
  hi = lo + per < n ? lo + per : n;
}

__global__ __launch_bounds__(256) void sp_hash_insert_kernel(const int* __restrict__ indices, int n_cap,
                                                             const int* __restrict__ n_dev, ConvGeom g,
                                                             uint2* __restrict__ slots, uint32_t mask, uint32_t region) {
  int n = n_dev ? *n_dev : n_cap;
  if (n > n_cap) n = n_cap;
  const int i = blockIdx.x * 256 + threadIdx.x;   // (dealing the rows to the XCDs in contiguous eighths, as the lookups do, was
  if (i >= n) return;                             //  measured: 78 -> 87 us — the CAS traffic of a sample then queues on one L2)
  const int4 c = ((const int4*)indices)[i];  // (b, x, y, z)
  uint32_t key = (uint32_t)((((long long)c.x * g.in_shape[0] + c.y) * g.in_shape[1] + c.z) * g.in_shape[2] + c.w);
  uint32_t slot = hash_home(key, (uint32_t)c.x, mask, region);
  for (uint32_t probe = 0; probe <= mask; ++probe) {  // bounded: a full table drops the row instead of spinning
    uint32_t prev = atomicCAS(&slots[slot].x, HASH_EMPTY, key);
    if (prev == HASH_EMPTY || prev == key) { slots[slot].y = (uint32_t)i; return; }
    slot = (slot + 1) & mask;
  }
}

// slots are (key, row) pairs: a probe is ONE 8-byte load (keys and rows in separate arrays cost two cache lines per hit)
__device__ __forceinline__ int hash_lookup(const uint2* __restrict__ slots, uint32_t mask, uint32_t region, uint32_t b,
                                           uint32_t key) {
  uint32_t slot = hash_home(key, b, mask, region);
  for (uint32_t probe = 0; probe <= mask; ++probe) {
    const uint2 e = slots[slot];
    if (e.x == key) return (int)e.y;
    if (e.x == HASH_EMPTY) return -1;
    slot = (slot + 1) & mask;
  }
  return -1;
}

// 16-byte fill (launched like any other kernel, so it is captured into HIP graphs as a kernel node)
__global__ __launch_bounds__(256) void sp_fill_kernel(uint4* __restrict__ p, size_t n16, uint32_t v) {
  const uint4 val = make_uint4(v, v, v, v);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = val;
}

static int fill_u32(void* p, size_t bytes, uint32_t v, hipStream_t stream) {  // bytes % 16 == 0, p 16-byte aligned
  const size_t n16 = bytes / 16;
  if (n16 == 0) return BEVAMD_OK;
  const size_t blocks = (n16 + 255) / 256;
  sp_fill_kernel<<<dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, stream>>>((uint4*)p, n16, v);
  BEVAMD_LAUNCH_CHECK("sp_fill");
  return BEVAMD_OK;
}

// ---- rank index: words[w] = (bits of cells 32w..32w+31, number of set bits before word w) ----------
__device__ __forceinline__ int rank_lookup(const uint2* __restrict__ words, uint32_t key) {
  const uint2 w = words[key >> 5];
  const uint32_t b = 1u << (key & 31);
  return (w.x & b) ? (int)(w.y + __popc(w.x & (b - 1u))) : -1;
}

enum { INDEX_HASH = 0, INDEX_RANK = 1 };

struct IndexRef {  // how to find the row of an input cell
  const uint2* slots;
  uint32_t mask, region;
  const uint2* words;
};

template <int KIND>
__device__ __forceinline__ int index_lookup(const IndexRef& ix, uint32_t b, uint32_t key) {
  if constexpr (KIND == INDEX_HASH) return hash_lookup(ix.slots, ix.mask, ix.region, b, key);
  else return rank_lookup(ix.words, key);
}

// nbr[k][o] for o in [0, m): input row at out*stride - pad + k.  `m_dev` (optional) bounds the rows when
// the count lives on the device; the grid is fixed-size and strides over the rows, so a launch sized for a
// large capacity does not pay for empty workgroups.
template <int KIND>
__global__ __launch_bounds__(256) void sp_nbr_kernel(const int* __restrict__ out_indices, int m_cap,
                                                     const int* __restrict__ m_dev, ConvGeom g, IndexRef ix,
                                                     int* __restrict__ nbr, int nbr_stride) {
  const int k = blockIdx.y;
  int m = m_dev ? *m_dev : m_cap;
  if (m > m_cap) m = m_cap;
  const int kz = k % g.ksize[2];
  const int ky = (k / g.ksize[2]) % g.ksize[1];
  const int kx = k / (g.ksize[2] * g.ksize[1]);
  int lo, hi;
  xcd_chunk(m, lo, hi);
  for (int o = lo + threadIdx.x; o < hi; o += 256) {
    const int4 c = ((const int4*)out_indices)[o];
    int ix_, iy, iz;
    int r = -1;
    if (axis_out_to_in(g, 0, c.y, kx, ix_) && axis_out_to_in(g, 1, c.z, ky, iy) && axis_out_to_in(g, 2, c.w, kz, iz)) {
      uint32_t key = (uint32_t)((((long long)c.x * g.in_shape[0] + ix_) * g.in_shape[1] + iy) * g.in_shape[2] + iz);
      r = index_lookup<KIND>(ix, (uint32_t)c.x, key);
    }
    nbr[(size_t)k * nbr_stride + o] = r;
  }
}

// The same table with ONE thread per output row walking all K offsets of an undilated, untransposed KX x KY x KZ convolution: the
// row's coordinates are read once (not K times), the index words its K cells live in are shared with the neighbouring rows of
// the workgroup (rank index: 8 bytes per 32 cells), every lookup is issued unconditionally (an input cell outside the grid looks
// the row's own first cell up and discards the answer) so that the K loads overlap, and the table column nbr[k][.] is written
// coalesced across the rows.  Every entry is written: nothing has to be cleared first.
template <int KX, int KY, int KZ>
__global__ __launch_bounds__(256) void sp_nbr_rows_kernel(const int* __restrict__ out_indices, int m_cap,
                                                          const int* __restrict__ m_dev, ConvGeom g, const uint2* __restrict__ words,
                                                          int* __restrict__ nbr, int nbr_stride) {
  int m = m_dev ? *m_dev : m_cap;
  if (m > m_cap) m = m_cap;
  int lo, hi;
  xcd_chunk(m, lo, hi);
  const int X = g.in_shape[0], Y = g.in_shape[1], Z = g.in_shape[2];
  for (int o = lo + threadIdx.x; o < hi; o += 256) {
    const int4 c = ((const int4*)out_indices)[o];
    const int x0 = c.y * g.stride[0] - g.pad[0], y0 = c.z * g.stride[1] - g.pad[1], z0 = c.w * g.stride[2] - g.pad[2];
    const uint32_t base = (uint32_t)c.x * (uint32_t)X;
    constexpr int K = KX * KY * KZ;
    uint32_t key[K];
    unsigned okmask = 0;
#pragma unroll
    for (int kx = 0; kx < KX; ++kx)
#pragma unroll
      for (int ky = 0; ky < KY; ++ky)
#pragma unroll
        for (int kz = 0; kz < KZ; ++kz) {
          const int x = x0 + kx, y = y0 + ky, z = z0 + kz, k = (kx * KY + ky) * KZ + kz;
          const bool ok = x >= 0 && x < X && y >= 0 && y < Y && z >= 0 && z < Z;
          key[k] = ok ? ((base + (uint32_t)x) * (uint32_t)Y + (uint32_t)y) * (uint32_t)Z + (uint32_t)z : base * (uint32_t)Y * (uint32_t)Z;
          okmask |= ok ? 1u << k : 0u;
        }
    uint2 wd[K];
#pragma unroll
    for (int k = 0; k < K; ++k) wd[k] = words[key[k] >> 5];
    // (opaque uses: the K loads are all in flight before the first one is waited for; hipcc otherwise pairs each with its use)
#pragma unroll
    for (int k = 0; k < K; ++k) asm volatile("" : "+v"(wd[k].x), "+v"(wd[k].y));
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const uint32_t b = 1u << (key[k] & 31);
      const int r = (wd[k].x & b) ? (int)(wd[k].y + __popc(wd[k].x & (b - 1u))) : -1;
      nbr[(size_t)k * nbr_stride + o] = (okmask >> k) & 1u ? r : -1;
    }
  }
}

// Strided convolution, neighbour table from the INPUT side: each input row knows the <= prod(ceil(k/s))
// (output cell, offset) pairs it feeds; the output row is the rank of that cell.  ~10x fewer lookups than
// probing all K offsets of every output row (most of which have no input).  nbr must be pre-filled with -1.
template <bool GENERAL>
__global__ __launch_bounds__(256) void sp_nbr_from_inputs_kernel(const int* __restrict__ indices, int n_cap,
                                                                 const int* __restrict__ n_dev, ConvGeom g,
                                                                 const uint2* __restrict__ out_words, int m_cap,
                                                                 int* __restrict__ nbr, int nbr_stride) {
  int n = n_dev ? *n_dev : n_cap;
  if (n > n_cap) n = n_cap;
  int lo, hi;
  xcd_chunk(n, lo, hi);
  for (int j = lo + threadIdx.x; j < hi; j += 256) {
    const int4 c = ((const int4*)indices)[j];
    for (int kx = 0; kx < g.ksize[0]; ++kx) {
      int ox, oy, oz;
      if (!axis_in_to_out<GENERAL>(g, 0, c.y, kx, ox)) continue;
      for (int ky = 0; ky < g.ksize[1]; ++ky) {
        if (!axis_in_to_out<GENERAL>(g, 1, c.z, ky, oy)) continue;
        for (int kz = 0; kz < g.ksize[2]; ++kz) {
          if (!axis_in_to_out<GENERAL>(g, 2, c.w, kz, oz)) continue;
          uint32_t key = (uint32_t)((((long long)c.x * g.out_shape[0] + ox) * g.out_shape[1] + oy) * g.out_shape[2] + oz);
          const int row = rank_lookup(out_words, key);
          if (row >= 0 && row < m_cap) nbr[(size_t)((kx * g.ksize[1] + ky) * g.ksize[2] + kz) * nbr_stride + row] = j;
        }
      }
    }
  }
}

// nbr[k][0..m) = -1 for offsets k in [k0, k0 + gridDim.y) (rows bounded by the device count); 16-byte stores
__global__ __launch_bounds__(256) void sp_nbr_clear_kernel(int* __restrict__ nbr, int nbr_stride, int m_cap,
                                                           const int* __restrict__ m_dev, int k0) {
  int m = m_dev ? *m_dev : m_cap;
  if (m > m_cap) m = m_cap;
  int* row = nbr + (size_t)(k0 + blockIdx.y) * nbr_stride;
  if ((nbr_stride & 3) == 0 && ((uintptr_t)nbr & 15) == 0) {
    const int m4 = m >> 2;
    const int4 neg = make_int4(-1, -1, -1, -1);
    for (int o = blockIdx.x * 256 + threadIdx.x; o < m4; o += gridDim.x * 256) ((int4*)row)[o] = neg;
    for (int o = (m4 << 2) + blockIdx.x * 256 + threadIdx.x; o < m; o += gridDim.x * 256) row[o] = -1;
  } else {
    for (int o = blockIdx.x * 256 + threadIdx.x; o < m; o += gridDim.x * 256) row[o] = -1;
  }
}

// SubM neighbour table with half the lookups: in a submanifold convolution input and output sets coincide, so
// nbr[k][o] = i  <=>  nbr[K-1-k][i] = o (the mirrored offset).  Offsets k < K/2 are looked up and scattered to their
// mirror (rows of the upper half are pre-cleared to -1), the centre offset is the identity.  Odd kernel sizes only.
template <int KIND>
__global__ __launch_bounds__(256) void sp_nbr_subm_sym_kernel(const int* __restrict__ indices, int m_cap,
                                                              const int* __restrict__ m_dev, ConvGeom g, IndexRef ix,
                                                              int* __restrict__ nbr, int nbr_stride) {
  const int k = blockIdx.y;  // 0 .. K/2 (K/2 = centre)
  int m = m_dev ? *m_dev : m_cap;
  if (m > m_cap) m = m_cap;
  const int kz = k % g.ksize[2];
  const int ky = (k / g.ksize[2]) % g.ksize[1];
  const int kx = k / (g.ksize[2] * g.ksize[1]);
  const bool centre = k == g.K / 2;
  int lo, hi;
  xcd_chunk(m, lo, hi);
  for (int o = lo + threadIdx.x; o < hi; o += 256) {
    if (centre) {
      nbr[(size_t)k * nbr_stride + o] = o;
      continue;
    }
    const int4 c = ((const int4*)indices)[o];
    const int ix_ = c.y - g.pad[0] + kx, iy = c.z - g.pad[1] + ky, iz = c.w - g.pad[2] + kz;
    int r = -1;
    if (ix_ >= 0 && ix_ < g.in_shape[0] && iy >= 0 && iy < g.in_shape[1] && iz >= 0 && iz < g.in_shape[2]) {
      uint32_t key = (uint32_t)((((long long)c.x * g.in_shape[0] + ix_) * g.in_shape[1] + iy) * g.in_shape[2] + iz);
      r = index_lookup<KIND>(ix, (uint32_t)c.x, key);
    }
    nbr[(size_t)k * nbr_stride + o] = r;
    if (r >= 0 && r < m) nbr[(size_t)(g.K - 1 - k) * nbr_stride + r] = o;
  }
}

// strided conv, pass 1: input j flags every output cell it touches in a BYTE map (one byte per cell) with plain
// stores.  A bitmap would need device-scope atomicOr, and device-scope atomics leave the XCD (the 8 L2s are not coherent
// with each other): 540 k of them cost 25-50 us per level.  Byte stores of the same value race benignly (L2 lines carry
// byte-granular dirty masks), are fire-and-forget, and the map is folded into bitmap words by the popcount pass below.
template <bool GENERAL>
__global__ __launch_bounds__(256) void sp_mark_outputs_kernel(const int* __restrict__ indices, int n_cap,
                                                              const int* __restrict__ n_dev, ConvGeom g,
                                                              uint8_t* __restrict__ cellmap) {
  int n = n_dev ? *n_dev : n_cap;
  if (n > n_cap) n = n_cap;
  int lo, hi;
  xcd_chunk(n, lo, hi);
  for (int j = lo + threadIdx.x; j < hi; j += 256) {
    const int4 c = ((const int4*)indices)[j];
    for (int kx = 0; kx < g.ksize[0]; ++kx) {
      int ox, oy, oz;
      if (!axis_in_to_out<GENERAL>(g, 0, c.y, kx, ox)) continue;
      for (int ky = 0; ky < g.ksize[1]; ++ky) {
        if (!axis_in_to_out<GENERAL>(g, 1, c.z, ky, oy)) continue;
        for (int kz = 0; kz < g.ksize[2]; ++kz) {
          if (!axis_in_to_out<GENERAL>(g, 2, c.w, kz, oz)) continue;
          cellmap[(size_t)((((long long)c.x * g.out_shape[0] + ox) * g.out_shape[1] + oy) * g.out_shape[2] + oz)] = 1;
        }
      }
    }
  }
}

constexpr int RANK_TILE = 2048;  // bitmap words per workgroup in the popcount scan (large grids; index buffers are sized for it)
// Small grids take 256-word tiles: the coarse levels are DENSE (a thread of a 2048-word tile emits up to 256 rows one after the
// other) and few (20 tiles at level 4 of 8 frames, 22 at level 3 of one frame: 37-55 us on a handful of CUs).
constexpr int RANK_TILE_SMALL = 256;
constexpr size_t RANK_SMALL_WORDS = (size_t)1 << 19;   // grids up to this many words (16 M cells) use the small tile
static int rank_tile_for(size_t nwords) { return nwords <= RANK_SMALL_WORDS ? RANK_TILE_SMALL : RANK_TILE; }
static size_t rank_tiles(size_t nwords) { const size_t t = (size_t)rank_tile_for(nwords); return (nwords + t - 1) / t; }

__device__ __forceinline__ unsigned block_exclusive_scan_256u(unsigned v, unsigned* lds_wave /*[4]*/, unsigned* total) {
  const unsigned inc = wave_inclusive_scan(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 63) lds_wave[w] = inc;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned t = lds_wave[i];
    if (i < w) base += t;
    tot += t;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// byte map -> bitmap words (+ per-tile popcount sums).  Thread t of a tile folds 32 consecutive cells (two 16-byte
// loads) into one word; `cellmap` is allocated in whole 32-byte groups (zero filled), so no tail handling.
template <int TILE>
__global__ __launch_bounds__(256) void sp_rank_tile_sums_kernel(const uint8_t* __restrict__ cellmap, uint2* __restrict__ words,
                                                                size_t nwords, uint32_t* __restrict__ tile_sums) {
  __shared__ unsigned lds_wave[4];
  const size_t base = (size_t)blockIdx.x * TILE;
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < TILE / 256; ++i) {
    const size_t w = base + (size_t)i * 256 + threadIdx.x;
    if (w < nwords) {
      const uint4 lo = ((const uint4*)cellmap)[2 * w], hi = ((const uint4*)cellmap)[2 * w + 1];
      const uint32_t v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      uint32_t bits = 0;
#pragma unroll
      for (int q = 0; q < 8; ++q) {  // bytes are 0 or 1: gather bit 0 of each of the 4 bytes of v[q]
        const uint32_t t = v[q] & 0x01010101u;
        bits |= (((t * 0x01020408u) >> 24) & 0xFu) << (4 * q);   // byte b (cell 4q + b) -> bit 24 + b of the product
      }
      words[w].x = bits;
      s += __popc(bits);
    }
  }
  unsigned tot;
  block_exclusive_scan_256u(s, lds_wave, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

// Exclusive popcount prefix of every bitmap word + the coordinates of every active output in ascending linear index, in ONE
// launch (rounds 1-3: a prefix kernel and an emit kernel): `tile_sums` are the raw per-tile popcounts, each workgroup adds up the
// tiles before it itself (a few hundred values: no scan launch in between), completes the prefix of its tile's words and — with
// the bits and the prefixes still in registers — emits the outputs those words hold; the last one publishes the row count,
// clamped to the capacity.
template <int TILE>
__global__ __launch_bounds__(256) void sp_rank_apply_emit_kernel(uint2* __restrict__ words, size_t nwords,
                                                                 const uint32_t* __restrict__ tile_sums, int* count, ConvGeom g,
                                                                 int* __restrict__ out_indices, int out_cap) {
  __shared__ unsigned lds_wave[4];
  unsigned part = 0;
  for (unsigned t = threadIdx.x; t < blockIdx.x; t += 256) part += tile_sums[t];
  part = (unsigned)wave_reduce_add((int)part);
  if ((threadIdx.x & 63) == 0) lds_wave[threadIdx.x >> 6] = part;
  __syncthreads();
  const unsigned tile_base = lds_wave[0] + lds_wave[1] + lds_wave[2] + lds_wave[3];
  __syncthreads();
  constexpr int PER = TILE / 256;
  const size_t w0 = (size_t)blockIdx.x * TILE + (size_t)threadIdx.x * PER;
  uint32_t bits[PER];
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    bits[i] = w0 + i < nwords ? words[w0 + i].x : 0u;
    s += __popc(bits[i]);
  }
  unsigned tot;
  unsigned run = tile_base + block_exclusive_scan_256u(s, lds_wave, &tot);
  // Cell -> (b, x, y, z): three exact divisions per non-empty WORD (cell 32 w), then per set bit the carry chain z -> y -> x -> b with
  // multiply-high by floor(2^32 / d) + 1, exact while numerator * d < 2^32 (numerators stay below d + 32).  The coarse levels are
  // dense — a thread emits up to 256 rows — and three 32-bit divisions per row (~100 instructions) were 37-55 us of this kernel
  // on ONE frame.
  const uint32_t Z = (uint32_t)g.out_shape[2], Y = (uint32_t)g.out_shape[1], X = (uint32_t)g.out_shape[0];
  const uint32_t mz = 0xFFFFFFFFu / Z + 1u, my = 0xFFFFFFFFu / Y + 1u, mx = 0xFFFFFFFFu / X + 1u;
  const bool small = Z - 2u < 32766u && Y - 2u < 32766u && X - 2u < 32766u;   // d = 1 has no 32-bit magic
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    if (w0 + i < nwords) words[w0 + i].y = run;
    uint32_t b = bits[i];
    if (!b) continue;
    uint32_t k = (uint32_t)((w0 + i) * 32);
    const uint32_t z0 = k % Z; k /= Z;
    const uint32_t y0 = k % Y; k /= Y;
    const uint32_t x0 = k % X; k /= X;
    while (b) {
      const int t = __ffs(b) - 1;
      b &= b - 1;
      if (run < (unsigned)out_cap) {
        uint32_t n = z0 + (uint32_t)t;
        uint32_t q = small ? __umulhi(n, mz) : n / Z;
        const int oz = (int)(n - q * Z);
        n = y0 + q;
        q = small ? __umulhi(n, my) : n / Y;
        const int oy = (int)(n - q * Y);
        n = x0 + q;
        q = small ? __umulhi(n, mx) : n / X;
        const int ox = (int)(n - q * X);
        ((int4*)out_indices)[run] = make_int4((int)(k + q), ox, oy, oz);
      }
      ++run;
    }
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    const unsigned total = tile_base + tot;
    *count = (int)(total < (unsigned)out_cap ? total : (unsigned)out_cap);
  }
}

// reference-shaped rulebook from nbr: per offset, compact (in,out) pairs ordered by out row
__global__ __launch_bounds__(256) void sp_pair_flags_kernel(const int* __restrict__ nbr, int nbr_stride, int m,
                                                            int K, uint32_t* __restrict__ flags) {
  int o = blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  if (o >= m) return;
  flags[(size_t)k * m + o] = nbr[(size_t)k * nbr_stride + o] >= 0 ? 1u : 0u;
}

__global__ __launch_bounds__(256) void sp_pairs_scatter_kernel(const int* __restrict__ nbr, int nbr_stride, int m,
                                                               int K, const uint32_t* __restrict__ scan,
                                                               int* __restrict__ pairs, int pairs_len,
                                                               int* __restrict__ indice_num) {
  int o = blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  if (o >= m) return;
  const size_t e = (size_t)k * m + o;
  const uint32_t base = scan[(size_t)k * m];
  int r = nbr[(size_t)k * nbr_stride + o];
  if (r >= 0) {
    uint32_t pos = scan[e] - base;
    if (pos < (uint32_t)pairs_len) {  // always true for pairs_len >= min(#inputs, #outputs)
      pairs[((size_t)k * 2 + 0) * pairs_len + pos] = r;
      pairs[((size_t)k * 2 + 1) * pairs_len + pos] = o;
    }
  }
  if (o == m - 1) indice_num[k] = (int)(scan[e] - base) + (r >= 0 ? 1 : 0);
}

// nbr (output-stationary) from reference-shaped pairs — for the drop-in indice_conv entry points
__global__ __launch_bounds__(256) void sp_nbr_from_pairs_kernel(const int* __restrict__ pairs, int pairs_len,
                                                                const int* __restrict__ indice_num, int K,
                                                                int in_col, int* __restrict__ nbr, int nbr_stride) {
  int t = blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  if (t >= indice_num[k]) return;
  int i = pairs[((size_t)k * 2 + in_col) * pairs_len + t];
  int o = pairs[((size_t)k * 2 + (1 - in_col)) * pairs_len + t];
  nbr[(size_t)k * nbr_stride + o] = i;
}

__global__ void sp_set_int_kernel(int* p, int v) { *p = v; }

// nbrT[k][i] = o  for every (k, o) with nbr[k][o] = i >= 0: the input-stationary view of the same
// rulebook (each input row feeds at most one output row per offset), used by the input-gradient pass.
__global__ __launch_bounds__(256) void sp_transpose_nbr_kernel(const int* __restrict__ nbr, int nbr_stride, int m,
                                                               int* __restrict__ nbr_t, int nbr_t_stride) {
  int o = blockIdx.x * 256 + threadIdx.x;
  int k = blockIdx.y;
  if (o >= m) return;
  int i = nbr[(size_t)k * nbr_stride + o];
  if (i >= 0) nbr_t[(size_t)k * nbr_t_stride + i] = o;
}

// Dense BEV tail of SparseEncoder (sparse_encoder.py:126-131: out.dense() -> permute(0,1,4,2,3) -> view):
// out[b][c*Z + z][x][y] = features[row(b,x,y,z)][c], zeros where no voxel is active.  Written as a GATHER over
// the output (every element stored exactly once, 128-byte runs along y) instead of zero-fill + scatter.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void sp_dense_bev_kernel(const T* __restrict__ feat, int pitch, int C, IndexRef ix,
                                                           int X, int Y, int Z, T* __restrict__ out) {
  extern __shared__ int rows[];  // [Z][64]
  const int b = blockIdx.z, h = blockIdx.y, w0 = blockIdx.x * 64;
  for (int t = threadIdx.x; t < 64 * Z; t += 256) {
    const int d = t >> 6, w = w0 + (t & 63);
    int r = -1;
    if (w < Y) r = index_lookup<KIND>(ix, (uint32_t)b, (uint32_t)((((long long)b * X + h) * Y + w) * Z + d));
    rows[t] = r;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, w = w0 + lane;
  const int CZ = C * Z;
  for (int idx = wave; idx < CZ; idx += 4) {
    const int c = idx / Z, d = idx - c * Z;
    const int r = rows[d * 64 + lane];
    T v = r >= 0 ? feat[(size_t)r * pitch + c] : (T)0;
    if (w < Y) out[(((size_t)b * CZ + idx) * X + h) * Y + w] = v;
  }
}

// Same result with the tile's rows staged through LDS: the 64*Z rows of a tile are copied in with coalesced loads (a row
// per wave instruction) instead of being read 2 bytes at a time, one row per lane, once per channel; the transposed
// reads then come out of LDS (pitch = row + 4 bytes: an odd number of dwords, conflict-free along the row index).
// Needs row bytes % 4 == 0 and 64*Z*(row + 4) bytes of LDS; the host falls back to the kernel above otherwise.
// WIDE: rows of 16 * LPR bytes with 64 % LPR == 0, 16-byte aligned (the encoder's 128-channel 16-bit rows: LPR = 16): the staging
// loop above it was ONE dependent chain per row — row id out of LDS, 4 bytes per lane from global, LDS store, 32 times per wave:
// ~10 us per tile of pure latency (round 6: 72 us per 8-frame step for a 190 MB kernel).  Here a wave instruction loads 64 / LPR
// whole rows in 16-byte pieces and all of a wave's rows are in flight before the first is stored: one round trip per tile.
template <typename T, int KIND, bool WIDE = false>
__global__ __launch_bounds__(256) void sp_dense_bev_staged_kernel(const T* __restrict__ feat, int pitch, int C, IndexRef ix,
                                                                  int X, int Y, int Z, T* __restrict__ out) {
  extern __shared__ int rows[];  // [Z][64] row ids, then the staged rows
  const int b = blockIdx.z, h = blockIdx.y, w0 = blockIdx.x * 64;
  const int nrows = 64 * Z;
  for (int t = threadIdx.x; t < nrows; t += 256) {
    const int d = t >> 6, w = w0 + (t & 63);
    int r = -1;
    if (w < Y) r = index_lookup<KIND>(ix, (uint32_t)b, (uint32_t)((((long long)b * X + h) * Y + w) * Z + d));
    rows[t] = r;
  }
  __syncthreads();
  const int row_dw = C * (int)sizeof(T) / 4, lpitch = row_dw + 1;   // dwords
  uint32_t* tile = (uint32_t*)(rows + nrows);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (WIDE) {
    const int lpr = row_dw >> 2, rpi = 64 / lpr;        // lanes per row, rows per wave instruction
    const int sub = lane / lpr, piece = lane - sub * lpr;
    constexpr int U = 8;                                // wave instructions in flight: U * rpi rows per wave and trip
    for (int t0 = wave * U * rpi; t0 < nrows; t0 += 4 * U * rpi) {
      uint4 v[U];
      int tt[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        tt[u] = t0 + u * rpi + sub;
        const int r = tt[u] < nrows ? rows[tt[u]] : -1;
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (r >= 0) v[u] = *(const uint4*)((const char*)feat + (size_t)r * pitch * sizeof(T) + piece * 16);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (tt[u] < nrows) {
          uint32_t* dst = tile + tt[u] * lpitch + piece * 4;   // the odd dword pitch keeps the transposed reads conflict-free: 4-byte stores
          dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
        }
      }
    }
  } else {
    for (int t = wave; t < nrows; t += 4) {   // one staged row per wave and trip; all loads of a trip are independent
      const int r = rows[t];
      const uint32_t* src = (const uint32_t*)(feat + (size_t)(r >= 0 ? r : 0) * pitch);
      for (int j = lane; j < row_dw; j += 64) tile[t * lpitch + j] = r >= 0 ? src[j] : 0u;
    }
  }
  __syncthreads();
  const int CZ = C * Z;
  const T* tl = (const T*)tile;
  constexpr int EPD = 4 / (int)sizeof(T);   // elements per dword
  constexpr int YPL = 8 / (int)sizeof(T);   // y positions per lane: one 8-byte store
  if (Y % YPL == 0) {
    // a lane stores YPL consecutive y of one (channel, z) plane (8 bytes: rows of the output start 8-byte aligned when
    // Y % YPL == 0), a wave 64*YPL/64 planes per instruction — 4x fewer store instructions than one element per lane
    constexpr int LPP = 64 / YPL;            // lanes per plane row of the tile
    constexpr int PPI = 64 / LPP;            // planes per wave instruction
    const int sub = lane / LPP, chunk = lane % LPP;
#pragma unroll 4   // four independent groups of LDS reads in flight per lane (the loop is a chain of LDS round trips otherwise)
    for (int idx0 = wave * PPI; idx0 < CZ; idx0 += 4 * PPI) {
      const int idx = idx0 + sub;
      if (idx >= CZ) continue;
      const int c = idx / Z, d = idx - c * Z;
      T v[YPL];
#pragma unroll
      for (int e = 0; e < YPL; ++e) v[e] = tl[(size_t)(d * 64 + chunk * YPL + e) * lpitch * EPD + c];
      const int w = w0 + chunk * YPL;
      if (w < Y) *(uint2*)(out + (((size_t)b * CZ + idx) * X + h) * Y + w) = *(const uint2*)v;   // w + YPL <= Y: both multiples of YPL
    }
    return;
  }
  const int w = w0 + lane;
  for (int idx = wave; idx < CZ; idx += 4) {
    const int c = idx / Z, d = idx - c * Z;
    const T v = tl[(size_t)(d * 64 + lane) * lpitch * EPD + c];
    if (w < Y) out[(((size_t)b * CZ + idx) * X + h) * Y + w] = v;
  }
}

// Bitmap words + tile popcounts of a strided convolution's output set WITHOUT the byte map, for an input set in ascending linear
// index: a workgroup owns one tile of output words, finds the input rows that can reach it — the x-planes (ox*s - p .. ox*s - p +
// k - 1) of its output planes are ONE contiguous row range, read off the input level's sorted-key directory (SRC 0) or its rank
// index (SRC 1: rows before a cell = prefix + popcount) — and sets their output cells in an LDS bitmap.  One launch instead of
// fill + sp_mark_outputs + sp_rank_tile_sums, and no 32-bytes-per-word map (87 MB at level 2 of 8 frames: written by the fill,
// read-modify-written by the marks, read back by the fold); inputs are read ~2.5 times (tiles share boundary planes).
// Undilated, untransposed convolutions only.
template <int TILE, int SRC>
__global__ __launch_bounds__(256) void sp_rank_tiles_from_rows_kernel(const int* __restrict__ indices, int n_cap,
                                                                      const int* __restrict__ n_dev, ConvGeom g,
                                                                      const int* __restrict__ in_xstart,
                                                                      const uint2* __restrict__ in_words,
                                                                      uint2* __restrict__ words, size_t nwords,
                                                                      uint32_t* __restrict__ tile_sums) {
  __shared__ unsigned bm[TILE];
  __shared__ unsigned lds_wave[4];
  int n = n_dev ? *n_dev : n_cap;
  if (n > n_cap) n = n_cap;
  for (int i = threadIdx.x; i < TILE; i += 256) bm[i] = 0u;
  __syncthreads();
  const int X = g.in_shape[0], Y = g.in_shape[1], Z = g.in_shape[2];
  const int OX = g.out_shape[0], OY = g.out_shape[1], OZ = g.out_shape[2];
  const unsigned long long volume = (unsigned long long)g.batch * OX * OY * OZ;
  const unsigned long long c0 = (unsigned long long)blockIdx.x * TILE * 32;
  const unsigned long long c1 = c0 + (unsigned long long)TILE * 32 < volume ? c0 + (unsigned long long)TILE * 32 : volume;
  if (c0 < c1 && n > 0) {
    const unsigned long long plane = (unsigned long long)OY * OZ;
    const unsigned long long p0 = c0 / plane, p1 = (c1 - 1) / plane;   // (b * OX + ox) of the first / last cell
    const int b_lo = (int)(p0 / OX), b_hi = (int)(p1 / OX);
    for (int b = b_lo; b <= b_hi; ++b) {
      const int ox_lo = b == b_lo ? (int)(p0 % OX) : 0, ox_hi = b == b_hi ? (int)(p1 % OX) : OX - 1;
      int ix_lo = ox_lo * g.stride[0] - g.pad[0], ix_hi = ox_hi * g.stride[0] - g.pad[0] + g.ksize[0] - 1;
      ix_lo = ix_lo < 0 ? 0 : ix_lo;
      ix_hi = ix_hi > X - 1 ? X - 1 : ix_hi;
      if (ix_lo > ix_hi) continue;
      int r0, r1;
      if constexpr (SRC == 0) {
        r0 = in_xstart[b * X + ix_lo];
        r1 = in_xstart[b * X + ix_hi + 1];
      } else {
        const uint32_t k0 = (uint32_t)(b * X + ix_lo) * (uint32_t)Y * (uint32_t)Z;
        const uint2 w0 = in_words[k0 >> 5];
        r0 = (int)(w0.y + __popc(w0.x & ((1u << (k0 & 31)) - 1u)));
        if (b * X + ix_hi + 1 >= g.batch * X) {
          r1 = n;
        } else {
          const uint32_t k1 = (uint32_t)(b * X + ix_hi + 1) * (uint32_t)Y * (uint32_t)Z;
          const uint2 w1 = in_words[k1 >> 5];
          r1 = (int)(w1.y + __popc(w1.x & ((1u << (k1 & 31)) - 1u)));
        }
      }
      r0 = r0 < 0 ? 0 : r0;            // (a directory built from rows that were not in linear order may hold anything)
      r1 = r1 > n ? n : r1;
      // four rows per thread in flight (a coarse level has a few hundred tiles and ~25 rows per thread: one dependent 16-byte
      // load per iteration left the kernel waiting on memory latency)
      for (int j0 = r0 + (int)threadIdx.x; j0 < r1; j0 += 4 * 256) {
        int4 cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cc[u] = ((const int4*)indices)[j0 + u * 256 < r1 ? j0 + u * 256 : r1 - 1];
#pragma unroll
        for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(cc[u].x), "+v"(cc[u].y), "+v"(cc[u].z), "+v"(cc[u].w));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
        const int4 c = cc[u];
        if (j0 + u * 256 >= r1 || c.x != b) continue;
        // the outputs an input feeds, axis by axis: o = (i + pad - k) / stride for the taps k where that division is exact
        // (at most 3 per axis for k <= 3; powers of two by shift — three runtime divisions per tap made this kernel 2-5 x slower
        // than the byte-map kernels it replaces)
        int lo[3][3], ln[3];
        const int ci[3] = {c.y, c.z, c.w};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          const int st = g.stride[a], sm = st - 1;
          const bool p2 = (st & sm) == 0;
          const int sh = __ffs(st) - 1;
          int cnt = 0;
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int t = ci[a] + g.pad[a] - k;
            const bool ok = k < g.ksize[a] && t >= 0 && (p2 ? (t & sm) == 0 : t % st == 0);
            const int o = p2 ? t >> sh : t / st;
            if (ok && o < g.out_shape[a]) {
#pragma unroll
              for (int q = 0; q < 3; ++q) lo[a][q] = q == cnt ? o : lo[a][q];
              ++cnt;
            }
          }
          ln[a] = cnt;
        }
        const uint32_t t0 = (uint32_t)c0, t1 = (uint32_t)c1;   // (batch * volume < 2^32 - 16: make_geom)
        for (int ix = 0; ix < ln[0]; ++ix) {
          const uint32_t bx = ((uint32_t)c.x * (uint32_t)OX + (uint32_t)(ix == 0 ? lo[0][0] : ix == 1 ? lo[0][1] : lo[0][2])) * (uint32_t)OY;
          for (int iy = 0; iy < ln[1]; ++iy) {
            const uint32_t by = (bx + (uint32_t)(iy == 0 ? lo[1][0] : iy == 1 ? lo[1][1] : lo[1][2])) * (uint32_t)OZ;
            for (int iz = 0; iz < ln[2]; ++iz) {
              const uint32_t cell = by + (uint32_t)(iz == 0 ? lo[2][0] : iz == 1 ? lo[2][1] : lo[2][2]);
              if (cell >= t0 && cell < t1) atomicOr(&bm[(cell - t0) >> 5], 1u << (cell & 31u));
            }
          }
        }
        }
      }
    }
  }
  __syncthreads();
  constexpr int PER = TILE / 256;
  const size_t w0i = (size_t)blockIdx.x * TILE + (size_t)threadIdx.x * PER;
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i)
    if (w0i + i < nwords) {
      const unsigned bits = bm[threadIdx.x * PER + i];
      words[w0i + i].x = bits;
      s += __popc(bits);
    }
  unsigned tot;
  block_exclusive_scan_256u(s, lds_wave, &tot);
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = tot;
}

static int make_geom(int batch, const int* in_shape, const int* out_shape, const int* ksize, const int* stride,
                     const int* pad, const int* dil, int subm, ConvGeom& g, int transpose = 0) {
  BEVAMD_REQUIRE(in_shape && out_shape && ksize && stride && pad, "spconv: null geometry (host pointers)");
  g.batch = batch;
  g.K = 1;
  g.transpose = (transpose && !subm) ? 1 : 0;  // the SubM functors ignore the flag (indice.cu.h:147-203)
  for (int i = 0; i < 3; ++i) {
    g.dil[i] = dil ? dil[i] : 1;
    BEVAMD_REQUIRE(g.dil[i] > 0, "spconv: dilation must be > 0 on axis %d", i);
    g.in_shape[i] = in_shape[i];
    g.ksize[i] = ksize[i];
    g.stride[i] = subm ? 1 : stride[i];
    g.pad[i] = subm ? ksize[i] / 2 : pad[i];  // spconv_ops.h:78-81
    g.out_shape[i] = subm ? in_shape[i] : out_shape[i];
    BEVAMD_REQUIRE(g.in_shape[i] > 0 && g.out_shape[i] > 0 && g.ksize[i] > 0 && g.stride[i] > 0 && g.pad[i] >= 0,
                   "spconv: bad geometry on axis %d", i);
    g.K *= ksize[i];
  }
  BEVAMD_REQUIRE(g.K <= 4096, "spconv: kernel volume %d > 4096", g.K);  // spconv_ops.h:51
  BEVAMD_REQUIRE(batch > 0, "spconv: batch_size must be > 0");
  unsigned long long vin = (unsigned long long)batch * g.in_shape[0] * g.in_shape[1] * g.in_shape[2];
  unsigned long long vout = (unsigned long long)batch * g.out_shape[0] * g.out_shape[1] * g.out_shape[2];
  BEVAMD_REQUIRE(vin < 0xFFFFFFF0ull && vout < 0xFFFFFFF0ull, "spconv: batch * volume must be < 2^32 - 16");
  return BEVAMD_OK;
}

static uint32_t hash_region(uint32_t cap, int batch) {  // slots per sample region: cap / (batch rounded up to a power of two)
  uint32_t parts = 1;
  while ((int)parts < batch && parts < cap) parts <<= 1;
  return cap / parts;
}

static uint32_t hash_capacity(size_t n) {
  uint32_t cap = 1024;
  while (cap < 2 * n + 16) cap <<= 1;
  return cap;
}

static int conv_bound(const ConvGeom& g) {  // outputs one input can activate
  int b = 1;
  for (int i = 0; i < 3; ++i)
    b *= (g.transpose || g.dil[i] != 1) ? g.ksize[i] : (g.ksize[i] + g.stride[i] - 1) / g.stride[i];
  return b;
}

static size_t grid_words(int batch, const int* shape) {
  unsigned long long v = (unsigned long long)(batch > 0 ? batch : 1) * shape[0] * shape[1] * shape[2];
  return (size_t)((v + 31) / 32);
}

static size_t rank_index_bytes(int batch, const int* shape) {
  const size_t nw = grid_words(batch, shape), nt = rank_tiles(nw);
  return align_up(nw * 8, 256) + align_up((nt + 1) * 4, 256) + align_up(scan_workspace_bytes(nt + 1), 256) +
         align_up(nw * 32, 256) /* byte map */ + 256;
}

static int hash_build(const int* indices, int n_cap, const int* n_dev, const ConvGeom& g, void* index, size_t bytes,
                      hipStream_t stream) {
  const uint32_t cap = hash_capacity((size_t)n_cap);
  if (!index || bytes < (size_t)cap * 8) {
    set_error("spconv hash index: buffer too small (%zu < %zu)", bytes, (size_t)cap * 8);
    return BEVAMD_ERR_WORKSPACE;
  }
  uint2* slots = (uint2*)index;
  int frc = fill_u32(slots, (size_t)cap * 8, HASH_EMPTY, stream);   // rows are overwritten by the insert
  if (frc) return frc;
  if (n_cap > 0) {
    sp_hash_insert_kernel<<<dim3(cdiv(n_cap, 256)), dim3(256), 0, stream>>>(indices, n_cap, n_dev, g, slots, cap - 1, hash_region(cap, g.batch));
    BEVAMD_LAUNCH_CHECK("sp_hash_insert");
  }
  return BEVAMD_OK;
}

static IndexRef hash_ref(const void* index, int n_cap, int batch) {
  const uint32_t cap = hash_capacity((size_t)n_cap);
  IndexRef r;
  r.slots = (const uint2*)index;
  r.mask = cap - 1;
  r.region = hash_region(cap, batch);
  r.words = nullptr;
  return r;
}

static IndexRef rank_ref(const void* index) {
  IndexRef r;
  r.slots = nullptr;
  r.mask = 0;
  r.region = 1;
  r.words = (const uint2*)index;
  return r;
}

// (Round 4: capping the rulebook kernels' grids so that they leave wave slots to the convolutions they run beside made the LiDAR
// branch SLOWER — 2048 workgroups 3.75 ms, 1024: 3.81, 512: 3.89, 256: 4.18 on one box: the convolutions do not wait for slots,
// and a slower chain lands on the critical path.)
static unsigned stride_grid(long long n) {  // fixed-size grid for grid-stride kernels
  long long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : b > 2048 ? 2048 : b);
}

// active outputs of a strided convolution, ascending linear index, + their rank index (+ optionally the
// output-stationary neighbour table, generated from the input side)
static int downsample(const int* indices, int n_cap, const int* n_dev, const ConvGeom& g, int* out_indices, int out_cap,
                      int* num_out_dev, void* out_index, size_t bytes, int* nbr, int nbr_stride, hipStream_t stream) {
  const size_t need = rank_index_bytes(g.batch, g.out_shape);
  if (!out_index || bytes < need) {
    set_error("spconv rank index: buffer too small (%zu < %zu)", bytes, need);
    return BEVAMD_ERR_WORKSPACE;
  }
  const size_t nw = grid_words(g.batch, g.out_shape), nt = rank_tiles(nw);
  const bool general = g.transpose || g.dil[0] != 1 || g.dil[1] != 1 || g.dil[2] != 1;
  Carver cv(out_index, bytes);
  uint2* words = cv.take<uint2>(nw);   // must stay first: the rank index IS this array
  uint32_t* tile_sums = cv.take<uint32_t>(nt + 1);
  void* sws = cv.take<char>(align_up(scan_workspace_bytes(nt + 1), 256));
  const size_t sws_bytes = align_up(scan_workspace_bytes(nt + 1), 256);
  uint8_t* cellmap = cv.take<uint8_t>(nw * 32);
  int frc = fill_u32(cellmap, nw * 32, 0u, stream);
  if (frc) return frc;
  if (n_cap > 0) {
    if (general) sp_mark_outputs_kernel<true><<<dim3(stride_grid(n_cap)), dim3(256), 0, stream>>>(indices, n_cap, n_dev, g, cellmap);
    else sp_mark_outputs_kernel<false><<<dim3(stride_grid(n_cap)), dim3(256), 0, stream>>>(indices, n_cap, n_dev, g, cellmap);
    BEVAMD_LAUNCH_CHECK("sp_mark_outputs");
  }
  (void)sws;
  (void)sws_bytes;
  if (rank_tile_for(nw) == RANK_TILE_SMALL) {
    sp_rank_tile_sums_kernel<RANK_TILE_SMALL><<<dim3((unsigned)nt), dim3(256), 0, stream>>>(cellmap, words, nw, tile_sums);
    BEVAMD_LAUNCH_CHECK("sp_rank_tile_sums");
    sp_rank_apply_emit_kernel<RANK_TILE_SMALL><<<dim3((unsigned)nt), dim3(256), 0, stream>>>(words, nw, tile_sums, num_out_dev, g, out_indices, out_cap);
  } else {
    sp_rank_tile_sums_kernel<RANK_TILE><<<dim3((unsigned)nt), dim3(256), 0, stream>>>(cellmap, words, nw, tile_sums);
    BEVAMD_LAUNCH_CHECK("sp_rank_tile_sums");
    sp_rank_apply_emit_kernel<RANK_TILE><<<dim3((unsigned)nt), dim3(256), 0, stream>>>(words, nw, tile_sums, num_out_dev, g, out_indices, out_cap);
  }
  BEVAMD_LAUNCH_CHECK("sp_rank_apply_emit");
  if (nbr) {
    sp_nbr_clear_kernel<<<dim3(stride_grid(((long long)out_cap + 3) / 4), g.K), dim3(256), 0, stream>>>(nbr, nbr_stride, out_cap, num_out_dev, 0);
    BEVAMD_LAUNCH_CHECK("sp_nbr_clear");
    if (n_cap > 0) {
      if (general) sp_nbr_from_inputs_kernel<true><<<dim3(stride_grid(n_cap)), dim3(256), 0, stream>>>(indices, n_cap, n_dev, g, words,
                                                                                                      out_cap, nbr, nbr_stride);
      else sp_nbr_from_inputs_kernel<false><<<dim3(stride_grid(n_cap)), dim3(256), 0, stream>>>(indices, n_cap, n_dev, g, words,
                                                                                               out_cap, nbr, nbr_stride);
      BEVAMD_LAUNCH_CHECK("sp_nbr_from_inputs");
    }
  }
  return BEVAMD_OK;
}

static int neighbors(const int* out_indices, int m_cap, const int* m_dev, const ConvGeom& g, int kind, const IndexRef& ix,
                     int* nbr, int nbr_stride, bool subm, hipStream_t stream) {
  if (m_cap <= 0) return BEVAMD_OK;
  const bool odd = (g.ksize[0] & 1) && (g.ksize[1] & 1) && (g.ksize[2] & 1);
  const bool undilated = g.dil[0] == 1 && g.dil[1] == 1 && g.dil[2] == 1;  // pad = k/2 centres the window only then
  if (subm && odd && undilated && g.K > 1) {
    // mirrored offsets: clear the upper half, look up the lower half + centre (measured with the hash index, 160 k rows:
    // 33 us against 55 us for all 26 lookups — the random probes set the time, not the scattered mirror stores)
    const int half = g.K / 2;
    sp_nbr_clear_kernel<<<dim3(stride_grid(((long long)m_cap + 3) / 4), half), dim3(256), 0, stream>>>(nbr, nbr_stride, m_cap,
                                                                                                     m_dev, half + 1);
    BEVAMD_LAUNCH_CHECK("sp_nbr_clear");
    dim3 grid(stride_grid(m_cap), half + 1), block(256);
    if (kind == INDEX_HASH) sp_nbr_subm_sym_kernel<INDEX_HASH><<<grid, block, 0, stream>>>(out_indices, m_cap, m_dev, g, ix, nbr, nbr_stride);
    else sp_nbr_subm_sym_kernel<INDEX_RANK><<<grid, block, 0, stream>>>(out_indices, m_cap, m_dev, g, ix, nbr, nbr_stride);
    BEVAMD_LAUNCH_CHECK("sp_nbr_subm_sym");
    return BEVAMD_OK;
  }
  if (kind == INDEX_RANK && !g.transpose && undilated) {   // neighbouring rows share index words: one thread per row, all offsets
    const bool k333 = g.ksize[0] == 3 && g.ksize[1] == 3 && g.ksize[2] == 3;
    const bool k113 = g.ksize[0] == 1 && g.ksize[1] == 1 && g.ksize[2] == 3;
    if (k333 || k113) {
      const dim3 grid(stride_grid(m_cap)), block(256);
      if (k333) sp_nbr_rows_kernel<3, 3, 3><<<grid, block, 0, stream>>>(out_indices, m_cap, m_dev, g, ix.words, nbr, nbr_stride);
      else sp_nbr_rows_kernel<1, 1, 3><<<grid, block, 0, stream>>>(out_indices, m_cap, m_dev, g, ix.words, nbr, nbr_stride);
      BEVAMD_LAUNCH_CHECK("sp_nbr_rows");
      return BEVAMD_OK;
    }
  }
  dim3 grid(stride_grid(m_cap), g.K), block(256);
  if (kind == INDEX_HASH) sp_nbr_kernel<INDEX_HASH><<<grid, block, 0, stream>>>(out_indices, m_cap, m_dev, g, ix, nbr, nbr_stride);
  else sp_nbr_kernel<INDEX_RANK><<<grid, block, 0, stream>>>(out_indices, m_cap, m_dev, g, ix, nbr, nbr_stride);
  BEVAMD_LAUNCH_CHECK("sp_nbr");
  return BEVAMD_OK;
}

// Slab metadata of the 3x3x3 SubM convolution straight from the index of the voxel set: thread t of block b looks its 27
// neighbour cells up itself (one 8-byte rank-index load each; the three kz taps of a (kx, ky) line sit in the same or the
// next word) and the block emits (range, slots) — the int32 table nbr [27, N] (108 B per row written by the neighbour
// kernel, read back by slab_build_kernel) never exists.
template <int BM, int KIND>
__global__ __launch_bounds__(BM) void sp_slab_from_index_kernel(const int* __restrict__ indices, int m_cap,
                                                                const int* __restrict__ m_dev, ConvGeom g, IndexRef ix,
                                                                int2* __restrict__ hdr, uint16_t* __restrict__ slots,
                                                                int* __restrict__ status, int fmt) {
  int m = m_dev ? *m_dev : m_cap;
  if (m > m_cap) m = m_cap;
  // Launches are sized by capacity (5-14x the live rows at 8 frames): nobody reads a dead block's metadata.  The live blocks
  // are dealt to the XCDs in contiguous eighths (see xcd_chunk): the 27 lookups of neighbouring rows hit the same index words.
  const int nblk = (m + BM - 1) / BM, per = (nblk + 7) >> 3;
  const int xcd = (int)blockIdx.x & 7, bix = (int)blockIdx.x >> 3;
  const int blk = (gridDim.x & 7u) == 0 ? xcd * per + bix : (int)blockIdx.x;
  if (((gridDim.x & 7u) == 0 && bix >= per) || blk >= nblk) return;
  const int t = threadIdx.x, row = blk * BM + t;
  const bool live = row < m;
  const int4 c = live ? ((const int4*)indices)[row] : make_int4(0, 0, 0, 0);
  int v[27];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        const int k = (kx * 3 + ky) * 3 + kz;
        const int x = c.y - 1 + kx, y = c.z - 1 + ky, z = c.w - 1 + kz;
        int r = -1;
        if (k == 13) {
          r = live ? row : -1;
        } else if (live && x >= 0 && x < g.in_shape[0] && y >= 0 && y < g.in_shape[1] && z >= 0 && z < g.in_shape[2]) {
          const uint32_t key = (uint32_t)((((long long)c.x * g.in_shape[0] + x) * g.in_shape[1] + y) * g.in_shape[2] + z);
          r = index_lookup<KIND>(ix, (uint32_t)c.x, key);
          if (r >= m) r = -1;
        }
        v[k] = r;
      }
  slab::slab_emit<BM>(v, blk, t, hdr, slots, status, fmt);
}

// The same from a RANK index with all 26 lookups of a row in flight together.  The generic kernel above compiles to one load, one
// wait and one branch per lookup — bits, then the prefix word behind a second branch: 52 dependent round trips per row, 126 us for
// the 2 M rows of level 2 at 8 frames.  Here every tap has a valid (clamped) key, the 8-byte words are fetched unconditionally
// and used behind opaque barriers (hipcc otherwise sinks each load to its use), the range checks only select the result.
template <int BM>
__global__ __launch_bounds__(BM) void sp_slab_from_rank_kernel(const int* __restrict__ indices, int m_cap,
                                                               const int* __restrict__ m_dev, ConvGeom g,
                                                               const uint2* __restrict__ words, int2* __restrict__ hdr,
                                                               uint16_t* __restrict__ slots, int* __restrict__ status, int fmt) {
  int m = m_dev ? *m_dev : m_cap;
  if (m > m_cap) m = m_cap;
  const int nblk = (m + BM - 1) / BM, per = (nblk + 7) >> 3;
  const int xcd = (int)blockIdx.x & 7, bix = (int)blockIdx.x >> 3;
  const int blk = (gridDim.x & 7u) == 0 ? xcd * per + bix : (int)blockIdx.x;
  if (((gridDim.x & 7u) == 0 && bix >= per) || blk >= nblk) return;
  const int t = threadIdx.x, row = blk * BM + t;
  const bool live = row < m;
  int4 c = live ? ((const int4*)indices)[row] : make_int4(0, 0, 0, 0);
  const int X = g.in_shape[0], Y = g.in_shape[1], Z = g.in_shape[2];
  // coordinates outside the grid (a caller's broken promise) must not turn into addresses outside the index
  const bool inside = c.x >= 0 && c.x < g.batch && c.y >= 0 && c.y < X && c.z >= 0 && c.z < Y && c.w >= 0 && c.w < Z;
  if (!inside) c = make_int4(0, 0, 0, 0);
  const uint32_t base = (uint32_t)c.x * (uint32_t)X;
  const uint32_t safe = ((base + (uint32_t)c.y) * (uint32_t)Y + (uint32_t)c.z) * (uint32_t)Z + (uint32_t)c.w;   // the row's own cell
  uint32_t key[27];
  unsigned okmask = 0;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kz = 0; kz < 3; ++kz) {
        const int x = c.y - 1 + kx, y = c.z - 1 + ky, z = c.w - 1 + kz, k = (kx * 3 + ky) * 3 + kz;
        const bool ok = live && inside && x >= 0 && x < X && y >= 0 && y < Y && z >= 0 && z < Z;
        key[k] = ok ? ((base + (uint32_t)x) * (uint32_t)Y + (uint32_t)y) * (uint32_t)Z + (uint32_t)z : safe;
        okmask |= ok ? 1u << k : 0u;
      }
  uint2 wd[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wd[k] = k == 13 ? make_uint2(0u, 0u) : words[key[k] >> 5];
#pragma unroll
  for (int k = 0; k < 27; ++k) asm volatile("" : "+v"(wd[k].x), "+v"(wd[k].y));
  int v[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const uint32_t b = 1u << (key[k] & 31);
    int r = (wd[k].x & b) ? (int)(wd[k].y + __popc(wd[k].x & (b - 1u))) : -1;
    if (r >= m || !((okmask >> k) & 1u)) r = -1;
    v[k] = r;
  }
  v[13] = live ? row : -1;
  slab::slab_emit<BM>(v, blk, t, hdr, slots, status, fmt);
}


// ---- sorted-key lookups: a voxel set whose rows are in ascending linear index is its own index -----------------------------
// (the voxelizer's key-order output, every level a strided convolution produced).  keys[i] = linear index of row i;
// xstart[b*X + x] = first row of x-plane (b, x), xstart[B*X] = n: a lookup is a binary search inside ONE x-plane's segment
// (~100 rows at level 1), whose keys the neighbouring rows of a workgroup keep in L1.  No hash insert, no table to clear.
// status bit 1 (value 2): rows are NOT in strictly ascending order (the caller's promise is broken).
__global__ __launch_bounds__(256) void sp_sorted_keys_kernel(const int* __restrict__ indices, int n_cap,
                                                             const int* __restrict__ n_dev, ConvGeom g,
                                                             uint32_t* __restrict__ keys, int* __restrict__ xstart,
                                                             int* __restrict__ status) {
  int n = n_dev ? *n_dev : n_cap;
  if (n > n_cap) n = n_cap;
  const int nplanes = g.batch * g.in_shape[0];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (n <= 0) {   // no rows: every plane is empty
    for (int j = i; j <= nplanes; j += gridDim.x * 256) xstart[j] = 0;
    return;
  }
  if (i >= n) return;
  const int4 c = ((const int4*)indices)[i];
  const int bx = c.x * g.in_shape[0] + c.y;
  const uint32_t key = ((uint32_t)bx * (uint32_t)g.in_shape[1] + (uint32_t)c.z) * (uint32_t)g.in_shape[2] + (uint32_t)c.w;
  keys[i] = key;
  // a coordinate outside the grid breaks the promise just like an unsorted row does (the key arithmetic above wraps)
  bool bad = c.x < 0 || c.x >= g.batch || c.y < 0 || c.y >= g.in_shape[0] || c.z < 0 || c.z >= g.in_shape[1] || c.w < 0 ||
             c.w >= g.in_shape[2];
  int prev = -1;
  if (i > 0) {
    const int4 p = ((const int4*)indices)[i - 1];
    prev = p.x * g.in_shape[0] + p.y;
    const uint32_t pk = ((uint32_t)prev * (uint32_t)g.in_shape[1] + (uint32_t)p.z) * (uint32_t)g.in_shape[2] + (uint32_t)p.w;
    bad = bad || pk >= key;
  }
  if (bad && status) atomicOr(status, 2);
  // Directory writes stay inside [0, nplanes] whatever the coordinates hold.  With a broken promise some entries may stay
  // unwritten (prev > top): readers clamp what they load (sp_slab_from_sorted_kernel), so the result is wrong (and flagged:
  // status bit 1), never a fault.
  prev = prev < -1 ? -1 : prev > nplanes ? nplanes : prev;
  const int top = bx < -1 ? -1 : bx < nplanes ? bx : nplanes;
  for (int j = prev + 1; j <= top; ++j) xstart[j] = i;
  if (i == n - 1)
    for (int j = top + 1; j <= nplanes; ++j) xstart[j] = n;
}

// first position in [lo, hi) whose key is >= t
__device__ __forceinline__ int sorted_lower_bound(const uint32_t* __restrict__ keys, int lo, int hi, uint32_t t) {
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < t) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// Slab metadata (slab_emit: per block of BM output rows and kernel plane kx the input range, + 16-bit slots) of a 3x3x3
// convolution — submanifold OR strided — whose INPUT set is in ascending linear index, found by sorted-key search.  The nine
// (ky, kz) taps of a kernel plane kx read a window of 3 y-columns x 3 z-cells of ONE input x-plane: in linear order that is a
// key interval of 2*Z + 3 cells holding a handful of rows.  Thread t therefore does ONE lower_bound per plane (inside the
// x-plane's segment of the directory) and then walks the rows up to the window's last key, four keys per round trip, turning
// each into its (ky, kz) tap — 3 searches per row instead of one lookup per tap.  Output rows must be in ascending linear
// index too (what makes a plane's inputs a contiguous range).  SUBM: the centre tap is the row itself.
// (Round 4: searching the three planes TOGETHER — one lower_bound step of each per round trip, twelve window keys in flight — was
// slower, 85 / 147 us against 64 / 114: the kernel is bound by instruction issue, not by the latency of its dependent loads.)
template <int BM, bool SUBM>
__global__ __launch_bounds__(BM) void sp_slab_from_sorted_kernel(const int* __restrict__ out_indices, int m_cap,
                                                                 const int* __restrict__ m_dev, ConvGeom g,
                                                                 const uint32_t* __restrict__ in_keys,
                                                                 const int* __restrict__ in_xstart, int in_n_cap,
                                                                 int2* __restrict__ hdr, uint16_t* __restrict__ slots,
                                                                 int* __restrict__ status, int fmt) {
  int m = m_dev ? *m_dev : m_cap;
  if (m > m_cap) m = m_cap;
  const int nblk = (m + BM - 1) / BM, per = (nblk + 7) >> 3;
  const int xcd = (int)blockIdx.x & 7, bix = (int)blockIdx.x >> 3;
  const int blk = (gridDim.x & 7u) == 0 ? xcd * per + bix : (int)blockIdx.x;
  if (((gridDim.x & 7u) == 0 && bix >= per) || blk >= nblk) return;
  const int t = threadIdx.x, row = blk * BM + t;
  const bool live = row < m;
  const int4 c = live ? ((const int4*)out_indices)[row] : make_int4(0, 0, 0, 0);
  const int X = g.in_shape[0], Y = g.in_shape[1], Z = g.in_shape[2];
  const int x0 = c.y * g.stride[0] - g.pad[0], y0 = c.z * g.stride[1] - g.pad[1], z0 = c.w * g.stride[2] - g.pad[2];
  const int ylo = y0 < 0 ? 0 : y0, yhi = y0 + 2 < Y ? y0 + 2 : Y - 1;
  const int zlo = z0 < 0 ? 0 : z0, zhi = z0 + 2 < Z ? z0 + 2 : Z - 1;
  const bool win = live && ylo <= yhi && zlo <= zhi;
  int v[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) v[k] = -1;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int x = x0 + kx;
    if (!(win && x >= 0 && x < X && c.x >= 0 && c.x < g.batch)) continue;
    const int bx = c.x * X + x;
    // Clamped to the key array: a directory built from rows that were NOT in linear order (status bit 1 of the index) may hold
    // anything in the entries it never wrote; the search below must not leave in_keys[0, in_n_cap).
    int q = in_xstart[bx];
    int hi = in_xstart[bx + 1];
    q = q < 0 ? 0 : q;
    hi = hi > in_n_cap ? in_n_cap : hi;
    if (q >= hi) continue;
    const uint32_t plane = (uint32_t)bx * (uint32_t)Y * (uint32_t)Z;
    const uint32_t kmax = plane + (uint32_t)(yhi * Z + zhi);
    q = sorted_lower_bound(in_keys, q, hi, plane + (uint32_t)(ylo * Z + zlo));
    for (bool more = q < hi; more;) {
      uint32_t key[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) key[u] = in_keys[q + u < hi ? q + u : hi - 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!more) break;
        if (q + u >= hi || key[u] > kmax) { more = false; break; }
        const int r = (int)(key[u] - plane) - y0 * Z;   // (y - y0) * Z + z, y - y0 in [0, 2]
        const int ky = r >= 2 * Z ? 2 : r >= Z ? 1 : 0;
        const int kz = r - ky * Z - z0;
        if (kz >= 0 && kz < 3) {
          const int tap = ky * 3 + kz;
#pragma unroll
          for (int d = 0; d < 9; ++d) v[kx * 9 + d] = tap == d ? q + u : v[kx * 9 + d];
        }
      }
      q += 4;
    }
  }
  if (SUBM) v[13] = live ? row : -1;
  slab::slab_emit<BM>(v, blk, t, hdr, slots, status, fmt);
}

}  // namespace bevamd

using namespace bevamd;

extern "C" {

/* ---- granular, sync-free building blocks ------------------------------------------------------------- */

size_t bevamd_spconv_hash_index_bytes(int n_cap) { return (size_t)hash_capacity((size_t)(n_cap > 0 ? n_cap : 1)) * 8; }

size_t bevamd_spconv_rank_index_bytes(int batch_size, const int* shape) {
  if (!shape || batch_size <= 0) return 0;
  return rank_index_bytes(batch_size, shape);
}

int bevamd_spconv_hash_index_build(const int* indices, int n_cap, const int* n_dev, int batch_size, const int* shape,
                                   void* index, size_t index_bytes, void* stream_) {
  ConvGeom g;
  const int one[3] = {1, 1, 1}, zero[3] = {0, 0, 0};
  int rc = make_geom(batch_size, shape, shape, one, one, zero, nullptr, 1, g);
  if (rc) return rc;
  BEVAMD_REQUIRE(n_cap >= 0 && (indices || n_cap == 0), "spconv_hash_index_build: bad input");
  return hash_build(indices, n_cap, n_dev, g, index, index_bytes, (hipStream_t)stream_);
}

int bevamd_spconv_downsample(const int* indices, int n_cap, const int* n_dev, int batch_size, const int* in_shape,
                             const int* out_shape, const int* ksize, const int* stride, const int* padding,
                             int* out_indices, int out_cap, int* num_out_dev, void* out_index, size_t out_index_bytes,
                             int* nbr, int nbr_stride, void* stream_) {
  ConvGeom g;
  int rc = make_geom(batch_size, in_shape, out_shape, ksize, stride, padding, nullptr, 0, g);
  if (rc) return rc;
  BEVAMD_REQUIRE(n_cap >= 0 && (indices || n_cap == 0), "spconv_downsample: bad input");
  BEVAMD_REQUIRE(num_out_dev && out_indices && out_cap >= 1, "spconv_downsample: null output / out_cap < 1");
  BEVAMD_REQUIRE(!nbr || nbr_stride >= out_cap, "spconv_downsample: nbr_stride %d < out_cap %d", nbr_stride, out_cap);
  return downsample(indices, n_cap, n_dev, g, out_indices, out_cap, num_out_dev, out_index, out_index_bytes, nbr,
                    nbr_stride, (hipStream_t)stream_);
}

/* bevamd_spconv_downsample for an input set whose rows are in ASCENDING LINEAR INDEX and that owns a lookup structure:
 * src_kind 0 = the x-plane directory of its sorted-key index (bevamd_spconv_sorted_index_build: int32 [batch * in_shape[0] + 1]),
 * src_kind 1 = its rank index words.  Same outputs (out_indices, num_out_dev, the outputs' rank index in out_index) from two
 * launches instead of four and without the 32-byte-per-word byte map: see sp_rank_tiles_from_rows_kernel.  Rows that are not
 * in linear order give a wrong (never out-of-bounds) result; the sorted-key index flags that case. */
int bevamd_spconv_downsample_sorted(const int* indices, int n_cap, const int* n_dev, int batch_size, const int* in_shape,
                                    const int* out_shape, const int* ksize, const int* stride, const int* padding, int src_kind,
                                    const void* src, int* out_indices, int out_cap, int* num_out_dev, void* out_index,
                                    size_t out_index_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom g;
  int rc = make_geom(batch_size, in_shape, out_shape, ksize, stride, padding, nullptr, 0, g);
  if (rc) return rc;
  BEVAMD_REQUIRE(n_cap >= 0 && (indices || n_cap == 0), "spconv_downsample_sorted: bad input");
  BEVAMD_REQUIRE(num_out_dev && out_indices && out_cap >= 1, "spconv_downsample_sorted: null output / out_cap < 1");
  BEVAMD_REQUIRE((src_kind == 0 || src_kind == 1) && src, "spconv_downsample_sorted: src_kind %d (0 | 1) / null src", src_kind);
  BEVAMD_REQUIRE(ksize[0] <= 3 && ksize[1] <= 3 && ksize[2] <= 3, "spconv_downsample_sorted: kernel sizes up to 3 (got %d %d %d)",
                 ksize[0], ksize[1], ksize[2]);
  const size_t need = rank_index_bytes(g.batch, g.out_shape);
  if (!out_index || out_index_bytes < need) {
    set_error("spconv rank index: buffer too small (%zu < %zu)", out_index_bytes, need);
    return BEVAMD_ERR_WORKSPACE;
  }
  const size_t nw = grid_words(g.batch, g.out_shape), nt = rank_tiles(nw);
  Carver cv(out_index, out_index_bytes);
  uint2* words = cv.take<uint2>(nw);   // must stay first: the rank index IS this array
  uint32_t* tile_sums = cv.take<uint32_t>(nt + 1);
  const int* xs = src_kind == 0 ? (const int*)src : nullptr;
  const uint2* iw = src_kind == 1 ? (const uint2*)src : nullptr;
#define BEVAMD_GO(TILE)                                                                                                       \
  do {                                                                                                                        \
    if (src_kind == 0) sp_rank_tiles_from_rows_kernel<TILE, 0><<<dim3((unsigned)nt), dim3(256), 0, stream>>>(indices, n_cap, n_dev, g, xs, iw, words, nw, tile_sums); \
    else sp_rank_tiles_from_rows_kernel<TILE, 1><<<dim3((unsigned)nt), dim3(256), 0, stream>>>(indices, n_cap, n_dev, g, xs, iw, words, nw, tile_sums); \
    BEVAMD_LAUNCH_CHECK("sp_rank_tiles_from_rows");                                                                           \
    sp_rank_apply_emit_kernel<TILE><<<dim3((unsigned)nt), dim3(256), 0, stream>>>(words, nw, tile_sums, num_out_dev, g, out_indices, out_cap); \
    BEVAMD_LAUNCH_CHECK("sp_rank_apply_emit");                                                                                \
  } while (0)
  if (rank_tile_for(nw) == RANK_TILE_SMALL) BEVAMD_GO(RANK_TILE_SMALL);
  else BEVAMD_GO(RANK_TILE);
#undef BEVAMD_GO
  return BEVAMD_OK;
}

int bevamd_spconv_neighbors(const int* out_indices, int m_cap, const int* m_dev, int batch_size, const int* in_shape,
                            const int* out_shape, const int* ksize, const int* stride, const int* padding, int subm,
                            int index_kind, const void* in_index, int in_index_n_cap, int* nbr, int nbr_stride,
                            void* stream_) {
  ConvGeom g;
  int rc = make_geom(batch_size, in_shape, out_shape, ksize, stride, padding, nullptr, subm, g);
  if (rc) return rc;
  BEVAMD_REQUIRE(index_kind == INDEX_HASH || index_kind == INDEX_RANK, "spconv_neighbors: index_kind %d", index_kind);
  BEVAMD_REQUIRE(m_cap >= 0 && nbr_stride >= m_cap, "spconv_neighbors: nbr_stride %d < m_cap %d", nbr_stride, m_cap);
  if (m_cap == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(out_indices && in_index && nbr, "spconv_neighbors: null buffer");
  const IndexRef ix = index_kind == INDEX_HASH ? hash_ref(in_index, in_index_n_cap, batch_size) : rank_ref(in_index);
  return neighbors(out_indices, m_cap, m_dev, g, index_kind, ix, nbr, nbr_stride, subm != 0, (hipStream_t)stream_);
}

int bevamd_spconv_dense_bev(const void* features, int elem_bytes, int pitch, int channels, int index_kind,
                            const void* index, int index_n_cap, int batch_size, const int* shape, void* out,
                            void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(elem_bytes == 2 || elem_bytes == 4, "spconv_dense_bev: elem_bytes %d", elem_bytes);
  BEVAMD_REQUIRE(index_kind == INDEX_HASH || index_kind == INDEX_RANK, "spconv_dense_bev: index_kind %d", index_kind);
  BEVAMD_REQUIRE(shape && batch_size > 0 && channels > 0 && pitch >= channels, "spconv_dense_bev: bad sizes");
  BEVAMD_REQUIRE(features && index && out, "spconv_dense_bev: null buffer");
  const int X = shape[0], Y = shape[1], Z = shape[2];
  BEVAMD_REQUIRE(X > 0 && Y > 0 && Z > 0 && Z <= 512 && X <= 65535 && batch_size <= 65535, "spconv_dense_bev: bad shape");
  BEVAMD_REQUIRE((unsigned long long)batch_size * X * Y * Z < 0xFFFFFFF0ull, "spconv_dense_bev: batch * volume must be < 2^32 - 16");
  const IndexRef ix = index_kind == INDEX_HASH ? hash_ref(index, index_n_cap, batch_size) : rank_ref(index);
  dim3 grid(cdiv(Y, 64), X, batch_size), block(256);
  size_t lds = (size_t)64 * Z * sizeof(int);
  const size_t row_bytes = (size_t)channels * elem_bytes;
  const size_t staged = lds + (size_t)64 * Z * (row_bytes + 4);
  const bool stage = ((uintptr_t)out & 7) == 0 && row_bytes % 4 == 0 && ((size_t)pitch * elem_bytes) % 4 == 0 && ((uintptr_t)features & 3) == 0 && staged <= 64 * 1024;
  const bool wide = row_bytes % 16 == 0 && row_bytes / 16 <= 64 && 64 % (row_bytes / 16) == 0 && ((size_t)pitch * elem_bytes) % 16 == 0 && ((uintptr_t)features & 15) == 0;
#define BEVAMD_DENSE(T, KIND)                                                                                               \
  do {                                                                                                                      \
    if (stage && wide) sp_dense_bev_staged_kernel<T, KIND, true><<<grid, block, staged, stream>>>((const T*)features, pitch, channels, ix, X, Y, Z, (T*)out); \
    else if (stage) sp_dense_bev_staged_kernel<T, KIND><<<grid, block, staged, stream>>>((const T*)features, pitch, channels, ix, X, Y, Z, (T*)out); \
    else sp_dense_bev_kernel<T, KIND><<<grid, block, lds, stream>>>((const T*)features, pitch, channels, ix, X, Y, Z, (T*)out);          \
  } while (0)
  if (elem_bytes == 2) {
    if (index_kind == INDEX_HASH) BEVAMD_DENSE(uint16_t, INDEX_HASH); else BEVAMD_DENSE(uint16_t, INDEX_RANK);
  } else {
    if (index_kind == INDEX_HASH) BEVAMD_DENSE(uint32_t, INDEX_HASH); else BEVAMD_DENSE(uint32_t, INDEX_RANK);
  }
#undef BEVAMD_DENSE
  BEVAMD_LAUNCH_CHECK("sp_dense_bev");
  return BEVAMD_OK;
}

/* ---- one-call rulebook (any input row order; optional host count) ------------------------------------- */

// workspace for bevamd_spconv_build_rulebook (n = number of active inputs)
size_t bevamd_spconv_rulebook_workspace_bytes(int n, int batch_size, const int* out_shape, int subm) {
  if (n < 1) n = 1;
  size_t b = align_up((size_t)hash_capacity((size_t)n) * 8, 256);
  if (!subm && out_shape && batch_size > 0) b += rank_index_bytes(batch_size, out_shape);
  return b + 1024;
}

/* max number of output rows a strided conv can activate for n inputs (size of out_indices / nbr rows) */
int bevamd_spconv_max_outputs(int n, const int* ksize, const int* stride, int subm) {
  if (subm) return n;
  long long bound = 1;
  for (int i = 0; i < 3; ++i) bound *= (ksize[i] + stride[i] - 1) / stride[i];
  long long m = (long long)n * bound;
  return m > 0x7FFFFFF0ll ? 0x7FFFFFF0 : (int)m;
}

/* same bound for any dilation / a transposed convolution (one input reaches at most prod(ksize) outputs then) */
int bevamd_spconv_max_outputs_ex(int n, const int* ksize, const int* stride, const int* dilation, int subm, int transpose) {
  if (subm) return n;
  long long bound = 1;
  for (int i = 0; i < 3; ++i)
    bound *= (transpose || (dilation && dilation[i] != 1)) ? ksize[i] : (ksize[i] + stride[i] - 1) / stride[i];
  long long m = (long long)n * bound;
  return m > 0x7FFFFFF0ll ? 0x7FFFFFF0 : (int)m;
}

int bevamd_spconv_build_rulebook(const int* indices, int n, int batch_size, const int* in_shape,
                                 const int* out_shape, const int* ksize, const int* stride, const int* padding,
                                 const int* dilation, int subm, int transpose, int* out_indices, int out_cap, int* nbr,
                                 int nbr_stride, int* num_out_dev, int* num_out_host, void* ws, size_t ws_bytes,
                                 void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom g;
  int rc = make_geom(batch_size, in_shape, out_shape, ksize, stride, padding, dilation, subm, g, transpose);
  if (rc) return rc;
  BEVAMD_REQUIRE(n >= 0, "spconv_build_rulebook: n < 0");
  BEVAMD_REQUIRE(num_out_dev != nullptr, "spconv_build_rulebook: num_out_dev is null");
  if (n == 0) {
    BEVAMD_HIP_CHECK(hipMemsetAsync(num_out_dev, 0, sizeof(int), stream));
    if (num_out_host) { BEVAMD_HIP_CHECK(hipStreamSynchronize(stream)); *num_out_host = 0; }
    return BEVAMD_OK;
  }
  BEVAMD_REQUIRE(indices && nbr && (subm || out_indices), "spconv_build_rulebook: null buffer");
  size_t need = bevamd_spconv_rulebook_workspace_bytes(n, batch_size, g.out_shape, subm);
  if (!ws || ws_bytes < need) {
    set_error("spconv_build_rulebook: workspace too small (%zu < %zu)", ws_bytes, need);
    return BEVAMD_ERR_WORKSPACE;
  }
  const size_t hbytes = align_up((size_t)hash_capacity((size_t)n) * 8, 256);

  if (subm) {
    BEVAMD_REQUIRE(nbr_stride >= n, "spconv_build_rulebook: nbr_stride %d < n %d", nbr_stride, n);
    rc = hash_build(indices, n, nullptr, g, ws, hbytes, stream);
    if (rc) return rc;
    rc = neighbors(indices, n, nullptr, g, INDEX_HASH, hash_ref(ws, n, batch_size), nbr, nbr_stride, true, stream);
    if (rc) return rc;
    if (out_indices && out_indices != indices)
      BEVAMD_HIP_CHECK(hipMemcpyAsync(out_indices, indices, (size_t)n * 4 * sizeof(int), hipMemcpyDeviceToDevice, stream));
    sp_set_int_kernel<<<1, 1, 0, stream>>>(num_out_dev, n);
    BEVAMD_LAUNCH_CHECK("sp_set_int");
    if (num_out_host) *num_out_host = n;  // known without asking the device
    return BEVAMD_OK;
  }

  BEVAMD_REQUIRE((long long)out_cap >= 1, "spconv_build_rulebook: out_cap must be >= 1");
  // rows are bounded by out_cap on the launch side and by *num_out_dev on the device side
  const long long most = (long long)n * conv_bound(g);
  const int m_cap = out_cap < most ? out_cap : (int)most;
  BEVAMD_REQUIRE(nbr_stride >= m_cap, "spconv_build_rulebook: nbr_stride %d < out_cap %d", nbr_stride, m_cap);
  rc = downsample(indices, n, nullptr, g, out_indices, m_cap, num_out_dev, (char*)ws + hbytes, ws_bytes - hbytes, nbr,
                  nbr_stride, stream);
  if (rc) return rc;
  if (num_out_host) {
    BEVAMD_HIP_CHECK(hipMemcpyAsync(num_out_host, num_out_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
    BEVAMD_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return BEVAMD_OK;
}

size_t bevamd_spconv_pairs_workspace_bytes(int m, int kernel_volume) {
  size_t e = (size_t)(m > 0 ? m : 1) * (size_t)(kernel_volume > 0 ? kernel_volume : 1);
  return 2 * align_up(e * 4, 256) + scan_workspace_bytes(e) + 512;
}

/* reference-shaped rulebook from nbr: indice_pairs [K,2,pairs_len] (filled with -1 first), indice_num [K] */
int bevamd_spconv_pairs_from_nbr(const int* nbr, int nbr_stride, int m, int kernel_volume, int* indice_pairs,
                                 int pairs_len, int* indice_num, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(kernel_volume > 0 && m >= 0 && pairs_len >= 0, "spconv_pairs_from_nbr: bad sizes");
  BEVAMD_REQUIRE(indice_pairs && indice_num, "spconv_pairs_from_nbr: null output");
  BEVAMD_HIP_CHECK(hipMemsetAsync(indice_pairs, 0xFF, (size_t)kernel_volume * 2 * pairs_len * sizeof(int), stream));
  BEVAMD_HIP_CHECK(hipMemsetAsync(indice_num, 0, (size_t)kernel_volume * sizeof(int), stream));
  if (m == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(nbr != nullptr, "spconv_pairs_from_nbr: nbr is null");
  size_t need = bevamd_spconv_pairs_workspace_bytes(m, kernel_volume);
  if (!ws || ws_bytes < need) {
    set_error("spconv_pairs_from_nbr: workspace too small (%zu < %zu)", ws_bytes, need);
    return BEVAMD_ERR_WORKSPACE;
  }
  Carver cv(ws, ws_bytes);
  const size_t e = (size_t)m * kernel_volume;
  uint32_t* flags = cv.take<uint32_t>(e);
  uint32_t* scan = cv.take<uint32_t>(e);
  void* sws = cv.base + cv.off;
  dim3 grid(cdiv(m, 256), kernel_volume), block(256);
  sp_pair_flags_kernel<<<grid, block, 0, stream>>>(nbr, nbr_stride, m, kernel_volume, flags);
  BEVAMD_LAUNCH_CHECK("sp_pair_flags");
  int rc = exclusive_scan_u32(flags, scan, e, nullptr, sws, ws_bytes - cv.off, stream);
  if (rc) return rc;
  sp_pairs_scatter_kernel<<<grid, block, 0, stream>>>(nbr, nbr_stride, m, kernel_volume, scan, indice_pairs, pairs_len,
                                                      indice_num);
  BEVAMD_LAUNCH_CHECK("sp_pairs_scatter");
  return BEVAMD_OK;
}

/* nbr [K, nbr_stride] (filled with -1 first) from reference-shaped pairs; in_col = 0 normally, 1 for the
 * "inverse" convolution (spconv_ops.h:317,348 swap the two pair columns). */
int bevamd_spconv_nbr_from_pairs(const int* indice_pairs, int pairs_len, const int* indice_num, int kernel_volume,
                                 int inverse, int* nbr, int nbr_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(kernel_volume > 0 && pairs_len >= 0 && nbr_stride >= 0, "spconv_nbr_from_pairs: bad sizes");
  BEVAMD_REQUIRE(nbr || nbr_stride == 0, "spconv_nbr_from_pairs: nbr is null");
  if (nbr_stride > 0) BEVAMD_HIP_CHECK(hipMemsetAsync(nbr, 0xFF, (size_t)kernel_volume * nbr_stride * sizeof(int), stream));
  if (pairs_len == 0 || nbr_stride == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(indice_pairs && indice_num, "spconv_nbr_from_pairs: null input");
  sp_nbr_from_pairs_kernel<<<dim3(cdiv(pairs_len, 256), kernel_volume), dim3(256), 0, stream>>>(
      indice_pairs, pairs_len, indice_num, kernel_volume, inverse ? 1 : 0, nbr, nbr_stride);
  BEVAMD_LAUNCH_CHECK("sp_nbr_from_pairs");
  return BEVAMD_OK;
}

/* nbr_t [K, nbr_t_stride] (filled with -1 first): nbr_t[k][nbr[k][o]] = o.  n_in rows. */
int bevamd_spconv_transpose_nbr(const int* nbr, int nbr_stride, int m, int kernel_volume, int* nbr_t,
                                int nbr_t_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(kernel_volume > 0 && m >= 0 && nbr_t_stride >= 0, "spconv_transpose_nbr: bad sizes");
  if (nbr_t_stride > 0) {
    BEVAMD_REQUIRE(nbr_t != nullptr, "spconv_transpose_nbr: nbr_t is null");
    BEVAMD_HIP_CHECK(hipMemsetAsync(nbr_t, 0xFF, (size_t)kernel_volume * nbr_t_stride * sizeof(int), stream));
  }
  if (m == 0 || nbr_t_stride == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(nbr != nullptr, "spconv_transpose_nbr: nbr is null");
  sp_transpose_nbr_kernel<<<dim3(cdiv(m, 256), kernel_volume), dim3(256), 0, stream>>>(nbr, nbr_stride, m, nbr_t,
                                                                                       nbr_t_stride);
  BEVAMD_LAUNCH_CHECK("sp_transpose_nbr");
  return BEVAMD_OK;
}

/* Slab metadata (bevamd_spconv_slab_build's hdr / slots, same layout and contents) of the 3x3x3 submanifold convolution
 * over the voxel set `indices` [m, 4] (b, x, y, z) on grid `shape`, computed from the set's own index (index_kind /
 * index / index_n_cap as for bevamd_spconv_neighbors) without building the neighbour table.  The caller vouches that the
 * rows are in ascending linear index (what the slab kernels need anyway). */
int bevamd_spconv_slab_build_from_index(const int* indices, int m_cap, const int* m_dev, int batch_size, const int* shape,
                                        int index_kind, const void* index, int index_n_cap, int block_rows, void* hdr,
                                        void* slots, int* status, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom g;
  const int k3[3] = {3, 3, 3}, one[3] = {1, 1, 1}, zero[3] = {0, 0, 0};
  int rc = make_geom(batch_size, shape, shape, k3, one, zero, nullptr, 1, g);
  if (rc) return rc;
  const int fmt = slab::fmt_of_code(block_rows);   // upper half: slot format (spconv_slab_meta.h), 0 = raw / implied by 64-row blocks
  block_rows = slab::rows_of_code(block_rows);
  BEVAMD_REQUIRE(block_rows == 64 || block_rows == 128 || block_rows == 256, "spconv_slab_build_from_index: block_rows %d (64 | 128 | 256)", block_rows);
  BEVAMD_REQUIRE(fmt == 0 || fmt == slab::FMT_BAKED128 || (slab::fmt_is_wg(fmt) && block_rows == 128), "spconv_slab_build_from_index: slot format %d (the filter-gradient formats need 128-row blocks)", fmt);
  BEVAMD_REQUIRE(index_kind == INDEX_HASH || index_kind == INDEX_RANK, "spconv_slab_build_from_index: index_kind %d", index_kind);
  BEVAMD_REQUIRE(m_cap >= 0, "spconv_slab_build_from_index: bad sizes");
  if (m_cap == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(indices && index && hdr && slots, "spconv_slab_build_from_index: null buffer");
  const IndexRef ix = index_kind == INDEX_HASH ? hash_ref(index, index_n_cap, batch_size) : rank_ref(index);
  const unsigned nblk = (unsigned)((m_cap + block_rows - 1) / block_rows + 7) / 8 * 8;   // a multiple of 8: XCD-contiguous block map
#define BEVAMD_GO(BM, KIND) \
  sp_slab_from_index_kernel<BM, KIND><<<dim3(nblk), dim3(BM), 0, stream>>>(indices, m_cap, m_dev, g, ix, (int2*)hdr, (uint16_t*)slots, status, fmt)
#define BEVAMD_GO_RANK(BM) \
  sp_slab_from_rank_kernel<BM><<<dim3(nblk), dim3(BM), 0, stream>>>(indices, m_cap, m_dev, g, ix.words, (int2*)hdr, (uint16_t*)slots, status, fmt)
  if (block_rows == 64) { if (index_kind == INDEX_HASH) BEVAMD_GO(64, INDEX_HASH); else BEVAMD_GO_RANK(64); }
  else if (block_rows == 128) { if (index_kind == INDEX_HASH) BEVAMD_GO(128, INDEX_HASH); else BEVAMD_GO_RANK(128); }
  else { if (index_kind == INDEX_HASH) BEVAMD_GO(256, INDEX_HASH); else BEVAMD_GO_RANK(256); }
#undef BEVAMD_GO_RANK
#undef BEVAMD_GO
  BEVAMD_LAUNCH_CHECK("sp_slab_from_index");
  return BEVAMD_OK;
}

/* Sorted-key index of a voxel set whose rows are in ascending linear index (b, x, y, z): keys [n_cap] uint32 and the
 * x-plane directory xstart [batch * shape[0] + 1] int32 (bevamd_spconv_sorted_index_bytes: both, keys first, directory at
 * byte offset align256(n_cap * 4)).  status (optional int32, device): bit 1 is set when the rows are not strictly ascending. */
size_t bevamd_spconv_sorted_index_bytes(int n_cap, int batch_size, const int* shape) {
  if (!shape || batch_size <= 0 || n_cap < 0) return 0;
  return align_up((size_t)(n_cap > 0 ? n_cap : 1) * 4, 256) + align_up(((size_t)batch_size * shape[0] + 1) * 4, 256);
}

int bevamd_spconv_sorted_index_build(const int* indices, int n_cap, const int* n_dev, int batch_size, const int* shape,
                                     void* index, size_t index_bytes, int* status, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom g;
  const int one[3] = {1, 1, 1}, zero[3] = {0, 0, 0};
  int rc = make_geom(batch_size, shape, shape, one, one, zero, nullptr, 1, g);
  if (rc) return rc;
  BEVAMD_REQUIRE(n_cap >= 0 && (indices || n_cap == 0), "spconv_sorted_index_build: bad input");
  const size_t need = bevamd_spconv_sorted_index_bytes(n_cap, batch_size, shape);
  if (!index || index_bytes < need) {
    set_error("spconv sorted index: buffer too small (%zu < %zu)", index_bytes, need);
    return BEVAMD_ERR_WORKSPACE;
  }
  uint32_t* keys = (uint32_t*)index;
  int* xstart = (int*)((char*)index + align_up((size_t)(n_cap > 0 ? n_cap : 1) * 4, 256));
  sp_sorted_keys_kernel<<<dim3(cdiv(n_cap > 0 ? n_cap : 1, 256)), dim3(256), 0, stream>>>(indices, n_cap, n_dev, g, keys, xstart, status);
  BEVAMD_LAUNCH_CHECK("sp_sorted_keys");
  return BEVAMD_OK;
}

/* Slab metadata (hdr / slots of bevamd_spconv_slab_build, same layout) of a 3x3x3 convolution from the SORTED-KEY index of its
 * input set: subm != 0 -> the submanifold convolution over the set itself (out_indices = the set, stride / padding ignored);
 * else the strided convolution whose active outputs are out_indices [m_cap, 4] on out_shape (rows in ascending linear index,
 * as bevamd_spconv_downsample emits them).  Replaces the neighbour table (108 B per row written and read back) for layers that
 * run on the slab kernels. */
int bevamd_spconv_slab_build_from_sorted(const int* out_indices, int m_cap, const int* m_dev, int batch_size,
                                         const int* in_shape, const int* out_shape, const int* stride, const int* padding,
                                         int subm, const void* in_index, int in_n_cap, int block_rows, void* hdr, void* slots,
                                         int* status, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  ConvGeom g;
  const int k3[3] = {3, 3, 3};
  int rc = make_geom(batch_size, in_shape, subm ? in_shape : out_shape, k3, stride, padding, nullptr, subm, g);
  if (rc) return rc;
  const int fmt = slab::fmt_of_code(block_rows);   // upper half: slot format (spconv_slab_meta.h), 0 = raw / implied by 64-row blocks
  block_rows = slab::rows_of_code(block_rows);
  BEVAMD_REQUIRE(block_rows == 64 || block_rows == 128 || block_rows == 256, "spconv_slab_build_from_sorted: block_rows %d (64 | 128 | 256)", block_rows);
  BEVAMD_REQUIRE(fmt == 0 || fmt == slab::FMT_BAKED128 || (slab::fmt_is_wg(fmt) && block_rows == 128), "spconv_slab_build_from_sorted: slot format %d (the filter-gradient formats need 128-row blocks)", fmt);
  BEVAMD_REQUIRE(m_cap >= 0 && in_n_cap >= 0, "spconv_slab_build_from_sorted: bad sizes");
  if (m_cap == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(out_indices && in_index && hdr && slots, "spconv_slab_build_from_sorted: null buffer");
  const uint32_t* keys = (const uint32_t*)in_index;
  const int* xstart = (const int*)((const char*)in_index + align_up((size_t)(in_n_cap > 0 ? in_n_cap : 1) * 4, 256));
  const unsigned nblk = (unsigned)((m_cap + block_rows - 1) / block_rows + 7) / 8 * 8;
#define BEVAMD_GO(BM, SUBM) \
  sp_slab_from_sorted_kernel<BM, SUBM><<<dim3(nblk), dim3(BM), 0, stream>>>(out_indices, m_cap, m_dev, g, keys, xstart, in_n_cap, (int2*)hdr, (uint16_t*)slots, status, fmt)
  if (block_rows == 64) { if (subm) BEVAMD_GO(64, true); else BEVAMD_GO(64, false); }
  else if (block_rows == 128) { if (subm) BEVAMD_GO(128, true); else BEVAMD_GO(128, false); }
  else { if (subm) BEVAMD_GO(256, true); else BEVAMD_GO(256, false); }
#undef BEVAMD_GO
  BEVAMD_LAUNCH_CHECK("sp_slab_from_sorted");
  return BEVAMD_OK;
}

}  // extern "C"
