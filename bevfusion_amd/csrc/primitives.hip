// Device-wide primitives used by bev_pool precompute, hard voxelization and the
// spconv rulebook builder: exclusive scan and a stable LSD radix sort, written
// for wave64 (ballot-based match inside a wave, LDS counters across the 4 waves
// of a 256-thread workgroup).  Integer-only, HBM/L2-bound; no MFMA here.
#include "common.h"
#include "single_pass.h"

#include <stdarg.h>
#include <stdio.h>

namespace bevamd {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }

// ============================================================================
// exclusive scan
// ============================================================================
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // 2048

// Exclusive scan of one value per thread across a 256-thread block.
// Returns the exclusive prefix; *block_total gets the block sum (all threads).
__device__ __forceinline__ unsigned block_exclusive_scan_256(unsigned v, unsigned* lds_wave /*[4]*/,
                                                             unsigned* block_total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned inc = wave_inclusive_scan(v);
  if (lane == 63) lds_wave[wave] = inc;
  __syncthreads();
  unsigned w0 = lds_wave[0], w1 = lds_wave[1], w2 = lds_wave[2], w3 = lds_wave[3];
  unsigned base = (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u);
  *block_total = w0 + w1 + w2 + w3;
  __syncthreads();
  return base + inc - v;
}

// Sum of sums[0 .. upto) by one 256-thread workgroup (every thread gets it).  With at most a few thousand tiles this is
// cheaper than a separate single-workgroup scan launch between the reduce and apply passes.
__device__ __forceinline__ unsigned block_prefix_of_tiles(const uint32_t* __restrict__ sums, unsigned upto,
                                                          unsigned* lds_wave /*[4]*/) {
  unsigned part = 0;
  for (unsigned t = threadIdx.x; t < upto; t += SCAN_THREADS) part += sums[t];
  part = (unsigned)wave_reduce_add((int)part);
  if ((threadIdx.x & 63) == 0) lds_wave[threadIdx.x >> 6] = part;
  __syncthreads();
  const unsigned r = lds_wave[0] + lds_wave[1] + lds_wave[2] + lds_wave[3];
  __syncthreads();
  return r;
}
constexpr size_t SCAN_INLINE_TILES_MAX = 8192;  // above this the tile sums get their own scan launch

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_reduce_kernel(const uint32_t* __restrict__ in,
                                                                        uint32_t* __restrict__ tile_sums,
                                                                        size_t n) {
  __shared__ unsigned lds_wave[4];
  const size_t base = (size_t)blockIdx.x * SCAN_TILE;
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    size_t idx = base + (size_t)i * SCAN_THREADS + threadIdx.x;
    if (idx < n) s += in[idx];
  }
  s = (unsigned)wave_reduce_add((int)s);
  if ((threadIdx.x & 63) == 0) lds_wave[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = lds_wave[0] + lds_wave[1] + lds_wave[2] + lds_wave[3];
}

// Single workgroup: exclusive scan of `n` values (in place allowed), total out.
// 16 items per thread per trip (16 384 per trip) so that ~10^5 entries — the radix-sort digit
// histogram of a 2 M-key pass — take a handful of trips, not a hundred barrier-bound ones.
constexpr int SSB_ITEMS = 16;
__global__ __launch_bounds__(1024) void scan_single_block_kernel(const uint32_t* __restrict__ in,
                                                                 uint32_t* __restrict__ out, size_t n,
                                                                 uint32_t* __restrict__ total) {
  __shared__ unsigned lds_wave[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned carry = 0;
  for (size_t base = 0; base < n; base += (size_t)1024 * SSB_ITEMS) {
    const size_t first = base + (size_t)threadIdx.x * SSB_ITEMS;
    unsigned v[SSB_ITEMS];
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < SSB_ITEMS; ++i) {
      v[i] = (first + i < n) ? in[first + i] : 0u;
      s += v[i];
    }
    unsigned inc = wave_inclusive_scan(s);
    if (lane == 63) lds_wave[wave] = inc;
    __syncthreads();
    unsigned wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      unsigned t = lds_wave[w];
      if (w < wave) wbase += t;
      tot += t;
    }
    unsigned run = carry + wbase + inc - s;
#pragma unroll
    for (int i = 0; i < SSB_ITEMS; ++i) {
      if (first + i < n) out[first + i] = run;
      run += v[i];
    }
    carry += tot;
    __syncthreads();  // lds_wave is rewritten next trip
  }
  if (threadIdx.x == 0 && total) *total = carry;
}

// INLINE: tile_offsets holds the raw tile SUMS; this workgroup adds up the ones before it itself and the last workgroup
// writes the grand total — no scan launch between the reduce and apply passes.
template <bool INLINE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_apply_kernel(const uint32_t* __restrict__ in,
                                                                       uint32_t* __restrict__ out,
                                                                       const uint32_t* __restrict__ tile_offsets,
                                                                       size_t n, uint32_t* __restrict__ total) {
  __shared__ unsigned lds_wave[4];
  const unsigned tile_base = INLINE ? block_prefix_of_tiles(tile_offsets, blockIdx.x, lds_wave) : tile_offsets[blockIdx.x];
  // blocked arrangement: thread t owns items [t*ITEMS, t*ITEMS+ITEMS) of the tile
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  unsigned v[SCAN_ITEMS];
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0u;
    s += v[i];
  }
  unsigned tot;
  unsigned ex = block_exclusive_scan_256(s, lds_wave, &tot);
  unsigned run = tile_base + ex;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = run;
    run += v[i];
  }
  if (INLINE && total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = tile_base + tot;
}

constexpr size_t SCAN_SINGLE_BLOCK_MAX = 1u << 14;  // above this one workgroup's serial trips cost more than two extra launches

size_t scan_workspace_bytes(size_t n) {
  size_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  return align_up((ntiles + 1) * sizeof(uint32_t), 256);
}

int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total, void* ws,
                       size_t ws_bytes, hipStream_t stream) {
  if (n == 0) {
    if (total) return device_fill_u32(total, 1, 0u, stream);
    return BEVAMD_OK;
  }
  if (n <= SCAN_SINGLE_BLOCK_MAX) {
    scan_single_block_kernel<<<1, 1024, 0, stream>>>(in, out, n, total);
    BEVAMD_LAUNCH_CHECK("scan_single_block");
    return BEVAMD_OK;
  }
  size_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (ws_bytes < scan_workspace_bytes(n) || ws == nullptr) {
    set_error("exclusive_scan_u32: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  uint32_t* tile_sums = (uint32_t*)ws;
  scan_tile_reduce_kernel<<<(unsigned)ntiles, SCAN_THREADS, 0, stream>>>(in, tile_sums, n);
  BEVAMD_LAUNCH_CHECK("scan_tile_reduce");
  if (ntiles <= SCAN_INLINE_TILES_MAX) {
    scan_tile_apply_kernel<true><<<(unsigned)ntiles, SCAN_THREADS, 0, stream>>>(in, out, tile_sums, n, total);
  } else {
    scan_single_block_kernel<<<1, 1024, 0, stream>>>(tile_sums, tile_sums, ntiles, total);
    BEVAMD_LAUNCH_CHECK("scan_single_block");
    scan_tile_apply_kernel<false><<<(unsigned)ntiles, SCAN_THREADS, 0, stream>>>(in, out, tile_sums, n, nullptr);
  }
  BEVAMD_LAUNCH_CHECK("scan_tile_apply");
  return BEVAMD_OK;
}

bool single_pass_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("BEVAMD_SINGLE_PASS");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

// BEVAMD_SINGLE_PASS: "1" always, "0" never, unset / "auto": by size.  With every tile of a launch resident at once a tile
// walks back over all its predecessors, so the single-pass kernels lose against the multi-launch ones once there are many tiles:
// 8 sweeps (2.5 M points) 375 us against 320 us for the whole voxelizer; one sweep: 124 against 145 us and 9 launches
// instead of 22.
bool single_pass_for(size_t n) {
  static int mode = -1;   // 0 never, 1 always, 2 auto
  if (mode < 0) {
    const char* e = getenv("BEVAMD_SINGLE_PASS");
    mode = !e || e[0] == 'a' ? 2 : e[0] == '0' ? 0 : 1;
  }
  return mode == 1 || (mode == 2 && n <= SINGLE_PASS_AUTO_MAX);
}

size_t scan_lookback_state_bytes(size_t n) { return align_up(sp::scan_state_words(n) * 8, 256); }

int exclusive_scan_u32_lookback(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total, void* state, int* err,
                                hipStream_t stream) {
  if (n == 0) {
    if (total) return device_fill_u32(total, 1, 0u, stream);
    return BEVAMD_OK;
  }
  if (!state || ((uintptr_t)state & 7u)) {
    set_error("exclusive_scan_u32_lookback: state is null or not 8-byte aligned");
    return BEVAMD_ERR_INVALID_ARG;
  }
  sp::scan_lookback_launch(sp::LoadU32{in}, out, n, total, (unsigned long long*)state, err, stream);
  BEVAMD_LAUNCH_CHECK("scan_lookback");
  return BEVAMD_OK;
}

// Two independent scans of equal length in the launches of one (voxelization scans its segment-head flags and its
// first-point flags back to back): blockIdx.y / blockIdx.x selects the array.
struct ScanPair { const uint32_t* in[2]; uint32_t* out[2]; uint32_t* total[2]; };

__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_reduce_dual_kernel(ScanPair p, uint32_t* __restrict__ tile_sums,
                                                                             size_t ntiles, size_t n) {
  __shared__ unsigned lds_wave[4];
  const uint32_t* in = p.in[blockIdx.y];
  const size_t base = (size_t)blockIdx.x * SCAN_TILE;
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    size_t idx = base + (size_t)i * SCAN_THREADS + threadIdx.x;
    if (idx < n) s += in[idx];
  }
  s = (unsigned)wave_reduce_add((int)s);
  if ((threadIdx.x & 63) == 0) lds_wave[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) tile_sums[blockIdx.y * ntiles + blockIdx.x] = lds_wave[0] + lds_wave[1] + lds_wave[2] + lds_wave[3];
}

__global__ __launch_bounds__(1024) void scan_single_block_dual_kernel(uint32_t* __restrict__ tile_sums, size_t ntiles,
                                                                      ScanPair p) {
  __shared__ unsigned lds_wave[16];
  uint32_t* data = tile_sums + blockIdx.x * ntiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned carry = 0;
  for (size_t base = 0; base < ntiles; base += (size_t)1024 * SSB_ITEMS) {
    const size_t first = base + (size_t)threadIdx.x * SSB_ITEMS;
    unsigned v[SSB_ITEMS];
    unsigned s = 0;
#pragma unroll
    for (int i = 0; i < SSB_ITEMS; ++i) {
      v[i] = (first + i < ntiles) ? data[first + i] : 0u;
      s += v[i];
    }
    unsigned inc = wave_inclusive_scan(s);
    if (lane == 63) lds_wave[wave] = inc;
    __syncthreads();
    unsigned wbase = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      unsigned t = lds_wave[w];
      if (w < wave) wbase += t;
      tot += t;
    }
    unsigned run = carry + wbase + inc - s;
#pragma unroll
    for (int i = 0; i < SSB_ITEMS; ++i) {
      if (first + i < ntiles) data[first + i] = run;
      run += v[i];
    }
    carry += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0 && p.total[blockIdx.x]) *p.total[blockIdx.x] = carry;
}

template <bool INLINE>
__global__ __launch_bounds__(SCAN_THREADS) void scan_tile_apply_dual_kernel(ScanPair p, const uint32_t* __restrict__ tile_offsets,
                                                                            size_t ntiles, size_t n) {
  __shared__ unsigned lds_wave[4];
  const uint32_t* in = p.in[blockIdx.y];
  uint32_t* out = p.out[blockIdx.y];
  const uint32_t* my_tiles = tile_offsets + blockIdx.y * ntiles;
  const unsigned tile_base = INLINE ? block_prefix_of_tiles(my_tiles, blockIdx.x, lds_wave) : my_tiles[blockIdx.x];
  const size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
  unsigned v[SCAN_ITEMS];
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0u;
    s += v[i];
  }
  unsigned tot;
  unsigned ex = block_exclusive_scan_256(s, lds_wave, &tot);
  unsigned run = tile_base + ex;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = run;
    run += v[i];
  }
  if (INLINE && p.total[blockIdx.y] && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *p.total[blockIdx.y] = tile_base + tot;
}

size_t scan_dual_workspace_bytes(size_t n) {
  size_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  return align_up((2 * ntiles + 2) * sizeof(uint32_t), 256);
}

int exclusive_scan_u32_dual(const uint32_t* in_a, uint32_t* out_a, uint32_t* total_a, const uint32_t* in_b, uint32_t* out_b,
                            uint32_t* total_b, size_t n, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (n == 0) {
    if (total_a) { int rc = device_fill_u32(total_a, 1, 0u, stream); if (rc) return rc; }
    if (total_b) return device_fill_u32(total_b, 1, 0u, stream);
    return BEVAMD_OK;
  }
  if (ws_bytes < scan_dual_workspace_bytes(n) || ws == nullptr) {
    set_error("exclusive_scan_u32_dual: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  const size_t ntiles = (n + SCAN_TILE - 1) / SCAN_TILE;
  uint32_t* tile_sums = (uint32_t*)ws;
  ScanPair p{{in_a, in_b}, {out_a, out_b}, {total_a, total_b}};
  scan_tile_reduce_dual_kernel<<<dim3((unsigned)ntiles, 2), SCAN_THREADS, 0, stream>>>(p, tile_sums, ntiles, n);
  BEVAMD_LAUNCH_CHECK("scan_tile_reduce_dual");
  if (ntiles <= SCAN_INLINE_TILES_MAX) {
    scan_tile_apply_dual_kernel<true><<<dim3((unsigned)ntiles, 2), SCAN_THREADS, 0, stream>>>(p, tile_sums, ntiles, n);
  } else {
    scan_single_block_dual_kernel<<<2, 1024, 0, stream>>>(tile_sums, ntiles, p);
    BEVAMD_LAUNCH_CHECK("scan_single_block_dual");
    scan_tile_apply_dual_kernel<false><<<dim3((unsigned)ntiles, 2), SCAN_THREADS, 0, stream>>>(p, tile_sums, ntiles, n);
  }
  BEVAMD_LAUNCH_CHECK("scan_tile_apply_dual");
  return BEVAMD_OK;
}

// ============================================================================
// stable LSD radix sort, (u32 key, u32 value)
// ============================================================================
// Tile = 1024 pairs per 256-thread workgroup; wave w owns the contiguous chunk
// [w*256, (w+1)*256) of the tile and walks it in 4 rounds of 64 lanes, so the
// order (wave, round, lane) is the input order: ranking by "items before me with
// my digit" in that order is a stable partition.
constexpr int RS_THREADS = 256;
constexpr int RS_ROUNDS = 4;                      // 1024-key tiles: 300 workgroups for a 310 k-point LiDAR frame
constexpr int RS_WAVE_CHUNK = 64 * RS_ROUNDS;     // 256
constexpr int RS_TILE = 4 * RS_WAVE_CHUNK;        // 1024
constexpr int RS_MAX_BITS = 9;   // 27-bit voxel keys in 3 passes; 512 digit counters per wave (8 KiB of LDS)
constexpr int RS_MAX_BINS = 1 << RS_MAX_BITS;

// hist element (digit d, workgroup b) lives at d * sd + b * sb: digit-major (sd = nblocks, sb = 1) for the generic
// flat scan, block-major (sd = 1, sb = nbins) for the single-launch scan (coalesced for every kernel that touches it)
//
// Both kernels are written over one tile [base, end) whose digit counters sit at hcol[d * sd]; the plain sort maps
// workgroup -> tile by blockIdx.x, the segmented sort (several independent arrays, one launch) through a SortSegs table.
__device__ __forceinline__ void radix_hist_tile(const uint32_t* __restrict__ keys, size_t base, size_t end, int shift,
                                                int bits, unsigned sd, uint32_t* __restrict__ hcol) {
  __shared__ unsigned lh[RS_MAX_BINS];
  const unsigned nbins = 1u << bits, mask = nbins - 1u;
  for (unsigned i = threadIdx.x; i < nbins; i += RS_THREADS) lh[i] = 0;
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < RS_TILE / RS_THREADS; ++i) {
    size_t idx = base + (size_t)i * RS_THREADS + threadIdx.x;
    if (idx < end) atomicAdd(&lh[(keys[idx] >> shift) & mask], 1u);
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < nbins; i += RS_THREADS) hcol[(size_t)i * sd] = lh[i];
}

__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const uint32_t* __restrict__ keys, size_t n,
                                                                int shift, int bits, unsigned sd, unsigned sb,
                                                                uint32_t* __restrict__ hist) {
  radix_hist_tile(keys, (size_t)blockIdx.x * RS_TILE, n, shift, bits, sd, hist + (size_t)blockIdx.x * sb);
}

// Segment s owns elements [off[s], off[s+1]) and workgroups [blk[s], blk[s+1]); its histogram block starts at
// blk[s] << bits, digit-major over its own workgroups.  A flat exclusive scan of the concatenated histograms is then
// the destination of every (segment, digit, workgroup) run in the concatenated output: everything before segment s
// sums to off[s].
struct SegTile { size_t base, end; unsigned sd; size_t hcol; bool live; };
__device__ __forceinline__ SegTile seg_tile(const SortSegs& sg, int bits) {
  // workgroup -> tile: XCD x (= blockIdx.x % 8, workgroups go to the XCDs round-robin) takes the x-th contiguous eighth of the
  // tiles, i.e. about one segment of an 8-sweep batch: a segment's keys, values and digit buckets (2 x 2.4 MB) stay in one L2
  // (the grid is 8 * ceil(tiles / 8) workgroups; the ones past the last tile leave: SegTile::live)
  const unsigned tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  SegTile t;
  t.live = tile < sg.blk[sg.nseg];
  if (!t.live) return t;
  int s = 0;
  while (s + 1 < sg.nseg && tile >= sg.blk[s + 1]) ++s;
  const unsigned lb = tile - sg.blk[s];
  t.base = (size_t)sg.off[s] + (size_t)lb * RS_TILE;
  t.end = sg.off[s + 1];
  t.sd = sg.blk[s + 1] - sg.blk[s];
  t.hcol = ((size_t)sg.blk[s] << bits) + lb;
  return t;
}

__global__ __launch_bounds__(RS_THREADS) void radix_hist_seg_kernel(const uint32_t* __restrict__ keys, SortSegs sg,
                                                                    int shift, int bits, uint32_t* __restrict__ hist) {
  const SegTile t = seg_tile(sg, bits);
  if (!t.live) return;
  radix_hist_tile(keys, t.base, t.end, shift, bits, t.sd, hist + t.hcol);
}

__device__ __forceinline__ void radix_scatter_tile(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, size_t tile_base, size_t n, int shift, int bits,
    unsigned sd, const uint32_t* __restrict__ hcol) {
  __shared__ unsigned cnt[4][RS_MAX_BINS];  // per-wave running digit counters -> later: bases
  const unsigned nbins = 1u << bits, mask = nbins - 1u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (unsigned i = threadIdx.x; i < 4 * RS_MAX_BINS; i += RS_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();

  const size_t wbase = tile_base + (size_t)wave * RS_WAVE_CHUNK;
  uint32_t k[RS_ROUNDS], v[RS_ROUNDS];
  unsigned short rnk[RS_ROUNDS];
  const unsigned long long lt = lanemask_lt();
  volatile unsigned* wc = cnt[wave];

#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    size_t idx = wbase + (size_t)r * 64 + lane;
    bool valid = idx < n;
    k[r] = valid ? keys_in[idx] : 0xFFFFFFFFu;
    v[r] = valid ? vals_in[idx] : 0u;
  }
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    size_t idx = wbase + (size_t)r * 64 + lane;
    bool valid = idx < n;
    unsigned d = (k[r] >> shift) & mask;
    // match-any over the wave: lanes holding the same digit
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
      unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    unsigned before = __popcll(peers & lt);
    unsigned old = valid ? wc[d] : 0u;
    // leader (lowest peer lane) bumps the wave's counter for this digit
    if (valid && before == 0) wc[d] = old + (unsigned)__popcll(peers);
    rnk[r] = (unsigned short)(old + before);
  }
  __syncthreads();
  // per-digit exclusive prefix over the 4 waves + global base of (digit, block)
  for (unsigned d = threadIdx.x; d < nbins; d += RS_THREADS) {
    unsigned g = hcol[(size_t)d * sd];
    unsigned c0 = cnt[0][d], c1 = cnt[1][d], c2 = cnt[2][d];
    cnt[0][d] = g;
    cnt[1][d] = g + c0;
    cnt[2][d] = g + c0 + c1;
    cnt[3][d] = g + c0 + c1 + c2;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    size_t idx = wbase + (size_t)r * 64 + lane;
    if (idx < n) {
      unsigned d = (k[r] >> shift) & mask;
      unsigned dst = cnt[wave][d] + rnk[r];
      keys_out[dst] = k[r];
      vals_out[dst] = v[r];
    }
  }
}

__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, size_t n, int shift, int bits,
    unsigned sd, unsigned sb, const uint32_t* __restrict__ hist_scanned) {
  radix_scatter_tile(keys_in, vals_in, keys_out, vals_out, (size_t)blockIdx.x * RS_TILE, n, shift, bits, sd,
                     hist_scanned + (size_t)blockIdx.x * sb);
}

__global__ __launch_bounds__(RS_THREADS) void radix_scatter_seg_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, SortSegs sg, int shift, int bits,
    const uint32_t* __restrict__ hist_scanned) {
  const SegTile t = seg_tile(sg, bits);
  if (!t.live) return;
  radix_scatter_tile(keys_in, vals_in, keys_out, vals_out, t.base, t.end, shift, bits, t.sd, hist_scanned + t.hcol);
}

// ---- one-sweep passes of the segmented sort ------------------------------------------------------------------------------
// The four launches of a pass (tile histograms, their scan in two launches, a scatter that ranks the tile AGAIN) become
// one: the scatter kernel has its tile's digit counts the moment it has ranked the tile; what it lacks is the sum of the
// counts of the tiles BEFORE it in its segment — obtained by look-back over the words those tiles publish (single_pass.h) — and
// the digit totals of the whole segment, which depend only on the keys, not on their order, and are therefore counted for ALL
// passes in one read of the unsorted keys (radix_digit_totals_seg_kernel).  A 3-pass sort: 12 launches -> 4.
//
// Tile order: tiles are handed out by ticket.  With >= 8 segments the tiles are split into 8 LANES of whole segments
// (SortSegs::lane_blk), lane x = blockIdx.x % 8 = the XCD the workgroup runs on, each with its own ticket counter: a
// segment's keys, values and buckets stay in one L2 like with the three-launch passes, chains never cross a lane (a chain
// ends at its segment's first tile), and inside a lane ticket order = tile order.  Fewer segments: one lane, tile = ticket.
constexpr int OS_MAX_PASSES = 4;
constexpr int OS_TOTALS_TILES = 8;   // tiles per workgroup of the digit-totals kernel
constexpr unsigned OS_M31 = 0x7FFFFFFFu;
struct OneSweepPlan { int npass; int shift[OS_MAX_PASSES]; int bits[OS_MAX_PASSES]; };

__global__ __launch_bounds__(RS_THREADS) void radix_digit_totals_seg_kernel(const uint32_t* __restrict__ keys, SortSegs sg,
                                                                            OneSweepPlan pl, uint32_t* __restrict__ ghist) {
  __shared__ unsigned lh[OS_MAX_PASSES][RS_MAX_BINS];
  // workgroup -> (segment, group of OS_TOTALS_TILES tiles), XCD-contiguous like seg_tile
  const unsigned grp = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  int s = 0;
  unsigned first = 0;   // first group of segment s
  bool live = false;
  for (int q = 0; q < sg.nseg; ++q) {
    const unsigned ng = (sg.blk[q + 1] - sg.blk[q] + OS_TOTALS_TILES - 1) / OS_TOTALS_TILES;
    if (grp >= first && grp < first + ng) { s = q; live = true; break; }
    first += ng;
  }
  if (!live) return;
  for (unsigned i = threadIdx.x; i < (unsigned)pl.npass * RS_MAX_BINS; i += RS_THREADS) (&lh[0][0])[i] = 0;
  __syncthreads();
  const size_t base = (size_t)sg.off[s] + (size_t)(grp - first) * (OS_TOTALS_TILES * RS_TILE);
  const size_t end = sg.off[s + 1];
#pragma unroll 4
  for (int i = 0; i < OS_TOTALS_TILES * RS_TILE / RS_THREADS; ++i) {
    const size_t idx = base + (size_t)i * RS_THREADS + threadIdx.x;
    if (idx < end) {
      const uint32_t k = keys[idx];
      for (int p = 0; p < pl.npass; ++p) atomicAdd(&lh[p][(k >> pl.shift[p]) & ((1u << pl.bits[p]) - 1u)], 1u);
    }
  }
  __syncthreads();
  for (int p = 0; p < pl.npass; ++p)
    for (unsigned d = threadIdx.x; d < (1u << pl.bits[p]); d += RS_THREADS) {
      const unsigned c = lh[p][d];
      if (c) atomicAdd(&ghist[((size_t)p * sg.nseg + s) * RS_MAX_BINS + d], c);
    }
}

__global__ __launch_bounds__(RS_THREADS) void radix_onesweep_seg_kernel(
    const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, SortSegs sg, int shift, int bits, const uint32_t* __restrict__ totals /*[nseg][512]*/,
    unsigned long long* __restrict__ status /*[tiles][256]*/, unsigned* __restrict__ tickets /*[8]*/, int* err) {
  __shared__ unsigned cnt[4][RS_MAX_BINS];
  __shared__ unsigned lds_wave[4];
  __shared__ unsigned s_ticket;
  const unsigned lx = sg.lanes == 8 ? (blockIdx.x & 7u) : 0u;
  if (threadIdx.x == 0) s_ticket = atomicAdd(&tickets[lx], 1u);
  for (unsigned i = threadIdx.x; i < 4 * RS_MAX_BINS; i += RS_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const unsigned tile = sg.lane_blk[lx] + s_ticket;
  if (tile >= sg.lane_blk[lx + 1]) return;
  int s = 0;
  while (s + 1 < sg.nseg && tile >= sg.blk[s + 1]) ++s;
  const unsigned lb = tile - sg.blk[s];
  const size_t tile_base = (size_t)sg.off[s] + (size_t)lb * RS_TILE, n = sg.off[s + 1];

  const unsigned nbins = 1u << bits, mask = nbins - 1u;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wbase = tile_base + (size_t)wave * RS_WAVE_CHUNK;
  uint32_t k[RS_ROUNDS], v[RS_ROUNDS];
  unsigned short rnk[RS_ROUNDS];
  const unsigned long long lt = lanemask_lt();
  volatile unsigned* wc = cnt[wave];
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const size_t idx = wbase + (size_t)r * 64 + lane;
    const bool valid = idx < n;
    k[r] = valid ? keys_in[idx] : 0xFFFFFFFFu;
    v[r] = valid ? vals_in[idx] : 0u;
  }
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {   // rank inside the wave's chunk (see radix_scatter_tile)
    const size_t idx = wbase + (size_t)r * 64 + lane;
    const bool valid = idx < n;
    const unsigned d = (k[r] >> shift) & mask;
    unsigned long long peers = __ballot(valid);
    for (int b = 0; b < bits; ++b) {
      const unsigned long long bal = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? bal : ~bal;
    }
    const unsigned before = __popcll(peers & lt);
    const unsigned old = valid ? wc[d] : 0u;
    if (valid && before == 0) wc[d] = old + (unsigned)__popcll(peers);
    rnk[r] = (unsigned short)(old + before);
  }
  __syncthreads();
  // thread t owns digits 2t and 2t + 1: their counts in this tile, the digit totals of the segment (-> first bucket position
  // of each digit by a scan across the workgroup) and the counts of the tiles before this one (look-back)
  const unsigned d0 = 2u * threadIdx.x, d1 = d0 + 1u;
  const bool mine = d0 < nbins;
  unsigned c0 = 0, c1 = 0, w0c = 0, w1c = 0, w2c = 0, x0c = 0, x1c = 0, x2c = 0, g0 = 0, g1 = 0;
  if (mine) {
    w0c = cnt[0][d0]; w1c = cnt[1][d0]; w2c = cnt[2][d0];
    c0 = w0c + w1c + w2c + cnt[3][d0];
    x0c = cnt[0][d1]; x1c = cnt[1][d1]; x2c = cnt[2][d1];
    c1 = x0c + x1c + x2c + cnt[3][d1];
    g0 = totals[(size_t)s * RS_MAX_BINS + d0];
    g1 = totals[(size_t)s * RS_MAX_BINS + d1];
  }
  unsigned e0 = 0, e1 = 0;
  if (mine) {
    unsigned long long* my = status + (size_t)tile * (RS_MAX_BINS / 2) + threadIdx.x;
    if (lb == 0) {
      sp::store_word(my, sp::INC | ((unsigned long long)c1 << 31) | c0);
    } else {
      sp::store_word(my, sp::AGG | ((unsigned long long)c1 << 31) | c0);
      // walk back over the words of this digit pair, one tile per round trip (eight per round trip — the loads do not depend
      // on each other, only where the walk stops does — was slower: 64 / 71 / 58 us per pass against 59 / 61 / 52)
      const unsigned long long* q = my - (RS_MAX_BINS / 2);
      unsigned spins = 0;
      for (;;) {
        const unsigned long long w = sp::load_word(q);
        const unsigned flag = (unsigned)(w >> 62);
        if (flag == 0u) {
          if (++spins > sp::SPIN_LIMIT) {
            if (err) atomicOr(err, 1);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        e0 += (unsigned)w & OS_M31;
        e1 += (unsigned)(w >> 31) & OS_M31;
        if (flag == 2u) break;
        q -= RS_MAX_BINS / 2;      // an aggregate: the chain continues (it ends at the segment's first tile, which is inclusive)
        spins = 0;
      }
      sp::store_word(my, sp::INC | ((unsigned long long)(e1 + c1) << 31) | (unsigned long long)(e0 + c0));
    }
  }
  unsigned tot;
  const unsigned dbase = block_exclusive_scan_256(g0 + g1, lds_wave, &tot);   // (has the barriers the cnt rewrite needs)
  if (mine) {
    const unsigned b0 = sg.off[s] + dbase + e0, b1 = sg.off[s] + dbase + g0 + e1;
    cnt[0][d0] = b0; cnt[1][d0] = b0 + w0c; cnt[2][d0] = b0 + w0c + w1c; cnt[3][d0] = b0 + w0c + w1c + w2c;
    cnt[0][d1] = b1; cnt[1][d1] = b1 + x0c; cnt[2][d1] = b1 + x0c + x1c; cnt[3][d1] = b1 + x0c + x1c + x2c;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ROUNDS; ++r) {
    const size_t idx = wbase + (size_t)r * 64 + lane;
    if (idx < n) {
      const unsigned d = (k[r] >> shift) & mask;
      const unsigned dst = cnt[wave][d] + rnk[r];
      keys_out[dst] = k[r];
      vals_out[dst] = v[r];
    }
  }
}

size_t radix_sort_workspace_bytes(size_t n) {
  size_t nblocks = (n + RS_TILE - 1) / RS_TILE;
  if (nblocks == 0) nblocks = 1;
  size_t hist = align_up(nblocks * RS_MAX_BINS * sizeof(uint32_t), 256);
  return hist + scan_workspace_bytes(nblocks * RS_MAX_BINS);
}

// ---- small utility kernels (kernel nodes in HIP graphs; no hipMemset/hipMemcpy on the hot path) ------------
__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t* __restrict__ p, size_t n, uint32_t v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v;
}
__global__ __launch_bounds__(256) void copy_u32_kernel(uint32_t* __restrict__ d, const uint32_t* __restrict__ s, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}
static unsigned util_grid(size_t n) {
  size_t b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : b > 2048 ? 2048 : b);
}
int device_fill_u32(uint32_t* p, size_t n, uint32_t v, hipStream_t stream) {
  if (n == 0) return BEVAMD_OK;
  fill_u32_kernel<<<util_grid(n), 256, 0, stream>>>(p, n, v);
  BEVAMD_LAUNCH_CHECK("fill_u32");
  return BEVAMD_OK;
}
int device_copy_u32(uint32_t* d, const uint32_t* s, size_t n, hipStream_t stream) {
  if (n == 0 || d == s) return BEVAMD_OK;
  copy_u32_kernel<<<util_grid(n), 256, 0, stream>>>(d, s, n);
  BEVAMD_LAUNCH_CHECK("copy_u32");
  return BEVAMD_OK;
}

// The sorted pairs end in (*result_keys, *result_vals): the *_b buffers after an odd number of passes, the
// *_a buffers after an even number.  Both buffer pairs are clobbered.
int radix_sort_pairs_u32_ex(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t n, int nbits,
                            void* ws, size_t ws_bytes, hipStream_t stream, uint32_t** result_keys,
                            uint32_t** result_vals) {
  *result_keys = keys_a;
  *result_vals = vals_a;
  if (n == 0) return BEVAMD_OK;
  if (n >= (1ull << 32)) {
    set_error("radix_sort_pairs_u32: n too large");
    return BEVAMD_ERR_INVALID_ARG;
  }
  if (nbits < 1) nbits = 1;
  if (nbits > 32) nbits = 32;
  if (ws == nullptr || ws_bytes < radix_sort_workspace_bytes(n)) {
    set_error("radix_sort_pairs_u32: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  const unsigned nblocks = (unsigned)((n + RS_TILE - 1) / RS_TILE);
  uint32_t* hist = (uint32_t*)ws;
  size_t hist_bytes = align_up((size_t)nblocks * RS_MAX_BINS * sizeof(uint32_t), 256);
  void* scan_ws = (char*)ws + hist_bytes;
  size_t scan_ws_bytes = ws_bytes - hist_bytes;

  const int npass = (nbits + RS_MAX_BITS - 1) / RS_MAX_BITS;
  const int bits_per_pass = (nbits + npass - 1) / npass;
  uint32_t *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
  int shift = 0;
  for (int p = 0; p < npass; ++p) {
    int bits = bits_per_pass;
    if (shift + bits > nbits) bits = nbits - shift;
    if (bits <= 0) bits = 1;
    size_t hn = (size_t)nblocks << bits;
    // digit-major histogram + flat exclusive scan = global base of every (digit, workgroup).  (A single-workgroup
    // scan of the 300 x 512 matrix was measured at 40 us per pass against 14 us for the generic three-launch scan.)
    const unsigned sd = nblocks, sb = 1u;
    radix_hist_kernel<<<nblocks, RS_THREADS, 0, stream>>>(ki, n, shift, bits, sd, sb, hist);
    BEVAMD_LAUNCH_CHECK("radix_hist");
    int rc = exclusive_scan_u32(hist, hist, hn, nullptr, scan_ws, scan_ws_bytes, stream);
    if (rc != BEVAMD_OK) return rc;
    radix_scatter_kernel<<<nblocks, RS_THREADS, 0, stream>>>(ki, vi, ko, vo, n, shift, bits, sd, sb, hist);
    BEVAMD_LAUNCH_CHECK("radix_scatter");
    uint32_t* t;
    t = ki; ki = ko; ko = t;
    t = vi; vi = vo; vo = t;
    shift += bits;
  }
  *result_keys = ki;
  *result_vals = vi;
  return BEVAMD_OK;
}

int radix_sort_pairs_u32(uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                         size_t n, int nbits, void* ws, size_t ws_bytes, hipStream_t stream) {
  uint32_t *rk, *rv;
  int rc = radix_sort_pairs_u32_ex(keys_in, vals_in, keys_out, vals_out, n, nbits, ws, ws_bytes, stream, &rk, &rv);
  if (rc) return rc;
  if (rk != keys_out) {  // even number of passes: the result sits in the input buffers
    rc = device_copy_u32(keys_out, rk, n, stream);
    if (rc) return rc;
    rc = device_copy_u32(vals_out, rv, n, stream);
  }
  return rc;
}

// Segmented variant: sg.nseg independent arrays laid end to end (segment s = elements [off[s], off[s+1])), each sorted
// on its own — stable, on the low nbits — by the launches of one sort.  Values are whatever the caller stored (local
// indices, usually).  The result is in (*result_keys, *result_vals) like radix_sort_pairs_u32_ex.
int sort_segs_init(SortSegs& sg, const int* counts, int nseg) {
  if (nseg < 1 || nseg > SORT_MAX_SEGS) {
    set_error("segmented sort: %d segments (1..%d supported)", nseg, SORT_MAX_SEGS);
    return BEVAMD_ERR_INVALID_ARG;
  }
  sg.nseg = nseg;
  sg.off[0] = 0;
  sg.blk[0] = 0;
  unsigned long long tot = 0;
  for (int s = 0; s < nseg; ++s) {
    if (counts[s] < 0) {
      set_error("segmented sort: negative count");
      return BEVAMD_ERR_INVALID_ARG;
    }
    tot += (unsigned long long)counts[s];
    if (tot >= (1ull << 32)) {
      set_error("segmented sort: more than 2^32 elements");
      return BEVAMD_ERR_INVALID_ARG;
    }
    sg.off[s + 1] = (uint32_t)tot;
    sg.blk[s + 1] = sg.blk[s] + (uint32_t)(((size_t)counts[s] + RS_TILE - 1) / RS_TILE);
  }
  // lanes of the one-sweep passes: 8 contiguous groups of whole segments, balanced by tiles (each boundary at the segment end
  // nearest to its share); fewer than 8 segments: one lane
  const uint32_t nt = sg.blk[nseg];
  sg.lanes = nseg >= 8 ? 8 : 1;
  for (int x = 0; x <= 8; ++x) sg.lane_blk[x] = nt;
  sg.lane_blk[0] = 0;
  if (sg.lanes == 8) {
    int q = 0;
    for (int x = 1; x < 8; ++x) {
      const uint32_t want = (uint32_t)(((unsigned long long)nt * x) / 8);
      while (q < nseg && sg.blk[q + 1] <= want) ++q;
      // boundary at blk[q] or blk[q + 1], whichever is nearer to `want`
      uint32_t cut = sg.blk[q];
      if (q < nseg && sg.blk[q + 1] - want < want - sg.blk[q]) cut = sg.blk[q + 1];
      if (cut < sg.lane_blk[x - 1]) cut = sg.lane_blk[x - 1];
      sg.lane_blk[x] = cut;
    }
  }
  return BEVAMD_OK;
}

static bool onesweep_fits(const SortSegs& sg) {   // 31-bit counts in the look-back words; BEVAMD_SORT_ONESWEEP=0 / the size rule
  static int env = -1;
  if (env < 0) {
    const char* e = getenv("BEVAMD_SORT_ONESWEEP");
    env = (e && e[0] == '0') ? 0 : 1;
  }
  for (int s = 0; s < sg.nseg; ++s)
    if (sg.off[s + 1] - sg.off[s] >= (1u << 31)) return false;
  return env == 1 && single_pass_for(sg.off[sg.nseg]);
}
static size_t legacy_sort_bytes(const SortSegs& sg) {
  size_t nblocks = sg.blk[sg.nseg] ? sg.blk[sg.nseg] : 1;
  size_t hist = align_up(nblocks * RS_MAX_BINS * sizeof(uint32_t), 256);
  return hist + scan_workspace_bytes(nblocks * RS_MAX_BINS);
}
// one-sweep state behind the three-launch workspace: [tickets u32[OS_MAX_PASSES][8], err at u32[32]: 256 B]
// [digit totals u32[OS_MAX_PASSES][nseg][512]] [look-back words u64[passes][tiles][256]]
static size_t onesweep_totals_bytes(const SortSegs& sg) { return align_up((size_t)OS_MAX_PASSES * sg.nseg * RS_MAX_BINS * 4, 256); }
static size_t onesweep_status_bytes(const SortSegs& sg) { return (size_t)(sg.blk[sg.nseg] ? sg.blk[sg.nseg] : 1) * (RS_MAX_BINS / 2) * 8; }
static int sort_passes(int nbits, int* bits_per_pass) {
  if (nbits < 1) nbits = 1;
  if (nbits > 32) nbits = 32;
  const int npass = (nbits + RS_MAX_BITS - 1) / RS_MAX_BITS;
  *bits_per_pass = (nbits + npass - 1) / npass;
  return npass;
}

void radix_sort_segmented_state(const SortSegs& sg, int nbits, void* ws, unsigned long long** state, size_t* words) {
  *state = nullptr;
  *words = 0;
  if (!ws || !onesweep_fits(sg) || sg.blk[sg.nseg] == 0) return;
  int bpp;
  const int npass = sort_passes(nbits, &bpp);
  *state = (unsigned long long*)((char*)ws + legacy_sort_bytes(sg));
  *words = (256 + onesweep_totals_bytes(sg) + (size_t)npass * onesweep_status_bytes(sg)) / 8;
}

size_t radix_sort_segmented_workspace_bytes(const SortSegs& sg) {
  return legacy_sort_bytes(sg) + 256 + onesweep_totals_bytes(sg) + (size_t)OS_MAX_PASSES * onesweep_status_bytes(sg);
}

static int radix_sort_segmented_onesweep(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b,
                                         const SortSegs& sg, int nbits, void* ws, hipStream_t stream,
                                         uint32_t** result_keys, uint32_t** result_vals, bool state_zeroed) {
  unsigned long long* state;
  size_t words;
  radix_sort_segmented_state(sg, nbits, ws, &state, &words);
  if (!state_zeroed) {
    int rc = device_fill_u32((uint32_t*)state, words * 2, 0u, stream);
    if (rc) return rc;
  }
  unsigned* tickets = (unsigned*)state;
  int* err = (int*)state + 32;
  uint32_t* totals = (uint32_t*)((char*)state + 256);
  unsigned long long* status = (unsigned long long*)((char*)totals + onesweep_totals_bytes(sg));
  OneSweepPlan pl;
  int bpp;
  pl.npass = sort_passes(nbits, &bpp);
  if (nbits < 1) nbits = 1;
  if (nbits > 32) nbits = 32;
  int shift = 0;
  for (int p = 0; p < pl.npass; ++p) {
    int bits = bpp;
    if (shift + bits > nbits) bits = nbits - shift;
    if (bits <= 0) bits = 1;
    pl.shift[p] = shift;
    pl.bits[p] = bits;
    shift += bits;
  }
  unsigned groups = 0;
  for (int s = 0; s < sg.nseg; ++s) groups += (sg.blk[s + 1] - sg.blk[s] + OS_TOTALS_TILES - 1) / OS_TOTALS_TILES;
  radix_digit_totals_seg_kernel<<<(groups + 7) / 8 * 8, RS_THREADS, 0, stream>>>(keys_a, sg, pl, totals);
  BEVAMD_LAUNCH_CHECK("radix_digit_totals_seg");
  unsigned grid = sg.blk[sg.nseg];
  if (sg.lanes == 8) {
    unsigned most = 0;
    for (int x = 0; x < 8; ++x) most = sg.lane_blk[x + 1] - sg.lane_blk[x] > most ? sg.lane_blk[x + 1] - sg.lane_blk[x] : most;
    grid = most * 8;
  }
  uint32_t *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
  for (int p = 0; p < pl.npass; ++p) {
    radix_onesweep_seg_kernel<<<grid, RS_THREADS, 0, stream>>>(
        ki, vi, ko, vo, sg, pl.shift[p], pl.bits[p], totals + (size_t)p * sg.nseg * RS_MAX_BINS,
        (unsigned long long*)((char*)status + (size_t)p * onesweep_status_bytes(sg)), tickets + 8 * p, err);
    BEVAMD_LAUNCH_CHECK("radix_onesweep_seg");
    uint32_t* t;
    t = ki; ki = ko; ko = t;
    t = vi; vi = vo; vo = t;
  }
  *result_keys = ki;
  *result_vals = vi;
  return BEVAMD_OK;
}

int radix_sort_pairs_u32_segmented(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b,
                                   const SortSegs& sg, int nbits, void* ws, size_t ws_bytes, hipStream_t stream,
                                   uint32_t** result_keys, uint32_t** result_vals, bool state_zeroed) {
  *result_keys = keys_a;
  *result_vals = vals_a;
  const unsigned nblocks = sg.blk[sg.nseg];
  if (nblocks == 0) return BEVAMD_OK;
  if (ws == nullptr || ws_bytes < radix_sort_segmented_workspace_bytes(sg)) {
    set_error("radix_sort_pairs_u32_segmented: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  if (onesweep_fits(sg))
    return radix_sort_segmented_onesweep(keys_a, vals_a, keys_b, vals_b, sg, nbits, ws, stream, result_keys, result_vals,
                                         state_zeroed);
  if (nbits < 1) nbits = 1;
  if (nbits > 32) nbits = 32;
  uint32_t* hist = (uint32_t*)ws;
  size_t hist_bytes = align_up((size_t)nblocks * RS_MAX_BINS * sizeof(uint32_t), 256);
  void* scan_ws = (char*)ws + hist_bytes;
  size_t scan_ws_bytes = ws_bytes - hist_bytes;
  const int npass = (nbits + RS_MAX_BITS - 1) / RS_MAX_BITS;
  const int bits_per_pass = (nbits + npass - 1) / npass;
  uint32_t *ki = keys_a, *vi = vals_a, *ko = keys_b, *vo = vals_b;
  int shift = 0;
  for (int p = 0; p < npass; ++p) {
    int bits = bits_per_pass;
    if (shift + bits > nbits) bits = nbits - shift;
    if (bits <= 0) bits = 1;
    radix_hist_seg_kernel<<<(nblocks + 7) / 8 * 8, RS_THREADS, 0, stream>>>(ki, sg, shift, bits, hist);
    BEVAMD_LAUNCH_CHECK("radix_hist_seg");
    int rc = exclusive_scan_u32(hist, hist, (size_t)nblocks << bits, nullptr, scan_ws, scan_ws_bytes, stream);
    if (rc != BEVAMD_OK) return rc;
    radix_scatter_seg_kernel<<<(nblocks + 7) / 8 * 8, RS_THREADS, 0, stream>>>(ki, vi, ko, vo, sg, shift, bits, hist);
    BEVAMD_LAUNCH_CHECK("radix_scatter_seg");
    uint32_t* t;
    t = ki; ki = ko; ko = t;
    t = vi; vi = vo; vo = t;
    shift += bits;
  }
  *result_keys = ki;
  *result_vals = vi;
  return BEVAMD_OK;
}

}  // namespace bevamd

// ---- C-ABI test hooks for the primitives (used by tests/ only) --------------
namespace bevamd {
// test hook: every CU's LDS filled with `pattern` (LDS is not cleared between kernels: a kernel that reads a word it never wrote sees
// what the previous tenant left) — tests poison it with NaN bits in front of kernels whose partial staging they want to prove harmless
__global__ __launch_bounds__(1024) void lds_poison_kernel(uint32_t pattern, uint32_t* sink) {
  extern __shared__ uint32_t lds_words[];
  const int words = 160 * 1024 / 4;
  for (int i = threadIdx.x; i < words; i += 1024) lds_words[i] = pattern;
  __syncthreads();
  if (lds_words[(threadIdx.x * 37) % words] != pattern && sink) sink[0] = 1;   // keeps the stores alive
}
}  // namespace bevamd

extern "C" {
const char* bevamd_last_error(void) { return bevamd::get_error(); }

size_t bevamd_scan_workspace_bytes(size_t n) { return bevamd::scan_workspace_bytes(n); }
int bevamd_exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total, void* ws,
                              size_t ws_bytes, void* stream) {
  return bevamd::exclusive_scan_u32(in, out, n, total, ws, ws_bytes, (hipStream_t)stream);
}
size_t bevamd_scan_single_pass_state_bytes(size_t n) { return bevamd::scan_lookback_state_bytes(n); }
int bevamd_exclusive_scan_u32_single_pass(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total, void* state,
                                          size_t state_bytes, void* stream) {
  if (n && (!state || state_bytes < bevamd::scan_lookback_state_bytes(n))) {
    bevamd::set_error("exclusive_scan_u32_single_pass: state too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  // (a caller inside the library has its previous kernel zero the state; this entry spends a launch on it)
  if (n) {
    int rc = bevamd::device_fill_u32((uint32_t*)state, bevamd::scan_lookback_state_bytes(n) / 4, 0u, (hipStream_t)stream);
    if (rc) return rc;
  }
  return bevamd::exclusive_scan_u32_lookback(in, out, n, total, state, nullptr, (hipStream_t)stream);
}
size_t bevamd_radix_sort_workspace_bytes(size_t n) { return bevamd::radix_sort_workspace_bytes(n); }
int bevamd_radix_sort_pairs_u32(uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out,
                                uint32_t* vals_out, size_t n, int nbits, void* ws, size_t ws_bytes,
                                void* stream) {
  return bevamd::radix_sort_pairs_u32(keys_in, vals_in, keys_out, vals_out, n, nbits, ws, ws_bytes,
                                      (hipStream_t)stream);
}
/* host-only: how the one-sweep passes deal the tiles of `nseg` segments to their lanes — tile_begin[nseg + 1] (first tile of every
 * segment), lane_begin[9] (first tile of every lane; lanes own whole segments), returns the lane count (8 from 8 segments on,
 * else 1) or a negative error code.  For tests of the partition (no GPU needed). */
int bevamd_radix_sort_segmented_lanes(const int* counts, int nseg, unsigned* tile_begin, unsigned* lane_begin) {
  bevamd::SortSegs sg;
  if (!counts || !tile_begin || !lane_begin) {
    bevamd::set_error("radix_sort_segmented_lanes: null argument");
    return -BEVAMD_ERR_INVALID_ARG;
  }
  int rc = bevamd::sort_segs_init(sg, counts, nseg);
  if (rc) return -rc;
  for (int s = 0; s <= nseg; ++s) tile_begin[s] = sg.blk[s];
  for (int x = 0; x <= 8; ++x) lane_begin[x] = sg.lane_blk[x];
  return sg.lanes;
}
size_t bevamd_radix_sort_segmented_workspace_bytes(const int* counts, int nseg) {
  bevamd::SortSegs sg;
  if (!counts || bevamd::sort_segs_init(sg, counts, nseg) != BEVAMD_OK) return 0;
  return bevamd::radix_sort_segmented_workspace_bytes(sg);
}
int bevamd_radix_sort_pairs_u32_segmented(uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                                          const int* counts, int nseg, int nbits, void* ws, size_t ws_bytes, void* stream) {
  bevamd::SortSegs sg;
  if (!counts) {
    bevamd::set_error("radix_sort_pairs_u32_segmented: counts is null (host array)");
    return BEVAMD_ERR_INVALID_ARG;
  }
  int rc = bevamd::sort_segs_init(sg, counts, nseg);
  if (rc) return rc;
  uint32_t *rk, *rv;
  rc = bevamd::radix_sort_pairs_u32_segmented(keys_in, vals_in, keys_out, vals_out, sg, nbits, ws, ws_bytes,
                                              (hipStream_t)stream, &rk, &rv);
  if (rc) return rc;
  if (rk != keys_out) {
    rc = bevamd::device_copy_u32(keys_out, rk, sg.off[nseg], (hipStream_t)stream);
    if (rc) return rc;
    rc = bevamd::device_copy_u32(vals_out, rv, sg.off[nseg], (hipStream_t)stream);
  }
  return rc;
}
/* test hook: fill the LDS of every CU with `pattern` (2048 workgroups x 160 KiB). */
int bevamd_debug_lds_poison(uint32_t pattern, void* stream) {
  static bool attr = false;
  if (!attr) {
    BEVAMD_HIP_CHECK(hipFuncSetAttribute((const void*)bevamd::lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  hipLaunchKernelGGL(bevamd::lds_poison_kernel, dim3(2048), dim3(1024), 160 * 1024, (hipStream_t)stream, pattern, (uint32_t*)nullptr);
  BEVAMD_LAUNCH_CHECK("lds_poison_kernel");
  return BEVAMD_OK;
}
}
