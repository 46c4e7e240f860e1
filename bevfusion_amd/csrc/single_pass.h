// Single-pass (one launch) building blocks: tiles chained by decoupled look-back (Merrill & Garland's scheme).
//
// A workgroup takes a TICKET (atomic counter) and works on the tile of that number: every tile with a lower number has then
// been handed to a workgroup that is already running, so waiting for a predecessor can not deadlock however the hardware
// orders its dispatches.  A tile publishes its aggregate in one 8-byte word (flag | value) as soon as it has it, walks back
// over its predecessors' words — adding aggregates — until it meets an inclusive prefix, then publishes its own inclusive
// prefix.  Words are written / polled with relaxed agent-scope atomics (the word carries the data itself: nothing else has
// to become visible with it; the XCDs' L2s are not coherent for plain accesses).
//
// The state words must be ZERO when a kernel starts.  No kernel of this file zeroes its own state (other workgroups may still
// be polling it): the caller's preceding kernel does, as a few extra stores — no launch is spent on it.  Spins are bounded: a
// word that stays invalid for ~seconds sets bit 0 of *err and the workgroup continues with what it has (wrong results, flagged;
// never a hung GPU).
#pragma once
#include "common.h"

namespace bevamd {
namespace sp {

constexpr unsigned SPIN_LIMIT = 1u << 21;
constexpr unsigned long long AGG = 1ull << 62, INC = 2ull << 62;

__device__ __forceinline__ unsigned long long load_word(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_word(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Exclusive prefix of tile `tile` from the tiles before it (32-bit sums), by the FIRST WAVE of the workgroup, 64 predecessors
// per poll; publishes the tile's inclusive prefix (prefix + tot).  Call with threadIdx.x < 64; every lane returns the prefix.
// status[t]: word of tile t.
__device__ __forceinline__ unsigned lookback_prefix_u32(unsigned long long* status, unsigned tile, unsigned tot, int* err) {
  const int lane = threadIdx.x & 63;
  if (tile == 0) {
    if (lane == 0) store_word(&status[0], INC | tot);
    return 0u;
  }
  if (lane == 0) store_word(&status[tile], AGG | tot);
  long long look = (long long)tile - 1 - lane;
  unsigned acc = 0, spins = 0;
  for (;;) {
    const unsigned long long w = look >= 0 ? load_word(&status[look]) : INC;   // before tile 0: an inclusive prefix of zero
    const unsigned flag = (unsigned)(w >> 62);
    const unsigned long long inc = __ballot(flag == 2u), inv = __ballot(flag == 0u);
    if (inc) {
      const int p = __ffsll((long long)inc) - 1;                      // nearest tile that knows its inclusive prefix
      const unsigned long long upto = p == 63 ? ~0ull : ((2ull << p) - 1ull);
      if ((inv & upto) == 0ull) {                                      // ... and every tile between it and us has an aggregate
        acc += (unsigned)wave_reduce_add(lane <= p ? (int)(unsigned)w : 0);
        break;
      }
    } else if (!inv) {                                                 // 64 aggregates: take them all, look further back
      acc += (unsigned)wave_reduce_add((int)(unsigned)w);
      look -= 64;
      spins = 0;
      continue;
    }
    if (++spins > SPIN_LIMIT) {
      if (lane == 0 && err) atomicOr(err, 1);
      break;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  if (lane == 0) store_word(&status[tile], INC | (unsigned long long)(acc + tot));
  return acc;
}

// 8192 elements per tile: all tiles of a launch are resident at once, so nobody finds an inclusive prefix early and a tile
// walks back over (nearly) all its predecessors, 64 per poll — 2048-element tiles: 1 220 tiles = 19 polls = 32 us for the
// 2.5 M rows of 8 LiDAR sweeps; 8192: 5 polls.  Lane l of wave w owns, in each of the 8 rows of its wave's 2048-element
// slice, 4 consecutive elements (16 bytes: loads and stores of a wave instruction cover 1 KB contiguously).
// Small inputs take 2048-element tiles (ROWS = 2): 8192 would leave a 310 k-row scan 38 workgroups.
constexpr int SCAN_THREADS = 256, SCAN_VEC = 4;
constexpr size_t SCAN_SMALL_MAX = (size_t)1 << 20;
static inline int scan_rows_for(size_t n) { return n <= SCAN_SMALL_MAX ? 2 : 8; }
static inline size_t scan_tile_for(size_t n) { return (size_t)SCAN_THREADS * SCAN_VEC * scan_rows_for(n); }

struct LoadU32 {
  const uint32_t* p;
  __device__ __forceinline__ uint32_t operator()(size_t i) const { return p[i]; }
};

// state: word 0 = ticket counter (low 32 bits), words 1.. = one status word per tile
static inline size_t scan_tiles(size_t n) { return (n + scan_tile_for(n) - 1) / scan_tile_for(n); }
static inline size_t scan_state_words(size_t n) { return 1 + scan_tiles(n); }

// out[i] = sum of load(j) for j < i; *total (optional) = the grand total.  `load` may compute its values on the fly (a producer
// kernel folded into the scan); it is called once per element.  Grid = tiles.
template <class Load, int SCAN_ROWS>
__global__ __launch_bounds__(SCAN_THREADS) void scan_lookback_kernel(Load load, uint32_t* __restrict__ out, size_t n,
                                                                     uint32_t* __restrict__ total,
                                                                     unsigned long long* __restrict__ state, int* err) {
  __shared__ unsigned lds_wave[4];
  __shared__ unsigned s_tile, s_prefix;
  if (threadIdx.x == 0) s_tile = atomicAdd((unsigned*)state, 1u);
  __syncthreads();
  const unsigned tile = s_tile;
  constexpr int SCAN_ITEMS = SCAN_ROWS * SCAN_VEC, SCAN_TILE = SCAN_THREADS * SCAN_ITEMS, SCAN_WAVE_SLICE = 64 * SCAN_ITEMS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t wbase = (size_t)tile * SCAN_TILE + (size_t)wave * SCAN_WAVE_SLICE + (size_t)lane * SCAN_VEC;
  unsigned v[SCAN_ROWS][SCAN_VEC], ex[SCAN_ROWS];
  unsigned run = 0;   // sum of the rows of this wave so far (every lane)
#pragma unroll
  for (int r = 0; r < SCAN_ROWS; ++r)
#pragma unroll
    for (int c = 0; c < SCAN_VEC; ++c) {
      const size_t i = wbase + (size_t)r * (64 * SCAN_VEC) + c;
      v[r][c] = i < n ? load(i) : 0u;
    }
#pragma unroll
  for (int r = 0; r < SCAN_ROWS; ++r) {
    const unsigned s = v[r][0] + v[r][1] + v[r][2] + v[r][3];
    const unsigned inc = wave_inclusive_scan(s);
    ex[r] = run + inc - s;                       // exclusive prefix of this lane's 4 elements inside the wave's slice
    run += __shfl(inc, 63, 64);
  }
  if (lane == 0) lds_wave[wave] = run;
  __syncthreads();
  const unsigned w0 = lds_wave[0], w1 = lds_wave[1], w2 = lds_wave[2], w3 = lds_wave[3];
  const unsigned wex = (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u);
  const unsigned tot = w0 + w1 + w2 + w3;
  if (threadIdx.x < 64) {
    const unsigned pre = lookback_prefix_u32(state + 1, tile, tot, err);
    if (threadIdx.x == 0) s_prefix = pre;
  }
  __syncthreads();
  const unsigned base = s_prefix + wex;
#pragma unroll
  for (int r = 0; r < SCAN_ROWS; ++r) {
    unsigned a = base + ex[r];
#pragma unroll
    for (int c = 0; c < SCAN_VEC; ++c) {
      const size_t i = wbase + (size_t)r * (64 * SCAN_VEC) + c;
      if (i < n) out[i] = a;
      a += v[r][c];
    }
  }
  if (total && tile == gridDim.x - 1 && threadIdx.x == 0) *total = s_prefix + tot;
}

// launch: grid = scan_tiles(n)
template <class Load>
static inline void scan_lookback_launch(const Load& load, uint32_t* out, size_t n, uint32_t* total, unsigned long long* state,
                                        int* err, hipStream_t stream) {
  const unsigned tiles = (unsigned)scan_tiles(n);
  if (scan_rows_for(n) == 2) scan_lookback_kernel<Load, 2><<<tiles, SCAN_THREADS, 0, stream>>>(load, out, n, total, state, err);
  else scan_lookback_kernel<Load, 8><<<tiles, SCAN_THREADS, 0, stream>>>(load, out, n, total, state, err);
}

}  // namespace sp
}  // namespace bevamd
