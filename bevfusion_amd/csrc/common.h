// Shared device/host helpers for the gfx950 kernels of the BEVFusion hot path.
// Everything here is wave64 / CDNA4 specific on purpose: no warp-size
// abstraction, no portability shims.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define BEVAMD_WAVE 64

// ---- status codes of the C-ABI (mirrored in include/bevfusion_amd.h) --------
#define BEVAMD_OK 0
#define BEVAMD_ERR_INVALID_ARG 1
#define BEVAMD_ERR_WORKSPACE 2
#define BEVAMD_ERR_HIP 3
#define BEVAMD_ERR_UNSUPPORTED 4

namespace bevamd {

// last-error string, thread-local; read through bevamd_last_error().
void set_error(const char* fmt, ...);
const char* get_error();

static inline int hip_fail(hipError_t e, const char* what) {
  set_error("%s: %s", what, hipGetErrorString(e));
  return BEVAMD_ERR_HIP;
}

#define BEVAMD_HIP_CHECK(expr)                                  \
  do {                                                          \
    hipError_t _e = (expr);                                     \
    if (_e != hipSuccess) return ::bevamd::hip_fail(_e, #expr); \
  } while (0)

#define BEVAMD_LAUNCH_CHECK(name)                                   \
  do {                                                              \
    hipError_t _e = hipGetLastError();                              \
    if (_e != hipSuccess) return ::bevamd::hip_fail(_e, "launch " name); \
  } while (0)

#define BEVAMD_REQUIRE(cond, ...)          \
  do {                                     \
    if (!(cond)) {                         \
      ::bevamd::set_error(__VA_ARGS__);    \
      return BEVAMD_ERR_INVALID_ARG;       \
    }                                      \
  } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Carves 256-byte aligned sub-buffers out of a caller-provided workspace.
struct Carver {
  char* base;
  size_t off;
  size_t cap;
  bool ok;
  Carver(void* p, size_t bytes) : base((char*)p), off(0), cap(bytes), ok(true) {}
  template <typename T>
  T* take(size_t count) {
    size_t b = align_up(count * sizeof(T), 256);
    T* r = (T*)(base + off);
    off += b;
    if (off > cap) ok = false;
    return base ? r : nullptr;
  }
};

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- device-side wave helpers ----------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ unsigned long long lanemask_lt() {
  return (1ull << (threadIdx.x & 63)) - 1ull;
}

__device__ __forceinline__ int wave_reduce_add(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ unsigned wave_inclusive_scan(unsigned v) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned t = __shfl_up(v, o, 64);
    if ((threadIdx.x & 63) >= o) v += t;
  }
  return v;
}

// ---- primitives shared by the three ops (scan.hip / radix_sort.hip) ---------
// Exclusive prefix sum of n uint32 values; `total` (device, optional) receives
// the grand total.  in == out allowed.  ws must hold scan_workspace_bytes(n).
size_t scan_workspace_bytes(size_t n);
int exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total,
                       void* ws, size_t ws_bytes, hipStream_t stream);

// Two exclusive scans of equal length n sharing their launches (in == out allowed per array).
size_t scan_dual_workspace_bytes(size_t n);
int exclusive_scan_u32_dual(const uint32_t* in_a, uint32_t* out_a, uint32_t* total_a, const uint32_t* in_b,
                            uint32_t* out_b, uint32_t* total_b, size_t n, void* ws, size_t ws_bytes,
                            hipStream_t stream);

// Stable LSD radix sort of (key,value) pairs on the low `nbits` bits of the key.
// Results land in keys_out/vals_out.  keys_in/vals_in are clobbered (used as the
// ping-pong partner).  ws must hold radix_sort_workspace_bytes(n).
size_t radix_sort_workspace_bytes(size_t n);
int radix_sort_pairs_u32(uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out,
                         uint32_t* vals_out, size_t n, int nbits, void* ws, size_t ws_bytes,
                         hipStream_t stream);

// Same sort without the final placement guarantee: the result lands in (*result_keys, *result_vals), which is
// one of the two buffer pairs (no copy for an even number of passes).
int radix_sort_pairs_u32_ex(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, size_t n,
                            int nbits, void* ws, size_t ws_bytes, hipStream_t stream, uint32_t** result_keys,
                            uint32_t** result_vals);
// Segmented sort: up to SORT_MAX_SEGS independent arrays laid end to end, each stably sorted on its own by the launches of
// one sort (a batch of LiDAR sweeps).  The table travels as a kernel argument.
constexpr int SORT_MAX_SEGS = 64;
struct SortSegs {
  int nseg;
  uint32_t off[SORT_MAX_SEGS + 1];  // element range of segment s: [off[s], off[s+1])
  uint32_t blk[SORT_MAX_SEGS + 1];  // workgroup range of segment s
  // one-sweep passes: lane x (= blockIdx.x % lanes) owns the tiles [lane_blk[x], lane_blk[x+1]) — whole segments — and hands
  // them out in ticket order (see radix_onesweep_seg_kernel); lanes = 8 when there are at least 8 segments, else 1
  int lanes;
  uint32_t lane_blk[9];
};
int sort_segs_init(SortSegs& sg, const int* counts, int nseg);
size_t radix_sort_segmented_workspace_bytes(const SortSegs& sg);
// state_zeroed: the caller's own first kernel has zeroed the words radix_sort_segmented_state() names (the one-sweep passes'
// digit totals, look-back words and tickets); otherwise the sort zeroes them with a launch of its own.
int radix_sort_pairs_u32_segmented(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b,
                                   const SortSegs& sg, int nbits, void* ws, size_t ws_bytes, hipStream_t stream,
                                   uint32_t** result_keys, uint32_t** result_vals, bool state_zeroed = false);
// the part of a segmented sort's workspace that must be zero when the sort starts (8-byte words); *words = 0: none (the
// three-launch passes are in use: BEVAMD_SORT_ONESWEEP=0)
void radix_sort_segmented_state(const SortSegs& sg, int nbits, void* ws, unsigned long long** state, size_t* words);

// Single-pass exclusive scan (chained tiles with decoupled look-back): ONE launch.  `state` — scan_lookback_state_bytes(n)
// bytes, 8-byte aligned — must be ZERO when the kernel starts (zeroed by an earlier kernel of the same stream: no launch for it)
// and is left dirty.  in == out allowed.
size_t scan_lookback_state_bytes(size_t n);
int exclusive_scan_u32_lookback(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total, void* state, int* err,
                                hipStream_t stream);
bool single_pass_enabled();   // BEVAMD_SINGLE_PASS != "0"
constexpr size_t SINGLE_PASS_AUTO_MAX = 1000000;   // elements up to which the single-pass kernels are the default
bool single_pass_for(size_t n);
// fill / copy of 32-bit words as ordinary kernels (captured as kernel nodes in HIP graphs)
int device_fill_u32(uint32_t* p, size_t n, uint32_t v, hipStream_t stream);
int device_copy_u32(uint32_t* d, const uint32_t* s, size_t n, hipStream_t stream);

static inline int bits_for(uint64_t max_value_exclusive) {
  int b = 1;
  while (b < 32 && (1ull << b) < max_value_exclusive) ++b;
  return b;
}

}  // namespace bevamd
