// bev_pool for gfx950: interval reduction of camera-frustum features into the
// BEV grid, plus the rank/sort/interval precompute.
//
// Replaces (reference, /root/reference/mmdet3d/ops/bev_pool):
//   src/bev_pool_cuda.cu:20-42   bev_pool_kernel       (thread per (interval, channel), serial walk)
//   src/bev_pool_cuda.cu:61-84   bev_pool_grad_kernel
//   src/bev_pool_cpu.cpp:22-87   host wrappers (zero-filled outputs)
//   bev_pool.py:37-62,83-93      rank formula, argsort, interval construction
//
// Design (HBM-bound: 1 add per 4 bytes read, no reuse):
//   * one 64-lane wavefront per BEV cell / interval; a feature row (C channels) is
//     covered by C/4 lanes holding one 16-byte vector each, so floor(64 / (C/4))
//     consecutive rows are fetched by ONE fully coalesced wave instruction
//     (C=80: 20 lanes/row, 3 rows = 960 contiguous bytes per instruction);
//   * each lane keeps 4 independent 16-byte loads in flight, accumulates in fp32 and
//     the row-slots are folded with __shfl at the end: no LDS round trip, no atomics,
//     one store of the finished cell;
//   * two forward flavours:
//       - "intervals" : the reference's contract (pre-sorted rows + interval arrays,
//                       output zero-filled first) — the drop-in;
//       - "cells"     : the native path.  Rows are read THROUGH the sort permutation
//                       (the reference's sorted copy `feats[indices]`, 581 MB read +
//                       581 MB written per frame, never exists) and the grid walks a
//                       CSR over ALL cells, so every output cell — empty ones as zeros —
//                       is written exactly once by a single launch: no memset pass,
//                       no interval compaction, no host sync.
//   * 64-bit address arithmetic throughout (reference overflows at N*C > 2^31).
#include "common.h"

namespace bevamd {

struct alignas(16) U4 { uint32_t x, y, z, w; };

template <int VEC> struct Acc { float v[VEC]; };

__device__ __forceinline__ void acc_add(Acc<4>& a, const float4& f) {
  a.v[0] += f.x; a.v[1] += f.y; a.v[2] += f.z; a.v[3] += f.w;
}
// 8 packed bf16 -> fp32 adds (bf16 -> f32 is a 16-bit left shift)
__device__ __forceinline__ void acc_add(Acc<8>& a, const U4& u) {
  a.v[0] += __uint_as_float(u.x << 16); a.v[1] += __uint_as_float(u.x & 0xFFFF0000u);
  a.v[2] += __uint_as_float(u.y << 16); a.v[3] += __uint_as_float(u.y & 0xFFFF0000u);
  a.v[4] += __uint_as_float(u.z << 16); a.v[5] += __uint_as_float(u.z & 0xFFFF0000u);
  a.v[6] += __uint_as_float(u.w << 16); a.v[7] += __uint_as_float(u.w & 0xFFFF0000u);
}

struct BevDims { int B, D, H, W, C; };

// cell -> element offset of out[b, z, x, y, 0]   (bev_pool_cuda.cu:33-35)
__device__ __forceinline__ size_t cell_offset(int gx, int gy, int gz, int gb, const BevDims& s) {
  return ((((size_t)gb * s.D + gz) * s.H + gx) * s.W + gy) * (size_t)s.C;
}

// rank = x*(W*D*B) + y*(D*B) + z*B + b   (bev_pool.py:86-91) -> (x,y,z,b)
__device__ __forceinline__ void decode_rank(uint32_t r, const BevDims& s, int& gx, int& gy, int& gz, int& gb) {
  gb = r % s.B; r /= s.B;
  gz = r % s.D; r /= s.D;
  gy = r % s.W; r /= s.W;
  gx = r;
}

// Sum rows [0,len) of one cell.  The calling lane takes rows r0, r0+step, ... and the 16-byte
// column vector cv; U independent loads in flight per lane.  INDEXED: row r is ord[r], else rows
// are contiguous from `first_row`.
__device__ __forceinline__ void opaque_use(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void opaque_use(U4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void set_zero(float4& v) { v = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void set_zero(U4& v) { v.x = v.y = v.z = v.w = 0u; }

// TAILP: the < U rows a lane has left after the unrolled trips go out as ONE batch of predicated loads (row indices first,
// then the rows; a missing row adds +0: the same bits as the one-at-a-time loop) instead of one load per round trip — most
// cells of a camera frustum are shorter than one unrolled trip, so for them this IS the whole walk.
template <typename VecT, int VEC, int U, bool INDEXED, bool TAILP = false>
__device__ __forceinline__ void accumulate_rows(Acc<VEC>& acc, const VecT* __restrict__ x,
                                                const uint32_t* __restrict__ ord, size_t first_row, int len,
                                                int r0, int step, int cv, int lpr) {
  int r = r0;
  for (; r + (U - 1) * step < len; r += U * step) {
    VecT a[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      size_t row = INDEXED ? (size_t)ord[r + u * step] : first_row + (size_t)(r + u * step);
      a[u] = x[row * lpr + cv];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc_add(acc, a[u]);
  }
  if constexpr (TAILP && U > 1) {
    if (r < len) {
      uint32_t rows[U - 1];
#pragma unroll
      for (int u = 0; u < U - 1; ++u) {
        const int rr = r + u * step;
        const int rc = rr < len ? rr : len - 1;                       // clamped: the index load itself is unconditional
        rows[u] = INDEXED ? ord[rc] : (uint32_t)(first_row + (size_t)rc);
      }
#pragma unroll
      for (int u = 0; u < U - 1; ++u) asm volatile("" : "+v"(rows[u]));   // (hipcc would sink an index load into its row's branch)
      VecT a[U - 1];
#pragma unroll
      for (int u = 0; u < U - 1; ++u) {
        set_zero(a[u]);
        if (r + u * step < len) a[u] = x[(size_t)rows[u] * lpr + cv];
      }
#pragma unroll
      for (int u = 0; u < U - 1; ++u) opaque_use(a[u]);                // every load issued before the first add waits
#pragma unroll
      for (int u = 0; u < U - 1; ++u) acc_add(acc, a[u]);
    }
  } else {
    for (; r < len; r += step) {
      size_t row = INDEXED ? (size_t)ord[r] : first_row + (size_t)r;
      VecT a0 = x[row * lpr + cv];
      acc_add(acc, a0);
    }
  }
}

// fold the row-slots into slot 0.  Wave-uniform trip count (every lane shuffles); the partials
// are read-only during the fold — lanes of other slots receive wrapped-around garbage in `tot`
// that nobody stores.
template <int VEC>
__device__ __forceinline__ void fold_slots(Acc<VEC>& acc, int lane, int lpr, int rpi) {
  Acc<VEC> tot = acc;
  for (int sl = 1; sl < rpi; ++sl) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) tot.v[j] += __shfl(acc.v[j], lane + sl * lpr, 64);
  }
  acc = tot;
}

template <int VEC>
__device__ __forceinline__ void store_cell(const Acc<VEC>& acc, float* __restrict__ cell, int cv) {
  float4* o = (float4*)(cell + (size_t)cv * VEC);
#pragma unroll
  for (int j = 0; j < VEC / 4; ++j)
    o[j] = make_float4(acc.v[4 * j], acc.v[4 * j + 1], acc.v[4 * j + 2], acc.v[4 * j + 3]);
}

// ---------------------------------------------------------------------------
// forward over the reference's interval arrays (drop-in contract)
// ---------------------------------------------------------------------------
template <typename VecT, int VEC>
__global__ __launch_bounds__(256) void bev_pool_fwd_intervals_vec_kernel(
    const VecT* __restrict__ x, const int* __restrict__ geom, const int* __restrict__ starts,
    const int* __restrict__ lengths, int n_int, float* __restrict__ out, int lpr, int rpi, BevDims s) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= n_int) return;
  const int lane = threadIdx.x & 63;
  const int slot = lane / lpr;
  const int cv = lane - slot * lpr;
  const int start = starts[wave];
  const int len = lengths[wave];
  Acc<VEC> acc;
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc.v[j] = 0.f;
  if (slot < rpi) accumulate_rows<VecT, VEC, 4, false>(acc, x, nullptr, (size_t)start, len, slot, rpi, cv, lpr);
  fold_slots<VEC>(acc, lane, lpr, rpi);
  if (slot == 0) {
    const int* g = geom + (size_t)start * 4;
    store_cell<VEC>(acc, out + cell_offset(g[0], g[1], g[2], g[3], s), cv);
  }
}

// any channel count / alignment: wave per interval, lanes stride the channels
template <bool IS_BF16>
__global__ __launch_bounds__(256) void bev_pool_fwd_intervals_scalar_kernel(
    const void* __restrict__ xv, const int* __restrict__ geom, const int* __restrict__ starts,
    const int* __restrict__ lengths, int n_int, float* __restrict__ out, BevDims s) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= n_int) return;
  const int lane = threadIdx.x & 63;
  const int start = starts[wave], len = lengths[wave];
  const int* g = geom + (size_t)start * 4;
  float* o = out + cell_offset(g[0], g[1], g[2], g[3], s);
  for (int c = lane; c < s.C; c += 64) {
    float psum = 0.f;
    for (int r = 0; r < len; ++r) {
      size_t e = (size_t)(start + r) * s.C + c;
      psum += IS_BF16 ? __uint_as_float((uint32_t)((const uint16_t*)xv)[e] << 16) : ((const float*)xv)[e];
    }
    o[c] = psum;
  }
}

// ---------------------------------------------------------------------------
// forward over the cell CSR (native path): one launch writes every output cell
// ---------------------------------------------------------------------------
// SW > 0 = XCD-striped walk.  Workgroups go to the 8 XCDs round-robin (workgroup i -> XCD i % 8) and every XCD has its own
// L2, so in plain order the half cache lines that the row runs of neighbouring cells share (a 320-byte row is 2.5 lines) are
// fetched from HBM once per XCD that needs them.  Here the launch is one padded line of `pb` workgroups per (frame, grid row
// x), pb % 8 == 0, and the 4-cell groups of a row are dealt so that stripe t (SW neighbouring groups) of EVERY grid row runs
// on XCD t % 8: the cells on either side of a run end — the same stripe one grid row on, or the neighbouring group of the
// stripe — then meet in ONE L2.  A row's groups rarely divide by 8 * SW (360 cells = 90 groups): the stripes left over after
// the last full round of 8 are dealt from a start that rotates with the line, so that every XCD gets the same number of
// stripes over 8 lines (dealt from XCD 0 every time, two XCDs would carry 12 groups a row against 11).  The map is host-callable:
// tests/test_bev_pool_striped_map.py checks it without a GPU (every group once, the XCD of a stripe, the balance, the rotation).
template <int SW>
__host__ __device__ __forceinline__ bool striped_group(uint32_t p, uint32_t line, uint32_t rb, uint32_t shift, uint32_t& j) {
  constexpr uint32_t G = 8u * SW;
  const uint32_t stripes = (rb + SW - 1) / SW, full = stripes & ~7u, extra = stripes - full;
  const uint32_t c = p / G, q = p - c * G, m = q >> 3;               // chunk of 8 stripes, member of the stripe
  const uint32_t x = ((q & 7u) - shift) & 7u;                        // XCD q % 8 takes stripe x of the chunk
  uint32_t st = c * 8u + x;
  if (st >= full) {                                                  // the last, partial round
    const uint32_t k = (x - line * extra) & 7u;
    if (k >= extra) return false;
    st = full + k;
  }
  j = st * SW + m;
  return j < rb;
}

// TX > 1 = a workgroup of 4 * TX waves takes a TX x 4 tile of cells (TX neighbouring grid rows): the lines shared across
// a grid-row boundary inside the tile are asked for by ONE compute unit, whichever XCD it sits on, and the launch order stays
// the plain one.
template <typename VecT, int VEC, int U, int SW = 0, bool TAILP = false, int TX = 1>
__global__ __launch_bounds__(256 * TX) void bev_pool_fwd_cells_vec_kernel(
    const VecT* __restrict__ x, const uint32_t* __restrict__ order, const uint32_t* __restrict__ cell_start,
    uint32_t ncells, float* __restrict__ out, int lpr, int rpi, BevDims s, uint32_t pb = 0, uint32_t rot_rows = 0) {
  // cells are numbered b-fastest (the reference's rank); waves walk them frame-major, so that neighbouring waves read one
  // frame's slab of the feature volume and write neighbouring output rows (ncells = B*D*H*W)
  uint32_t cell;
  if constexpr (TX > 1) {
    static_assert(SW == 0, "tiles or stripes");
    const uint32_t w = threadIdx.x >> 6, tx = w >> 2, ty = w & 3u;
    const uint32_t row_cells = (uint32_t)s.W * (uint32_t)s.D, rb = (row_cells + 3u) >> 2;
    const uint32_t lp = ((uint32_t)s.H + TX - 1) / TX;               // tile rows per frame
    const uint32_t lineg = blockIdx.x / rb, j = blockIdx.x - lineg * rb;
    const uint32_t f = lineg / lp, xr = (lineg - f * lp) * TX + tx;
    const uint32_t in_row = j * 4u + ty;
    if (xr >= (uint32_t)s.H || in_row >= row_cells) return;
    cell = (xr * row_cells + in_row) * (uint32_t)s.B + f;
  } else if constexpr (SW == 0) {
    const uint32_t lin = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (lin >= ncells) return;
    const uint32_t per_frame = ncells / (uint32_t)s.B;
    cell = (lin % per_frame) * (uint32_t)s.B + lin / per_frame;
  } else {
    const uint32_t line = blockIdx.x / pb, p = blockIdx.x - line * pb;   // line = frame * H + grid row
    const uint32_t row_cells = (uint32_t)s.W * (uint32_t)s.D;
    uint32_t j;
    // rot_rows > 0: the stripe -> XCD map moves on by one XCD every rot_rows lines (a hot stripe — the cells beside the ego
    // vehicle — then visits every XCD instead of loading one; the lines where it moves lose the sharing with the line before)
    if (!striped_group<SW>(p, line, (row_cells + 3u) >> 2, rot_rows ? line / rot_rows : 0u, j)) return;
    const uint32_t in_row = j * 4u + (threadIdx.x >> 6);
    if (in_row >= row_cells) return;
    const uint32_t f = line / (uint32_t)s.H, xr = line - f * (uint32_t)s.H;
    cell = (xr * row_cells + in_row) * (uint32_t)s.B + f;
  }
  const int lane = threadIdx.x & 63;
  const int slot = lane / lpr;
  const int cv = lane - slot * lpr;
  const uint32_t start = cell_start[cell];
  const int len = (int)(cell_start[cell + 1] - start);
  Acc<VEC> acc;
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc.v[j] = 0.f;
  if (len > 0) {  // wave-uniform
    if (slot < rpi) accumulate_rows<VecT, VEC, U, true, TAILP>(acc, x, order + start, 0, len, slot, rpi, cv, lpr);
    fold_slots<VEC>(acc, lane, lpr, rpi);
  }
  if (slot == 0) {
    int gx, gy, gz, gb;
    decode_rank(cell, s, gx, gy, gz, gb);
    store_cell<VEC>(acc, out + cell_offset(gx, gy, gz, gb, s), cv);
  }
}

// Cooperative flavour: a workgroup of NW waves owns NW consecutive cells and EVERY wave works on
// EVERY cell (row r of a cell goes to wave (r / rpi) % NW), so a 900-row cell is walked by NW*rpi
// row-slots at once instead of rpi: the longest dependent-load chain — which bounds the kernel when
// cell populations are heavy-tailed — shrinks NW-fold, and each lane has loads of up to NW cells in
// flight.  Partials meet in LDS once; wave w finishes and stores cell w.
template <typename VecT, int VEC, int U, int NW>
__global__ __launch_bounds__(NW * 64) void bev_pool_fwd_cells_coop_kernel(
    const VecT* __restrict__ x, const uint32_t* __restrict__ order, const uint32_t* __restrict__ cell_start,
    uint32_t ncells, float* __restrict__ out, int lpr, int rpi, BevDims s) {
  extern __shared__ __attribute__((aligned(16))) float part[];  // [NW cells][NW waves][lpr][VEC]
  const uint32_t cell0 = blockIdx.x * (uint32_t)NW;
  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int slot = lane / lpr;
  const int cv = lane - slot * lpr;

  uint32_t cs[NW + 1];
#pragma unroll
  for (int q = 0; q <= NW; ++q) {
    uint32_t c = cell0 + (uint32_t)q;
    cs[q] = cell_start[c < ncells ? c : ncells];
  }
  Acc<VEC> acc[NW];
#pragma unroll
  for (int q = 0; q < NW; ++q)
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[q].v[j] = 0.f;

  const bool any = cs[NW] != cs[0];  // block-uniform
  if (any) {
    if (slot < rpi) {
#pragma unroll
      for (int q = 0; q < NW; ++q)
        accumulate_rows<VecT, VEC, U, true>(acc[q], x, order + cs[q], 0, (int)(cs[q + 1] - cs[q]),
                                            wave * rpi + slot, NW * rpi, cv, lpr);
    }
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      fold_slots<VEC>(acc[q], lane, lpr, rpi);
      if (slot == 0) {
        float4* p = (float4*)(part + ((size_t)(q * NW + wave) * lpr + cv) * VEC);
#pragma unroll
        for (int j = 0; j < VEC / 4; ++j)
          p[j] = make_float4(acc[q].v[4 * j], acc[q].v[4 * j + 1], acc[q].v[4 * j + 2], acc[q].v[4 * j + 3]);
      }
    }
    __syncthreads();
  }
  const uint32_t cell = cell0 + (uint32_t)wave;
  if (cell < ncells && slot == 0) {
    Acc<VEC> tot;
#pragma unroll
    for (int j = 0; j < VEC; ++j) tot.v[j] = 0.f;
    if (any) {
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const float4* p = (const float4*)(part + ((size_t)(wave * NW + w) * lpr + cv) * VEC);
#pragma unroll
        for (int j = 0; j < VEC / 4; ++j) {
          float4 t = p[j];
          tot.v[4 * j] += t.x; tot.v[4 * j + 1] += t.y; tot.v[4 * j + 2] += t.z; tot.v[4 * j + 3] += t.w;
        }
      }
    }
    int gx, gy, gz, gb;
    decode_rank(cell, s, gx, gy, gz, gb);
    store_cell<VEC>(tot, out + cell_offset(gx, gy, gz, gb, s), cv);
  }
}

template <bool IS_BF16>
__global__ __launch_bounds__(256) void bev_pool_fwd_cells_scalar_kernel(
    const void* __restrict__ xv, const uint32_t* __restrict__ order, const uint32_t* __restrict__ cell_start,
    uint32_t ncells, float* __restrict__ out, BevDims s) {
  const uint32_t cell = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (cell >= ncells) return;
  const int lane = threadIdx.x & 63;
  const uint32_t start = cell_start[cell];
  const int len = (int)(cell_start[cell + 1] - start);
  int gx, gy, gz, gb;
  decode_rank(cell, s, gx, gy, gz, gb);
  float* o = out + cell_offset(gx, gy, gz, gb, s);
  for (int c = lane; c < s.C; c += 64) {
    float psum = 0.f;
    for (int r = 0; r < len; ++r) {
      size_t e = (size_t)order[start + r] * s.C + c;
      psum += IS_BF16 ? __uint_as_float((uint32_t)((const uint16_t*)xv)[e] << 16) : ((const float*)xv)[e];
    }
    o[c] = psum;
  }
}

// ---------------------------------------------------------------------------
// backward: broadcast the cell gradient to every point of its interval
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bev_pool_bwd_intervals_vec_kernel(
    const float* __restrict__ out_grad, const int* __restrict__ geom, const int* __restrict__ starts,
    const int* __restrict__ lengths, int n_int, float4* __restrict__ x_grad, int lpr, int rpi, BevDims s) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= n_int) return;
  const int lane = threadIdx.x & 63;
  const int slot = lane / lpr;
  const int cv = lane - slot * lpr;
  if (slot >= rpi) return;
  const int start = starts[wave], len = lengths[wave];
  const int* g = geom + (size_t)start * 4;
  const float4 gval = *(const float4*)(out_grad + cell_offset(g[0], g[1], g[2], g[3], s) + (size_t)cv * 4);
  float4* xr = x_grad + (size_t)start * lpr + cv;
  for (int r = slot; r < len; r += rpi) xr[(size_t)r * lpr] = gval;
}

__global__ __launch_bounds__(256) void bev_pool_bwd_intervals_scalar_kernel(
    const float* __restrict__ out_grad, const int* __restrict__ geom, const int* __restrict__ starts,
    const int* __restrict__ lengths, int n_int, float* __restrict__ x_grad, BevDims s) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= n_int) return;
  const int lane = threadIdx.x & 63;
  const int start = starts[wave], len = lengths[wave];
  const int* g = geom + (size_t)start * 4;
  const float* go = out_grad + cell_offset(g[0], g[1], g[2], g[3], s);
  for (int c = lane; c < s.C; c += 64) {
    float gval = go[c];
    for (int r = 0; r < len; ++r) x_grad[(size_t)(start + r) * s.C + c] = gval;
  }
}

// native backward: row-parallel over the SORTED rows, so a 900-row cell and a 1-row cell
// cost the same per row; rows the range mask dropped (sentinel rank) get zeros.
__global__ __launch_bounds__(256) void bev_pool_bwd_rows_vec_kernel(
    const float* __restrict__ out_grad, const uint32_t* __restrict__ order,
    const uint32_t* __restrict__ ranks_sorted, uint32_t ncells, int n, float4* __restrict__ x_grad, int lpr,
    int rpi, BevDims s) {
  const int lane = threadIdx.x & 63;
  const int slot = lane / lpr;
  const int cv = lane - slot * lpr;
  if (slot >= rpi) return;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t j = wave * rpi + slot;
  if (j >= (size_t)n) return;
  const uint32_t r = ranks_sorted[j];
  float4 gval = make_float4(0.f, 0.f, 0.f, 0.f);
  if (r < ncells) {
    int gx, gy, gz, gb;
    decode_rank(r, s, gx, gy, gz, gb);
    gval = *(const float4*)(out_grad + cell_offset(gx, gy, gz, gb, s) + (size_t)cv * 4);
  }
  x_grad[(size_t)order[j] * lpr + cv] = gval;
}

// The same gradient walked in POINT order: x_grad[i] = out_grad[cell(i)], cell(i) from the plan's cell_of_point (the rank of every
// frustum point in point order, static per plan).  The row-parallel kernel above writes each 320-byte row where the sort put it —
// 2.5 cache lines at a scattered address, half lines left to merge in L2 (measured 2.5-2.6 TB/s = 0.31-0.32 of the HBM peak for
// what is a pure streaming write).  Here consecutive threads write consecutive 16-byte pieces of x_grad — whole lines, in address
// order — and the scattered side is the READ of out_grad (41.5 MB per frame: it lives in L2 / Infinity Cache).  Four pieces per
// thread in flight (index -> gradient -> store chains are independent).
__global__ __launch_bounds__(256) void bev_pool_bwd_points_vec_kernel(const float* __restrict__ out_grad,
                                                                      const uint32_t* __restrict__ cell_of_point,
                                                                      uint32_t ncells, size_t total4, int lpr,
                                                                      float4* __restrict__ x_grad, BevDims s) {
  constexpr int U = 4;
  const size_t base = (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  uint32_t r[U];
  int cv[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t e = base + (size_t)u * 256;
    const size_t i = e / (unsigned)lpr;
    cv[u] = (int)(e - i * (unsigned)lpr);
    r[u] = e < total4 ? cell_of_point[i] : 0xFFFFFFFFu;
  }
  float4 g[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    g[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r[u] < ncells) {
      int gx, gy, gz, gb;
      decode_rank(r[u], s, gx, gy, gz, gb);
      g[u] = *(const float4*)(out_grad + cell_offset(gx, gy, gz, gb, s) + (size_t)cv[u] * 4);
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const size_t e = base + (size_t)u * 256;
    typedef float f32x4n __attribute__((ext_vector_type(4)));
    if (e < total4) __builtin_nontemporal_store(f32x4n{g[u].x, g[u].y, g[u].z, g[u].w}, (f32x4n*)&x_grad[e]);
  }
}

__global__ __launch_bounds__(256) void bev_pool_bwd_rows_scalar_kernel(
    const float* __restrict__ out_grad, const uint32_t* __restrict__ order,
    const uint32_t* __restrict__ ranks_sorted, uint32_t ncells, int n, float* __restrict__ x_grad, BevDims s) {
  const int lane = threadIdx.x & 63;
  const size_t j = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (j >= (size_t)n) return;
  const uint32_t r = ranks_sorted[j];
  const float* go = nullptr;
  if (r < ncells) {
    int gx, gy, gz, gb;
    decode_rank(r, s, gx, gy, gz, gb);
    go = out_grad + cell_offset(gx, gy, gz, gb, s);
  }
  float* xr = x_grad + (size_t)order[j] * s.C;
  for (int c = lane; c < s.C; c += 64) xr[c] = go ? go[c] : 0.f;
}

// ---------------------------------------------------------------------------
// precompute: rank (+ per-cell histogram) -> stable sort -> CSR / interval boundaries
// ---------------------------------------------------------------------------
// key = reference rank for in-range cells; `ncells` (sentinel) for anything outside
// [0,H)x[0,W)x[0,D)x[0,B), so a caller may hand over unfiltered coordinates: the range
// mask of vtransforms/base.py:160-169 folds into the sort instead of a boolean gather.
template <typename CoordT>
__global__ __launch_bounds__(256) void bev_rank_kernel(const CoordT* __restrict__ coords, int n, BevDims s,
                                                       uint32_t ncells, uint32_t* __restrict__ keys,
                                                       uint32_t* __restrict__ vals) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const CoordT* c = coords + (size_t)i * 4;
  long long gx = c[0], gy = c[1], gz = c[2], gb = c[3];
  bool ok = gx >= 0 && gx < s.H && gy >= 0 && gy < s.W && gz >= 0 && gz < s.D && gb >= 0 && gb < s.B;
  uint32_t r = ok ? (uint32_t)(((gx * s.W + gy) * s.D + gz) * s.B + gb) : ncells;
  keys[i] = r;
  vals[i] = (uint32_t)i;
}

// Same, straight from fp32 lidar-frame geometry (vtransforms/base.py:149):
//   idx = trunc((p - (bx - dx/2)) / dx)  as int64 (C++ cast semantics == .long()),
// batch index = i / points_per_batch.
__global__ __launch_bounds__(256) void bev_rank_from_geom_kernel(const float* __restrict__ geom, int n,
                                                                 int points_per_batch, float ox, float oy,
                                                                 float oz, float dx, float dy, float dz,
                                                                 BevDims s, uint32_t ncells,
                                                                 uint32_t* __restrict__ keys,
                                                                 uint32_t* __restrict__ vals) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float* p = geom + (size_t)i * 3;
  // the subtraction and the division are separate fp32 roundings, as in torch
  float fx = __fdiv_rn(__fsub_rn(p[0], ox), dx);
  float fy = __fdiv_rn(__fsub_rn(p[1], oy), dy);
  float fz = __fdiv_rn(__fsub_rn(p[2], oz), dz);
  // truncation toward zero; NaN / huge values fall out of range below
  bool finite = (fabsf(fx) < 1.0e9f) && (fabsf(fy) < 1.0e9f) && (fabsf(fz) < 1.0e9f);
  long long gx = finite ? (long long)fx : -1, gy = finite ? (long long)fy : -1,
            gz = finite ? (long long)fz : -1;
  int gb = i / points_per_batch;
  bool ok = gx >= 0 && gx < s.H && gy >= 0 && gy < s.W && gz >= 0 && gz < s.D && gb < s.B;
  uint32_t r = ok ? (uint32_t)(((gx * s.W + gy) * s.D + gz) * s.B + gb) : ncells;
  keys[i] = r;
  vals[i] = (uint32_t)i;
}

// CSR over rank-ordered cells from the sorted keys: cell_start[c] = lower_bound(keys, c).
// (A histogram with atomics costs 2 ms here: 178 k dropped rows hammer one sentinel counter.)
__global__ __launch_bounds__(256) void bev_cell_start_kernel(const uint32_t* __restrict__ keys, uint32_t n,
                                                             uint32_t ncells, uint32_t* __restrict__ cell_start) {
  uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c > ncells + 1u) return;
  uint32_t lo = 0, hi = n;
  if (c > ncells) lo = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (keys[mid] < c) lo = mid + 1; else hi = mid;
  }
  cell_start[c] = lo;
}

// interval arrays in the reference's shape, compacted from the CSR: interval k is the k-th
// non-empty cell in rank order (== bev_pool.py:39-46 on the sorted ranks).
__global__ __launch_bounds__(256) void bev_cell_flags_kernel(const uint32_t* __restrict__ cell_start,
                                                             uint32_t ncells, uint32_t* __restrict__ flags) {
  uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= ncells) return;
  flags[c] = cell_start[c + 1] > cell_start[c] ? 1u : 0u;
}

__global__ __launch_bounds__(256) void bev_intervals_from_cells_kernel(const uint32_t* __restrict__ cell_start,
                                                                       const uint32_t* __restrict__ flag_scan,
                                                                       uint32_t ncells, int* __restrict__ starts,
                                                                       int* __restrict__ lengths) {
  uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c >= ncells) return;
  uint32_t a = cell_start[c], b = cell_start[c + 1];
  if (b > a) {
    uint32_t k = flag_scan[c];
    starts[k] = (int)a;
    lengths[k] = (int)(b - a);
  }
}

// sorted coordinate rows (x,y,z,b) int32 for the reference-shaped API
__global__ __launch_bounds__(256) void bev_geom_from_ranks_kernel(const uint32_t* __restrict__ keys, int n,
                                                                  BevDims s, uint32_t ncells,
                                                                  int* __restrict__ geom) {
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t k = keys[j];
  int gx = -1, gy = -1, gz = -1, gb = -1;
  if (k < ncells) decode_rank(k, s, gx, gy, gz, gb);
  ((int4*)geom)[j] = make_int4(gx, gy, gz, gb);
}

// ---------------------------------------------------------------------------
// host-side dispatch
// ---------------------------------------------------------------------------
static int check_dims(int n, int c, int b, int d, int h, int w) {
  BEVAMD_REQUIRE(n >= 0 && c > 0 && b > 0 && d > 0 && h > 0 && w > 0,
                 "bev_pool: bad sizes n=%d c=%d b=%d d=%d h=%d w=%d", n, c, b, d, h, w);
  BEVAMD_REQUIRE((unsigned long long)b * d * h * w < 0xFFFFFFF0ull, "bev_pool: B*D*H*W must be < 2^32-16");
  return BEVAMD_OK;
}

static bool vec_path_ok(int C, int vec, const void* a, const void* b) {
  return (C % vec == 0) && (C / vec <= 64) && (((uintptr_t)a & 15) == 0) && (((uintptr_t)b & 15) == 0);
}

static int launch_forward_intervals(const void* x, int x_is_bf16, const int* geom, const int* starts,
                                    const int* lengths, int n_int, float* out, BevDims s, hipStream_t stream) {
  if (n_int <= 0) return BEVAMD_OK;
  const int vec = x_is_bf16 ? 8 : 4;
  dim3 grid(cdiv(n_int, 4)), block(256);
  if (vec_path_ok(s.C, vec, x, out)) {
    int lpr = s.C / vec, rpi = 64 / lpr;
    if (x_is_bf16)
      bev_pool_fwd_intervals_vec_kernel<U4, 8><<<grid, block, 0, stream>>>((const U4*)x, geom, starts, lengths,
                                                                          n_int, out, lpr, rpi, s);
    else
      bev_pool_fwd_intervals_vec_kernel<float4, 4><<<grid, block, 0, stream>>>((const float4*)x, geom, starts,
                                                                              lengths, n_int, out, lpr, rpi, s);
  } else {
    if (x_is_bf16)
      bev_pool_fwd_intervals_scalar_kernel<true><<<grid, block, 0, stream>>>(x, geom, starts, lengths, n_int, out, s);
    else
      bev_pool_fwd_intervals_scalar_kernel<false><<<grid, block, 0, stream>>>(x, geom, starts, lengths, n_int, out, s);
  }
  BEVAMD_LAUNCH_CHECK("bev_pool_fwd_intervals");
  return BEVAMD_OK;
}

static size_t prepare_ws_bytes(size_t n, size_t ncells) {
  size_t b = 0;
  b += 2 * align_up(n * sizeof(uint32_t), 256);             // keys_a, vals_a
  b += 2 * align_up((ncells + 2) * sizeof(uint32_t), 256);  // cell flags, flag scan
  size_t s1 = radix_sort_workspace_bytes(n), s2 = scan_workspace_bytes(ncells + 2);
  b += align_up(s1 > s2 ? s1 : s2, 256);
  return b;
}

struct PrepBuffers {
  uint32_t *keys_a, *vals_a, *flags, *flag_scan;
  void* sort_ws; size_t sort_ws_bytes;
};

static int carve_prep(PrepBuffers& pb, int n, uint32_t ncells, void* ws, size_t ws_bytes) {
  size_t need = prepare_ws_bytes((size_t)n, ncells);
  if (ws == nullptr || ws_bytes < need) {
    set_error("bev_pool_prepare: workspace too small (%zu < %zu)", ws_bytes, need);
    return BEVAMD_ERR_WORKSPACE;
  }
  Carver cv(ws, ws_bytes);
  pb.keys_a = cv.take<uint32_t>(n);
  pb.vals_a = cv.take<uint32_t>(n);
  pb.flags = cv.take<uint32_t>((size_t)ncells + 2);
  pb.flag_scan = cv.take<uint32_t>((size_t)ncells + 2);
  pb.sort_ws = cv.base + cv.off;
  pb.sort_ws_bytes = ws_bytes - cv.off;
  return BEVAMD_OK;
}

// after keys/vals/cell_count are filled: sort, CSR, optional reference-shaped arrays
static int prepare_tail(PrepBuffers& pb, int n, BevDims s, uint32_t ncells, uint32_t* ranks_sorted,
                        uint32_t* order, uint32_t* cell_start, int* starts, int* lengths, int* n_int_dev,
                        int* geom_sorted, hipStream_t stream) {
  const int nbits = bits_for((uint64_t)ncells + 1);
  int rc = radix_sort_pairs_u32(pb.keys_a, pb.vals_a, ranks_sorted, order, (size_t)n, nbits, pb.sort_ws,
                                pb.sort_ws_bytes, stream);
  if (rc != BEVAMD_OK) return rc;
  bev_cell_start_kernel<<<dim3(cdiv((long long)ncells + 2, 256)), dim3(256), 0, stream>>>(ranks_sorted, (uint32_t)n,
                                                                                           ncells, cell_start);
  BEVAMD_LAUNCH_CHECK("bev_cell_start");
  if (starts && lengths && n_int_dev) {
    dim3 grid(cdiv(ncells, 256)), block(256);
    bev_cell_flags_kernel<<<grid, block, 0, stream>>>(cell_start, ncells, pb.flags);
    BEVAMD_LAUNCH_CHECK("bev_cell_flags");
    rc = exclusive_scan_u32(pb.flags, pb.flag_scan, (size_t)ncells, (uint32_t*)n_int_dev, pb.sort_ws,
                            pb.sort_ws_bytes, stream);
    if (rc != BEVAMD_OK) return rc;
    bev_intervals_from_cells_kernel<<<grid, block, 0, stream>>>(cell_start, pb.flag_scan, ncells, starts, lengths);
    BEVAMD_LAUNCH_CHECK("bev_intervals_from_cells");
  }
  if (geom_sorted) {
    bev_geom_from_ranks_kernel<<<dim3(cdiv(n, 256)), dim3(256), 0, stream>>>(ranks_sorted, n, s, ncells, geom_sorted);
    BEVAMD_LAUNCH_CHECK("bev_geom_from_ranks");
  }
  return BEVAMD_OK;
}

// variant: 0 = default (tuned), 1 = wave-per-cell U4, 2 = wave-per-cell U8,
//          3 = coop NW4 U4, 4 = coop NW4 U8, 5 = coop NW8 U4, 6 = coop NW8 U2, 7 = coop NW4 U2,
//          14 / 15 = wave-per-cell U4 / U8 with the tail rows as one predicated batch (the defaults: bf16 / fp32),
//          16 / 17 = XCD-striped walk, stripes of 2 groups, U4 / U8, batched tail;  18 / 19 = stripes of 1 group;
//          + 100 * R on 16-19: the stripe -> XCD map moves on every R lines;  23 / 25 = 2 x 4 cell tiles, U8 / U4.
//          (the striped and tiled walks cut the L2 -> fabric fetch counter by 3-7 % and are SLOWER: EXPERIMENTS C.7)
// BEVAMD_BEV_POOL_LDS_PAD = bytes of (unused) dynamic LDS per 4-wave workgroup of the default kernel: an occupancy cap (160 KB per
// CU / pad workgroups) for schedules in which the kernel shares the machine (round 6 experiment; 0 = none)
static size_t lds_pad_bytes() {
  static long v = -1;
  if (v < 0) { const char* e = getenv("BEVAMD_BEV_POOL_LDS_PAD"); v = e ? atol(e) : 0; v = v < 0 ? 0 : v > 160 * 1024 ? 160 * 1024 : v; }
  return (size_t)v;
}

template <typename VecT, int VEC>
static int launch_cells_vec(const void* x, const uint32_t* order, const uint32_t* cell_start, uint32_t ncells,
                            float* out, int lpr, int rpi, BevDims s, int variant, hipStream_t stream) {
  const VecT* xv = (const VecT*)x;
#define BEVAMD_COOP(U, NW)                                                                                 \
  bev_pool_fwd_cells_coop_kernel<VecT, VEC, U, NW>                                                         \
      <<<dim3(cdiv(ncells, NW)), dim3(NW * 64), (size_t)NW * NW * lpr * VEC * sizeof(float), stream>>>(    \
          xv, order, cell_start, ncells, out, lpr, rpi, s)
  const uint32_t rot = (uint32_t)(variant / 100);   // striped walks: variant + 100 * R = the stripe map moves on every R lines
  switch (variant % 100) {
    case 1: bev_pool_fwd_cells_vec_kernel<VecT, VEC, 4><<<dim3(cdiv(ncells, 4)), dim3(256), 0, stream>>>(
                xv, order, cell_start, ncells, out, lpr, rpi, s); break;
    case 2: bev_pool_fwd_cells_vec_kernel<VecT, VEC, 8><<<dim3(cdiv(ncells, 4)), dim3(256), 0, stream>>>(
                xv, order, cell_start, ncells, out, lpr, rpi, s); break;
#define BEVAMD_STRIPED(U, SW, TP)                                                                                     \
  {                                                                                                                   \
    const uint32_t pb = cdiv(cdiv(cdiv((uint32_t)s.W * (uint32_t)s.D, 4u), SW), 8u) * 8u * SW;                       \
    bev_pool_fwd_cells_vec_kernel<VecT, VEC, U, SW, TP>                                                               \
        <<<dim3(pb * (uint32_t)s.H * (uint32_t)s.B), dim3(256), 0, stream>>>(xv, order, cell_start, ncells, out, lpr, rpi, s, pb, rot); \
  }
    case 14: bev_pool_fwd_cells_vec_kernel<VecT, VEC, 4, 0, true><<<dim3(cdiv(ncells, 4)), dim3(256), 0, stream>>>(
                 xv, order, cell_start, ncells, out, lpr, rpi, s); break;
    case 15: bev_pool_fwd_cells_vec_kernel<VecT, VEC, 8, 0, true><<<dim3(cdiv(ncells, 4)), dim3(256), lds_pad_bytes(), stream>>>(
                 xv, order, cell_start, ncells, out, lpr, rpi, s); break;
#ifdef BEVAMD_PROFILING   // measured slower than the plain walk and rejected (EXPERIMENTS C.7): kept for sweeps, not shipped
    case 16: BEVAMD_STRIPED(4, 2, true); break;
    case 17: BEVAMD_STRIPED(8, 2, true); break;
    case 18: BEVAMD_STRIPED(4, 1, true); break;
    case 19: BEVAMD_STRIPED(8, 1, true); break;
#endif
#define BEVAMD_TILED(U, TX)                                                                                           \
  {                                                                                                                   \
    const uint32_t rb = cdiv((uint32_t)s.W * (uint32_t)s.D, 4u), lp = cdiv((uint32_t)s.H, TX);                        \
    bev_pool_fwd_cells_vec_kernel<VecT, VEC, U, 0, true, TX>                                                          \
        <<<dim3(rb * lp * (uint32_t)s.B), dim3(256 * TX), 0, stream>>>(xv, order, cell_start, ncells, out, lpr, rpi, s); \
  }
#ifdef BEVAMD_PROFILING
    case 23: BEVAMD_TILED(8, 2); break;
    case 25: BEVAMD_TILED(4, 2); break;
#else
    case 16: case 17: case 18: case 19: case 23: case 25:
      set_error("bev_pool_forward_cells: variant %d (XCD-striped / tiled walk) is compiled into -DBEVAMD_PROFILING builds only", variant);
      return BEVAMD_ERR_UNSUPPORTED;
#endif
#undef BEVAMD_TILED
#undef BEVAMD_STRIPED
    case 3: BEVAMD_COOP(4, 4); break;
    case 4: BEVAMD_COOP(8, 4); break;
    case 5: BEVAMD_COOP(4, 8); break;
    case 6: BEVAMD_COOP(2, 8); break;
    case 7: BEVAMD_COOP(2, 4); break;
    default: set_error("bev_pool_forward_cells: unknown variant %d", variant); return BEVAMD_ERR_INVALID_ARG;
  }
#undef BEVAMD_COOP
  BEVAMD_LAUNCH_CHECK("bev_pool_fwd_cells");
  return BEVAMD_OK;
}

}  // namespace bevamd

using namespace bevamd;

extern "C" {

int bevamd_bev_pool_forward(const float* x, const int* geom_feats, const int* interval_lengths,
                            const int* interval_starts, float* out, int n, int c, int n_intervals, int b,
                            int d, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_dims(n, c, b, d, h, w);
  if (rc) return rc;
  BEVAMD_REQUIRE(n_intervals >= 0, "bev_pool_forward: n_intervals < 0");
  BEVAMD_REQUIRE(out != nullptr, "bev_pool_forward: out is null");
  BEVAMD_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)b * d * h * w * c * sizeof(float), stream));
  if (n == 0 || n_intervals == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(x && geom_feats && interval_lengths && interval_starts, "bev_pool_forward: null input");
  return launch_forward_intervals(x, 0, geom_feats, interval_starts, interval_lengths, n_intervals, out,
                                  BevDims{b, d, h, w, c}, stream);
}

int bevamd_bev_pool_forward_bf16(const uint16_t* x, const int* geom_feats, const int* interval_lengths,
                                 const int* interval_starts, float* out, int n, int c, int n_intervals,
                                 int b, int d, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_dims(n, c, b, d, h, w);
  if (rc) return rc;
  BEVAMD_REQUIRE(n_intervals >= 0, "bev_pool_forward_bf16: n_intervals < 0");
  BEVAMD_REQUIRE(out != nullptr, "bev_pool_forward_bf16: out is null");
  BEVAMD_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)b * d * h * w * c * sizeof(float), stream));
  if (n == 0 || n_intervals == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(x && geom_feats && interval_lengths && interval_starts, "bev_pool_forward_bf16: null input");
  return launch_forward_intervals(x, 1, geom_feats, interval_starts, interval_lengths, n_intervals, out,
                                  BevDims{b, d, h, w, c}, stream);
}

int bevamd_bev_pool_backward(const float* out_grad, const int* geom_feats, const int* interval_lengths,
                             const int* interval_starts, float* x_grad, int n, int c, int n_intervals, int b,
                             int d, int h, int w, int skip_zero_fill, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_dims(n, c, b, d, h, w);
  if (rc) return rc;
  BEVAMD_REQUIRE(n_intervals >= 0, "bev_pool_backward: n_intervals < 0");
  if (n == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(x_grad != nullptr, "bev_pool_backward: x_grad is null");
  if (!skip_zero_fill) BEVAMD_HIP_CHECK(hipMemsetAsync(x_grad, 0, (size_t)n * c * sizeof(float), stream));
  if (n_intervals == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(out_grad && geom_feats && interval_lengths && interval_starts, "bev_pool_backward: null input");
  BevDims s{b, d, h, w, c};
  dim3 grid(cdiv(n_intervals, 4)), block(256);
  if (vec_path_ok(c, 4, out_grad, x_grad)) {
    int lpr = c / 4, rpi = 64 / lpr;
    bev_pool_bwd_intervals_vec_kernel<<<grid, block, 0, stream>>>(out_grad, geom_feats, interval_starts,
                                                                  interval_lengths, n_intervals, (float4*)x_grad,
                                                                  lpr, rpi, s);
  } else {
    bev_pool_bwd_intervals_scalar_kernel<<<grid, block, 0, stream>>>(out_grad, geom_feats, interval_starts,
                                                                     interval_lengths, n_intervals, x_grad, s);
  }
  BEVAMD_LAUNCH_CHECK("bev_pool_bwd_intervals");
  return BEVAMD_OK;
}

size_t bevamd_bev_pool_prepare_workspace_bytes(int n, int b, int d, int h, int w) {
  if (n < 1) n = 1;
  unsigned long long ncells = (unsigned long long)(b > 0 ? b : 1) * (d > 0 ? d : 1) * (h > 0 ? h : 1) * (w > 0 ? w : 1);
  return prepare_ws_bytes((size_t)n, (size_t)ncells);
}

static int prepare_common(const void* coords, int coords_kind /*0 i32, 1 i64, 2 geom f32*/, int n, int b, int d,
                          int h, int w, const float* origin, const float* dx, uint32_t* ranks_sorted,
                          uint32_t* order, uint32_t* cell_start, int* interval_starts, int* interval_lengths,
                          int* n_intervals_dev, int* geom_sorted, void* ws, size_t ws_bytes, hipStream_t stream) {
  int rc = check_dims(n, 1, b, d, h, w);
  if (rc) return rc;
  BEVAMD_REQUIRE(cell_start != nullptr, "bev_pool_prepare: cell_start is null");
  const uint32_t ncells = (uint32_t)((unsigned long long)b * d * h * w);
  if (n == 0) {
    BEVAMD_HIP_CHECK(hipMemsetAsync(cell_start, 0, ((size_t)ncells + 2) * sizeof(uint32_t), stream));
    if (n_intervals_dev) BEVAMD_HIP_CHECK(hipMemsetAsync(n_intervals_dev, 0, sizeof(int), stream));
    return BEVAMD_OK;
  }
  BEVAMD_REQUIRE(coords && ranks_sorted && order, "bev_pool_prepare: null buffer");
  PrepBuffers pb;
  rc = carve_prep(pb, n, ncells, ws, ws_bytes);
  if (rc) return rc;
  BevDims s{b, d, h, w, 1};
  dim3 grid(cdiv(n, 256)), block(256);
  if (coords_kind == 1)
    bev_rank_kernel<long long><<<grid, block, 0, stream>>>((const long long*)coords, n, s, ncells, pb.keys_a,
                                                           pb.vals_a);
  else if (coords_kind == 0)
    bev_rank_kernel<int><<<grid, block, 0, stream>>>((const int*)coords, n, s, ncells, pb.keys_a, pb.vals_a);
  else
    bev_rank_from_geom_kernel<<<grid, block, 0, stream>>>((const float*)coords, n, n / b, origin[0], origin[1],
                                                          origin[2], dx[0], dx[1], dx[2], s, ncells, pb.keys_a,
                                                          pb.vals_a);
  BEVAMD_LAUNCH_CHECK("bev_rank");
  return prepare_tail(pb, n, s, ncells, ranks_sorted, order, cell_start, interval_starts, interval_lengths,
                      n_intervals_dev, geom_sorted, stream);
}

int bevamd_bev_pool_prepare(const void* coords, int coords_are_int64, int n, int b, int d, int h, int w,
                            uint32_t* ranks_sorted, uint32_t* order, uint32_t* cell_start, int* interval_starts,
                            int* interval_lengths, int* n_intervals_dev, int* geom_sorted, void* ws,
                            size_t ws_bytes, void* stream_) {
  return prepare_common(coords, coords_are_int64 ? 1 : 0, n, b, d, h, w, nullptr, nullptr, ranks_sorted, order,
                        cell_start, interval_starts, interval_lengths, n_intervals_dev, geom_sorted, ws, ws_bytes,
                        (hipStream_t)stream_);
}

int bevamd_bev_pool_prepare_from_geom(const float* geom_xyz, int n, int b, int d, int h, int w,
                                      const float* bx_minus_half_dx, const float* dx, uint32_t* ranks_sorted,
                                      uint32_t* order, uint32_t* cell_start, int* interval_starts,
                                      int* interval_lengths, int* n_intervals_dev, int* geom_sorted, void* ws,
                                      size_t ws_bytes, void* stream_) {
  BEVAMD_REQUIRE(bx_minus_half_dx && dx, "bev_pool_prepare_from_geom: null grid origin/step (host pointers)");
  BEVAMD_REQUIRE(b > 0 && n % b == 0, "bev_pool_prepare_from_geom: n=%d not divisible by batch=%d", n, b);
  return prepare_common(geom_xyz, 2, n, b, d, h, w, bx_minus_half_dx, dx, ranks_sorted, order, cell_start,
                        interval_starts, interval_lengths, n_intervals_dev, geom_sorted, ws, ws_bytes,
                        (hipStream_t)stream_);
}

int bevamd_bev_pool_forward_cells_tuned(const void* x, int x_is_bf16, const uint32_t* order,
                                        const uint32_t* cell_start, float* out, int n, int c, int b, int d, int h,
                                        int w, int variant, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_dims(n, c, b, d, h, w);
  if (rc) return rc;
  BEVAMD_REQUIRE(out && cell_start, "bev_pool_forward_cells: null out / cell_start");
  BEVAMD_REQUIRE(n == 0 || (x && order), "bev_pool_forward_cells: null input");
  const uint32_t ncells = (uint32_t)((unsigned long long)b * d * h * w);
  BevDims s{b, d, h, w, c};
  const int vec = x_is_bf16 ? 8 : 4;
  // measured on MI355X (tools/sweep_bev_pool.py, profiles/r05_bev_pool_variant_sweep.txt): one wave per cell in plain launch
  // order with the tail rows as one batch of predicated loads wins everywhere — 8 loads per lane for fp32 rows (3 rows per
  // wave instruction: 954 vs 999 us at 8 frames, 129 vs 131 at one), 4 for bf16 rows (6 rows per instruction: 565 vs 590 us).
  if (variant == 0) variant = x_is_bf16 ? 14 : 15;
  if (vec_path_ok(c, vec, x, out)) {
    int lpr = c / vec, rpi = 64 / lpr;
    return x_is_bf16 ? launch_cells_vec<U4, 8>(x, order, cell_start, ncells, out, lpr, rpi, s, variant, stream)
                     : launch_cells_vec<float4, 4>(x, order, cell_start, ncells, out, lpr, rpi, s, variant, stream);
  }
  dim3 grid(cdiv(ncells, 4)), block(256);
  if (x_is_bf16)
    bev_pool_fwd_cells_scalar_kernel<true><<<grid, block, 0, stream>>>(x, order, cell_start, ncells, out, s);
  else
    bev_pool_fwd_cells_scalar_kernel<false><<<grid, block, 0, stream>>>(x, order, cell_start, ncells, out, s);
  BEVAMD_LAUNCH_CHECK("bev_pool_fwd_cells_scalar");
  return BEVAMD_OK;
}

#ifdef BEVAMD_PROFILING
/* host-only (tests of profiling builds): the XCD-striped walk's map of one padded line of workgroups — groups[p] = the 4-cell group workgroup p of
 * line `line` takes, or -1 for a padding slot; stripe_groups in {1, 2, 4}, rot_lines = lines per step of the rotating stripe map
 * (0: fixed).  Returns the padded line length (a multiple of 8: workgroup p runs on XCD p % 8), negative on bad arguments. */
int bevamd_bev_pool_striped_line(int row_groups, int stripe_groups, int line, int rot_lines, int* groups, int max_n) {
  if (!(row_groups > 0 && line >= 0 && rot_lines >= 0 && (stripe_groups == 1 || stripe_groups == 2 || stripe_groups == 4))) {
    set_error("bev_pool_striped_line: bad arguments");
    return -BEVAMD_ERR_INVALID_ARG;
  }
  const uint32_t rb = (uint32_t)row_groups, sw = (uint32_t)stripe_groups;
  const uint32_t pb = (uint32_t)cdiv(cdiv(rb, sw), 8) * 8u * sw;
  if (!groups || max_n < 0 || (uint32_t)max_n < pb) {
    set_error("bev_pool_striped_line: need room for %u entries", pb);
    return -BEVAMD_ERR_INVALID_ARG;
  }
  const uint32_t shift = rot_lines ? (uint32_t)line / (uint32_t)rot_lines : 0u;
  for (uint32_t p = 0; p < pb; ++p) {
    uint32_t j = 0;
    const bool live = sw == 1 ? striped_group<1>(p, (uint32_t)line, rb, shift, j)
                    : sw == 2 ? striped_group<2>(p, (uint32_t)line, rb, shift, j) : striped_group<4>(p, (uint32_t)line, rb, shift, j);
    groups[p] = live ? (int)j : -1;
  }
  return (int)pb;
}
#endif

int bevamd_bev_pool_forward_cells(const void* x, int x_is_bf16, const uint32_t* order, const uint32_t* cell_start,
                                  float* out, int n, int c, int b, int d, int h, int w, void* stream_) {
  return bevamd_bev_pool_forward_cells_tuned(x, x_is_bf16, order, cell_start, out, n, c, b, d, h, w, 0, stream_);
}

int bevamd_bev_pool_backward_points(const float* out_grad, const uint32_t* cell_of_point, float* x_grad, int n, int c, int b,
                                    int d, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_dims(n, c, b, d, h, w);
  if (rc) return rc;
  if (n == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(out_grad && cell_of_point && x_grad, "bev_pool_backward_points: null buffer");
  BEVAMD_REQUIRE(vec_path_ok(c, 4, out_grad, x_grad), "bev_pool_backward_points: c=%d must be a multiple of 4 with 16-byte aligned buffers (use bevamd_bev_pool_backward_rows otherwise)", c);
  const uint32_t ncells = (uint32_t)((unsigned long long)b * d * h * w);
  BevDims s{b, d, h, w, c};
  const int lpr = c / 4;
  const size_t total4 = (size_t)n * lpr;
  const size_t blocks = (total4 + 1023) / 1024;
  BEVAMD_REQUIRE(blocks < 0x7FFFFFFFull, "bev_pool_backward_points: too many rows");
  bev_pool_bwd_points_vec_kernel<<<dim3((unsigned)blocks), dim3(256), 0, stream>>>(out_grad, cell_of_point, ncells, total4, lpr,
                                                                                   (float4*)x_grad, s);
  BEVAMD_LAUNCH_CHECK("bev_pool_bwd_points");
  return BEVAMD_OK;
}

int bevamd_bev_pool_backward_rows(const float* out_grad, const uint32_t* order, const uint32_t* ranks_sorted,
                                  float* x_grad, int n, int c, int b, int d, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  int rc = check_dims(n, c, b, d, h, w);
  if (rc) return rc;
  if (n == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(out_grad && order && ranks_sorted && x_grad, "bev_pool_backward_rows: null buffer");
  const uint32_t ncells = (uint32_t)((unsigned long long)b * d * h * w);
  BevDims s{b, d, h, w, c};
  if (vec_path_ok(c, 4, out_grad, x_grad)) {
    int lpr = c / 4, rpi = 64 / lpr;
    dim3 grid(cdiv(cdiv(n, rpi), 4)), block(256);
    bev_pool_bwd_rows_vec_kernel<<<grid, block, 0, stream>>>(out_grad, order, ranks_sorted, ncells, n,
                                                             (float4*)x_grad, lpr, rpi, s);
  } else {
    dim3 grid(cdiv(n, 4)), block(256);
    bev_pool_bwd_rows_scalar_kernel<<<grid, block, 0, stream>>>(out_grad, order, ranks_sorted, ncells, n, x_grad, s);
  }
  BEVAMD_LAUNCH_CHECK("bev_pool_bwd_rows");
  return BEVAMD_OK;
}

}  // extern "C"
