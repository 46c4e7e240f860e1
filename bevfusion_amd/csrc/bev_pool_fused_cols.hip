// Fused depth (x) context -> BEV by image COLUMNS, for gfx950 (round 4).
//
// Same op as bev_pool_fused.hip —  out[cell, c] = sum over the frustum points p of the cell of depth[p] * ctx[pixel(p), c]
// (models/vtransforms/depth_lss.py:92-97 + base.py:141-176 without the [N', C] volume) — evaluated the other way round.
// The cell-centric kernel walks every cell's points through the sort permutation: one 4-byte depth gather (a 64-byte sector)
// and one 320-byte context row from L2 per POINT, 14.5 M of them per 8 frames: 2.85 GB of HBM-side traffic for 0.43 GB of
// algorithmic bytes (profiles/r03_pmc_infer_per_kernel.txt), 577 us.
//
// The BEV grid has ONE z cell, so the fH frustum points of an image column (camera, depth bin d, feature column w) — one ray
// direction in the ground plane, fH elevations — fall into one BEV cell, or a few when the camera is pitched / rolled or the
// image augmentation rotates: maximal runs of consecutive rows h with the same cell.  Per plan (static per calibration):
//   keep[col], end[col]   32-bit row masks of column col = (cam * D + d) * fW + w: row kept by the range mask / row closes a run
//   run_first[col]        number of runs before this column; slot_of_run[run] = position of the run among all runs sorted by
//                         (frame, cell) — stable, so the rows of a cell stay in (camera, d, w, h) order: a FIXED summation order
//   prow_start[cell]      CSR of the sorted runs over the frame-major cells
// Pass 1 (bev_fused_cols_kernel): a workgroup stages the context rows of 4 image columns x all fH rows (1280 contiguous bytes
// per row: coalesced, each context element read ONCE per depth half) and the depth distribution of those columns (range mask
// folded in as zeros) in LDS and forms  partial[run, :] = sum_{h in run} depth * ctx  in registers — a [D x fH] x [fH x C]
// product per column, 16 FMAs per two LDS reads — writing one 4*C-byte row per RUN (~20-30x fewer than points).
// Pass 2 (bev_fused_reduce_kernel): every BEV cell sums its consecutive partial rows and is stored once, empty cells as zeros.
// Algorithmic-side traffic per 8 flagship frames: depth 64 MB + context 2 x 43 MB + partial rows 2 x ~190 MB + output 332 MB.
//
// Summation order differs from the reference's (it adds a cell's points one by one in (camera, d, h, w) order; here rows of a
// run first, then runs): fp32 throughout, <= 1e-4 against float64 (tests/test_gpu_bev_pool.py), deterministic.
// Long-tailed plans (a camera rolled by 90 degrees: as many runs as points) keep the cell-centric kernel: the host decides from
// the run count.
#include <stdlib.h>

#include "common.h"

namespace bevamd {

struct ColDims {
  int BN, D, fH, fW, C;   // cameras of the whole batch, depth bins, feature rows / columns, channels
  int DH, ndh, nwb;       // depth bins per tile (multiple of 4), tiles along d, tiles along w (4 columns each)
};

constexpr int COL_WB = 4;            // image columns per tile: one float4 of depth along w, 4*C*4 contiguous context bytes per row
static_assert(COL_WB == 4, "the depth tile, the keep / end mask quads and the item map are written for 4 columns");
constexpr int COL_THREADS = 320;     // 16 items of 20 lanes at C = 80
constexpr int CTX_BATCH = 4;         // context float4 per thread and staging batch (2 and 8 measured: EXPERIMENTS C.8)
constexpr int COL_DEP_PITCH = COL_WB * 4 + 4;   // floats per (d-group, h) row of the depth tile: +4 keeps b128 writes conflict-free

// ---- plan ---------------------------------------------------------------------------------------------------------------
// one thread per image column: row masks + run count
__global__ __launch_bounds__(256) void bev_fused_col_masks_kernel(const uint32_t* __restrict__ cell_of_point, uint32_t ncells,
                                                                  int ncols, int fH, int fW, uint32_t* __restrict__ keep,
                                                                  uint32_t* __restrict__ endm, uint32_t* __restrict__ nruns) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= ncols) return;
  const int w = col % fW;
  const size_t base = (size_t)(col / fW) * fH * fW + w;    // (cam * D + d) * fH * fW + w
  uint32_t k = 0, e = 0;
  uint32_t cur = cell_of_point[base];
  for (int h = 0; h < fH; ++h) {
    const uint32_t nxt = h + 1 < fH ? cell_of_point[base + (size_t)(h + 1) * fW] : 0xFFFFFFFFu;
    if (cur < ncells) {
      k |= 1u << h;
      if (nxt != cur) e |= 1u << h;
    }
    cur = nxt;
  }
  keep[col] = k;
  endm[col] = e;
  nruns[col] = (uint32_t)__popc(e);
}

// frame-major cell key of a rank (rank = local * B + b, bev_pool.py:86-91)
__device__ __forceinline__ uint32_t frame_major(uint32_t rank, uint32_t B, uint32_t per_frame) {
  return (rank % B) * per_frame + rank / B;
}

// one thread per column: (key, run id) of its runs
__global__ __launch_bounds__(256) void bev_fused_run_keys_kernel(const uint32_t* __restrict__ cell_of_point,
                                                                 const uint32_t* __restrict__ endm,
                                                                 const uint32_t* __restrict__ run_first, int ncols, int fH, int fW,
                                                                 uint32_t B, uint32_t per_frame, uint32_t* __restrict__ keys,
                                                                 uint32_t* __restrict__ vals) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= ncols) return;
  uint32_t e = endm[col];
  if (!e) return;
  const int w = col % fW;
  const size_t base = (size_t)(col / fW) * fH * fW + w;
  uint32_t run = run_first[col];
  while (e) {
    const int h = __ffs((int)e) - 1;
    e &= e - 1;
    keys[run] = frame_major(cell_of_point[base + (size_t)h * fW], B, per_frame);
    vals[run] = run;
    ++run;
  }
}

__global__ __launch_bounds__(256) void bev_fused_slot_of_run_kernel(const uint32_t* __restrict__ perm, uint32_t nruns,
                                                                    uint32_t* __restrict__ slot_of_run) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < nruns) slot_of_run[perm[i]] = i;
}

// prow_start[c] = lower_bound(sorted keys, c), c in [0, ncells]
__global__ __launch_bounds__(256) void bev_fused_prow_start_kernel(const uint32_t* __restrict__ keys, uint32_t nruns,
                                                                   uint32_t ncells, uint32_t* __restrict__ prow_start) {
  const uint32_t c = blockIdx.x * 256u + threadIdx.x;
  if (c > ncells) return;
  uint32_t lo = 0, hi = nruns;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (keys[mid] < c) lo = mid + 1; else hi = mid;
  }
  prow_start[c] = lo;
}

// Round 5: a run that is ALONE in its cell needs no second pass — its partial row IS the cell (x + 0 is exact, so the bits are the
// ones pass 2 would have stored).  On the benchmark rig 453 888 runs fall into 363 752 cells: four cells of five hold one run.
// Such a run's entry in slot_of_run becomes (1 << 31 | output row of its cell); pass 1 stores it straight into `out`, pass 2
// skips one-run cells: 145 MB of partial rows written and read back per 8 frames shrink to the multi-run cells' share.
constexpr uint32_t RUN_DIRECT = 0x80000000u;
__global__ __launch_bounds__(256) void bev_fused_mark_single_kernel(const uint32_t* __restrict__ sorted_keys,
                                                                    const uint32_t* __restrict__ prow_start, uint32_t nruns,
                                                                    uint32_t per_frame, int D, int H, int W,
                                                                    uint32_t* __restrict__ slot_of_run) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= nruns) return;
  const uint32_t slot = slot_of_run[i];
  const uint32_t cell = sorted_keys[slot];                      // frame-major: b * per_frame + (x * W + y) * D + z
  if (prow_start[cell + 1] - prow_start[cell] != 1u) return;
  const uint32_t b = cell / per_frame;
  uint32_t local = cell - b * per_frame;
  const uint32_t gz = local % (uint32_t)D; local /= (uint32_t)D;
  const uint32_t gy = local % (uint32_t)W, gx = local / (uint32_t)W;
  slot_of_run[i] = RUN_DIRECT | (((b * (uint32_t)D + gz) * (uint32_t)H + gx) * (uint32_t)W + gy);   // row of out [b, z, x, y, :]
}

// ---- pass 1 ---------------------------------------------------------------------------------------------------------------
struct alignas(16) CU4 { uint32_t x, y, z, w; };

__device__ __forceinline__ void fma4(float4& a, float d, const float4& c) {
  a.x = fmaf(d, c.x, a.x); a.y = fmaf(d, c.y, a.y); a.z = fmaf(d, c.z, a.z); a.w = fmaf(d, c.w, a.w);
}

template <bool CTX_BF16>
__global__ __launch_bounds__(COL_THREADS) void bev_fused_cols_kernel(
    const float* __restrict__ depth, const void* __restrict__ ctx_, const uint32_t* __restrict__ keep,
    const uint32_t* __restrict__ endm, const uint32_t* __restrict__ run_first, const uint32_t* __restrict__ slot_of_run,
    float* __restrict__ partial, float* __restrict__ out, ColDims s) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int C = s.C, fH = s.fH, lpr = C >> 2;
  float* s_ctx = lds;                                        // [fH][4][C]
  float* s_dep = lds + (size_t)fH * COL_WB * C;              // [DH / 4][fH][COL_DEP_PITCH]: (w, d % 4) -> w * 4 + d % 4
  // tile: XCD x = blockIdx.x % 8 owns a contiguous eighth of the tiles, i.e. whole cameras — the depth lines a 4-column tile
  // touches (16 of every 352 bytes) are shared with the neighbouring column tiles, which then run on the same L2
  const int tiles_per_cam = s.nwb * s.ndh;
  const int total = s.BN * tiles_per_cam, per = (total + 7) >> 3;
  const int t = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (((int)blockIdx.x >> 3) >= per || t >= total) return;
  const int bn = t / tiles_per_cam, rem = t - bn * tiles_per_cam;
  const int wb = rem / s.ndh, dh = rem - wb * s.ndh;
  const int w0 = wb * COL_WB, d0 = dh * s.DH;
  const int tid = threadIdx.x;

  // Run metadata of the tile's DH x 4 image columns -> LDS (round 5).  The item loop used to fetch end mask and first run of its
  // four columns from global memory at the top of every pass and the run's output slot when the run closed: two dependent round
  // trips per pass with nothing else to issue.  Here one thread per column asks for mask + first run BEFORE the context rows go
  // out, for the slot of the column's first run between context and depth staging (its address has arrived by then), and the
  // item loop reads all three from LDS; only columns with several runs (pitched / rolled rigs) look further slots up.
  const int n_cols = s.DH * COL_WB;
  // [n_cols] end masks, [n_cols] slot of the first run, [DH] first run of the bin's column w0 (the other three follow from the
  // popcounts of the masks before them: run_first is the exclusive scan of the run counts in column order) — 2 160 bytes at
  // DH = 60: with the context and depth tiles 81 520, two workgroups per CU (a third array of n_cols words was one too many:
  // 164 480 bytes for two, ONE workgroup per CU, measured 8 % slower than before the change instead of faster)
  uint32_t* s_meta = (uint32_t*)(s_dep + (size_t)(s.DH >> 2) * fH * COL_DEP_PITCH);
  uint32_t m_end[2], m_rf[2];   // n_cols <= 2 * COL_THREADS (DH <= 128); clamped addresses, no branches: nothing waits here
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = tid + u * COL_THREADS;
    const int d = d0 + i / COL_WB;
    const size_t col = ((size_t)bn * s.D + (d < s.D ? d : s.D - 1)) * s.fW + w0 + (i & (COL_WB - 1));
    m_end[u] = endm[col];
    m_rf[u] = run_first[col];
  }

  // Staging: every global load of a batch is issued before the first LDS write (indices clamped instead of branched around, so
  // that hipcc keeps the loads of a batch in flight together: branches made it wait for each load in turn).  Batches of 4
  // context vectors / one depth pair (8 loads) per thread — issuing the WHOLE tile's 28 loads per thread up front (one exposed
  // round trip instead of four) was measured 27-46 us per 8 frames SLOWER (round 5, EXPERIMENTS C.8).
  // context rows: fH rows of 4 * C contiguous floats
  {
    const int per_row = COL_WB * lpr;                        // float4 per row
    const int nvec = fH * per_row;
    for (int i0 = 0; i0 < nvec; i0 += CTX_BATCH * COL_THREADS) {
      float4 v[CTX_BATCH];
#pragma unroll
      for (int u = 0; u < CTX_BATCH; ++u) {
        int i = i0 + u * COL_THREADS + tid;
        i = i < nvec ? i : nvec - 1;
        const int h = i / per_row, j = i - h * per_row;
        const size_t row = ((size_t)bn * fH + h) * s.fW + w0;   // context pixel row (cam, h, w0)
        if constexpr (CTX_BF16) {
          const uint2 q = ((const uint2*)ctx_)[row * lpr + j];  // 4 channels = 8 bytes of bf16
          v[u] = make_float4(__uint_as_float(q.x << 16), __uint_as_float(q.x & 0xFFFF0000u), __uint_as_float(q.y << 16),
                             __uint_as_float(q.y & 0xFFFF0000u));
        } else {
          v[u] = ((const float4*)ctx_)[row * lpr + j];
        }
      }
      // (opaque uses: without them hipcc sinks each load into the guarded store below and waits for them one by one)
#pragma unroll
      for (int u = 0; u < CTX_BATCH; ++u) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
      for (int u = 0; u < CTX_BATCH; ++u) {
        const int i = i0 + u * COL_THREADS + tid;
        if (i < nvec) ((float4*)s_ctx)[i] = v[u];
      }
    }
  }
  uint32_t m_slot[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = tid + u * COL_THREADS;
    if (!(i < n_cols && d0 + i / COL_WB < s.D)) m_end[u] = 0u;     // past the tile / past the last depth bin: no runs
    m_slot[u] = slot_of_run[m_end[u] ? m_rf[u] : 0u];             // (run 0 exists: the kernel is not launched without runs)
  }
  // depth tile with the range mask folded in: thread -> (d-group of 4 bins, h); 4 float4 loads along w, 4 float4 stores along d
  {
    const int ndg = s.DH >> 2, npair = ndg * fH;
    for (int i0 = 0; i0 < npair; i0 += COL_THREADS) {
      int i = i0 + tid;
      const bool live = i < npair;
      i = live ? i : npair - 1;
      const int dg = i / fH, h = i - dg * fH;
      float4 v[4];
      CU4 k[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int d = d0 + dg * 4 + j;
        const size_t cam_d = (size_t)bn * s.D + (d < s.D ? d : s.D - 1);
        k[j] = *(const CU4*)(keep + cam_d * s.fW + w0);
        v[j] = *(const float4*)(depth + (cam_d * fH + h) * s.fW + w0);
      }
      float m[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool in = d0 + dg * 4 + j < s.D;
        m[j][0] = in && ((k[j].x >> h) & 1u) ? v[j].x : 0.f;
        m[j][1] = in && ((k[j].y >> h) & 1u) ? v[j].y : 0.f;
        m[j][2] = in && ((k[j].z >> h) & 1u) ? v[j].z : 0.f;
        m[j][3] = in && ((k[j].w >> h) & 1u) ? v[j].w : 0.f;
      }
      if (live) {
        float4* dst = (float4*)(s_dep + ((size_t)dg * fH + h) * COL_DEP_PITCH);
#pragma unroll
        for (int wl = 0; wl < 4; ++wl) dst[wl] = make_float4(m[0][wl], m[1][wl], m[2][wl], m[3][wl]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = tid + u * COL_THREADS;
    if (i < n_cols) {
      s_meta[i] = m_end[u];
      s_meta[n_cols + i] = m_slot[u];
      if ((i & (COL_WB - 1)) == 0) s_meta[2 * n_cols + i / COL_WB] = m_rf[u];
    }
  }
  __syncthreads();

  // items: (column wl, d-group dg) x lpr lanes; a lane owns 4 channels of the 4 depth bins of its item
  const int items_per_pass = COL_THREADS / lpr;
  const int item0 = tid / lpr, cv = tid - item0 * lpr;
  if (item0 >= items_per_pass) return;
  const int n_items = COL_WB * (s.DH >> 2);
  for (int item = item0; item < n_items; item += items_per_pass) {
    const int wl = item & (COL_WB - 1), dg = item >> 2;
    uint32_t en[4], rf[4], sl[4], eany = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int dl = dg * 4 + j;                      // (bins past D hold zeros)
      const CU4 em = ((const CU4*)s_meta)[dl];        // the end masks of the bin's four columns
      en[j] = wl == 0 ? em.x : wl == 1 ? em.y : wl == 2 ? em.z : em.w;
      rf[j] = s_meta[2 * n_cols + dl] + (wl > 0 ? __popc(em.x) : 0) + (wl > 1 ? __popc(em.y) : 0) + (wl > 2 ? __popc(em.z) : 0);
      sl[j] = s_meta[n_cols + dl * COL_WB + wl];
      eany |= en[j];
    }
    if (!eany) continue;
    uint32_t later = 0;      // bit j: bin j has closed a run already
    float4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4* cx = (const float4*)s_ctx + (size_t)wl * lpr + cv;           // + h * 4 * lpr
    const float4* dp = (const float4*)(s_dep + (size_t)dg * fH * COL_DEP_PITCH) + wl;   // + h * (COL_DEP_PITCH / 4)
    int h = 0;
    while (eany) {
      const int e = __ffs((int)eany) - 1;      // next row that closes a run of one of the 4 bins; rows after the last one hold no kept point
      eany &= eany - 1;
      // rows h .. e: four at a time with their eight LDS reads issued together (one exposed LDS round trip per four rows)
      for (; h + 3 <= e; h += 4) {
        float4 c[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          c[u] = cx[(size_t)(h + u) * COL_WB * lpr];
          dv[u] = dp[(size_t)(h + u) * (COL_DEP_PITCH / 4)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          fma4(acc[0], dv[u].x, c[u]); fma4(acc[1], dv[u].y, c[u]); fma4(acc[2], dv[u].z, c[u]); fma4(acc[3], dv[u].w, c[u]);
        }
      }
      for (; h <= e; ++h) {
        const float4 c = cx[(size_t)h * COL_WB * lpr];
        const float4 dv = dp[(size_t)h * (COL_DEP_PITCH / 4)];
        fma4(acc[0], dv.x, c); fma4(acc[1], dv.y, c); fma4(acc[2], dv.z, c); fma4(acc[3], dv.w, c);
      }
      // the bins that close a run at row e.  A column's first run has its slot in LDS; a later one (pitched / rolled rigs) is
      // looked up here.  All lookups first, then all stores: a lookup in front of each store made every store wait for the one
      // before it, and a lookup left pending around the loop makes hipcc wait for ALL memory traffic at the top of every trip.
      uint32_t slot[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        slot[j] = sl[j];
        if (((en[j] >> e) & 1u) && ((later >> j) & 1u)) slot[j] = slot_of_run[rf[j]];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(slot[j]));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if ((en[j] >> e) & 1u) {
          later |= 1u << j;
          ++rf[j];
          float4* dst = (slot[j] & RUN_DIRECT) ? (float4*)out + (size_t)(slot[j] & ~RUN_DIRECT) * lpr
                                               : (float4*)partial + (size_t)slot[j] * lpr;
          dst[cv] = acc[j];   // (alone in its cell: straight into the output row)
          acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  }
}

// ---- pass 2 ---------------------------------------------------------------------------------------------------------------
// lpr lanes per cell, rpi cells per wave instruction, G cell groups per wave in flight (two thirds of the cells are empty and the
// rest hold one or two rows: the kernel is a chain of short dependent loads — CSR bounds, rows, store — so each lane keeps G
// independent chains going: 1 -> 2 groups 138 -> 111 us in round 4; 4 groups measured in round 5: 90.5 vs 88.5 us, no gain, so
// G = 2); a cell's partial rows are consecutive and are added in a fixed order
__device__ __forceinline__ float4 reduce_rows(const float4* __restrict__ p, int len, int lpr) {
  // four interleaved partial sums (row r goes to sum r % 4), folded at the end: a fixed order, and a quarter of the rounding
  // growth of one long chain on the rare cells with hundreds of rows
  float4 a4[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) a4[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < len; r += 4) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = r + u < len ? p[(size_t)(r + u) * lpr] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 4; ++u) { a4[u].x += v[u].x; a4[u].y += v[u].y; a4[u].z += v[u].z; a4[u].w += v[u].w; }
  }
  float4 acc;
  acc.x = (a4[0].x + a4[1].x) + (a4[2].x + a4[3].x);
  acc.y = (a4[0].y + a4[1].y) + (a4[2].y + a4[3].y);
  acc.z = (a4[0].z + a4[1].z) + (a4[2].z + a4[3].z);
  acc.w = (a4[0].w + a4[1].w) + (a4[2].w + a4[3].w);
  return acc;
}

template <int G>
__global__ __launch_bounds__(256) void bev_fused_reduce_kernel(const float4* __restrict__ partial,
                                                               const uint32_t* __restrict__ prow_start, uint32_t ncells,
                                                               float* __restrict__ out, int lpr, int rpi, int B, int D, int H, int W,
                                                               int C, int direct) {
  const int lane = threadIdx.x & 63;
  const int slot = lane / lpr, cv = lane - slot * lpr;
  if (slot >= rpi) return;
  const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6);
  // frame-major cells: b * (D*H*W) + (x * W + y) * D + z; this wave owns G * rpi consecutive cells, G per lane group
  uint32_t cell[G], st[G];
  int len[G];
  bool ok[G], small = true;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    cell[g] = wave * (uint32_t)(G * rpi) + (uint32_t)(g * rpi + slot);
    ok[g] = cell[g] < ncells;
    const uint32_t cc = ok[g] ? cell[g] : 0u;
    st[g] = prow_start[cc];
    len[g] = ok[g] ? (int)(prow_start[cc + 1] - st[g]) : 0;
    small = small && len[g] <= 2;
  }
  if (!ok[0]) return;
  float4 acc[G];
  if (small) {   // the common case: every cell's rows requested together
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a[G], b[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4* p = partial + (size_t)st[g] * lpr + cv;
      // (a one-run cell's row went straight to `out` in pass 1: its slot of `partial` was never written and is not read)
      a[g] = len[g] > (direct ? 1 : 0) ? p[0] : z;
      b[g] = len[g] > 1 ? p[lpr] : z;
    }
    // same association as reduce_rows for <= 2 rows: (a + b) + (0 + 0)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = make_float4(a[g].x + b[g].x, a[g].y + b[g].y, a[g].z + b[g].z, a[g].w + b[g].w);
  } else {
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = reduce_rows(partial + (size_t)st[g] * lpr + cv, len[g], lpr);
  }
  const uint32_t per_frame = ncells / (uint32_t)B;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    // a cell with exactly one run was stored by pass 1 (RUN_DIRECT): nothing to do here (direct == 0: plans built without the marks)
    if (!ok[g] || (direct && len[g] == 1)) continue;
    const uint32_t b = cell[g] / per_frame;
    uint32_t local = cell[g] - b * per_frame;
    const uint32_t gz = local % (uint32_t)D; local /= (uint32_t)D;
    const uint32_t gy = local % (uint32_t)W;
    const uint32_t gx = local / (uint32_t)W;
    // out[b, z, x, y, :] (bev_pool_cuda.cu:33-35)
    *((float4*)(out + ((((size_t)b * D + gz) * H + gx) * W + gy) * (size_t)C) + cv) = acc[g];
  }
}

// ---- backward by columns ---------------------------------------------------------------------------------------------------
//   d_depth[p]      = < g[cell(p), :], ctx[pixel(p), :] >            (0 for points the range mask dropped)
//   d_ctx[pixel, :] = sum over the depth bins d of the pixel of  depth[p] * g[cell(p), :]
// The point-wise kernels of bev_pool_fused.hip fetch one 4*C-byte gradient row and one context row from L2 per POINT (882 + 357 us
// per 4 flagship frames).  A workgroup here owns ONE image column (camera, w): the gradient rows of its runs — one row per (depth
// bin, run), ~D of them — its fH context rows and its masked depth values sit in LDS, both products are formed from there, and
// every global byte is read or written once: d_depth leaves as [camera][w][d][h] (contiguous per column; the host hands it on as
// a permuted view), d_ctx as the column's fH rows.
constexpr int BWD_RCAP = 64;       // gradient rows resident at a time (a column with more runs takes several passes; 64 rows keep two workgroups per CU)
constexpr int BWD_THREADS = 256;

__global__ __launch_bounds__(BWD_THREADS) void bev_fused_bwd_cols_kernel(
    const float* __restrict__ out_grad, const float* __restrict__ depth, const float* __restrict__ ctx,
    const uint32_t* __restrict__ keep, const uint32_t* __restrict__ endm, const uint32_t* __restrict__ cell_of_point,
    float* __restrict__ d_depth_t /*[BN][fW][D][fH]*/, float* __restrict__ d_ctx, int BN, int D, int fH, int fW, int C, int B, int Dz,
    int H, int W) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lpr = C >> 2;
  float* s_cx = lds;                                   // [fH][C]
  float* s_g = s_cx + (size_t)fH * C;                  // [BWD_RCAP][C]
  float* s_dep = s_g + (size_t)BWD_RCAP * C;           // [D][fH] masked depth
  uint32_t* s_row = (uint32_t*)(s_dep + (size_t)D * fH);   // [D * fH]: gradient row (cell) index of every run, by local run id
  int* s_pre = (int*)(s_row + (size_t)D * fH);         // [D + 1] runs before depth bin d
  uint16_t* s_ridx = (uint16_t*)(s_pre + D + 1);       // [D][fH] local run id of the point, 0xFFFF = dropped
  const int tid = threadIdx.x;
  // column: XCD x takes a contiguous eighth of the (camera, w) columns — whole cameras, whose cell gradients it then keeps in its L2
  const int total = BN * fW, per = (total + 7) >> 3;
  const int t = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  if (((int)blockIdx.x >> 3) >= per || t >= total) return;
  const int bn = t / fW, w = t - bn * fW;

  for (int i = tid; i < fH * lpr; i += BWD_THREADS) {
    const int h = i / lpr, j = i - h * lpr;
    ((float4*)s_cx)[i] = ((const float4*)ctx)[(((size_t)bn * fH + h) * fW + w) * lpr + j];
  }
  for (int d = tid; d < D; d += BWD_THREADS) s_pre[d + 1] = __popc(endm[((size_t)bn * D + d) * fW + w]);
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int d = 0; d < D; ++d) { const int c = s_pre[d + 1]; s_pre[d] = run; run += c; }
    s_pre[D] = run;
  }
  __syncthreads();
  const int nrun = s_pre[D];
  const uint32_t ncells = (uint32_t)B * Dz * H * W;
  for (int d = tid; d < D; d += BWD_THREADS) {
    const size_t col = ((size_t)bn * D + d) * fW + w;
    const uint32_t k = keep[col], e = endm[col];
    int run = s_pre[d];
    for (int h = 0; h < fH; ++h) {
      s_ridx[d * fH + h] = (k >> h) & 1u ? (uint16_t)run : (uint16_t)0xFFFFu;
      if ((e >> h) & 1u) {
        uint32_t r = cell_of_point[(((size_t)bn * D + d) * fH + h) * fW + w];   // rank = ((x * W + y) * Dz + z) * B + b
        r = r < ncells ? r : 0u;
        const uint32_t gb = r % (uint32_t)B; r /= (uint32_t)B;
        const uint32_t gz = r % (uint32_t)Dz; r /= (uint32_t)Dz;
        const uint32_t gy = r % (uint32_t)W;
        const uint32_t gx = r / (uint32_t)W;
        s_row[run] = ((gb * (uint32_t)Dz + gz) * (uint32_t)H + gx) * (uint32_t)W + gy;   // row of out_grad [B, Dz, H, W, C]
        ++run;
      }
    }
  }
  for (int i0 = 0; i0 < D * fH; i0 += 4 * BWD_THREADS) {   // masked depth: four strided loads in flight per lane
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      int i = i0 + u * BWD_THREADS + tid;
      i = i < D * fH ? i : D * fH - 1;
      v[u] = depth[((size_t)bn * D * fH + i) * fW + w];
    }
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * BWD_THREADS + tid;
      if (i < D * fH) s_dep[i] = v[u];
    }
  }
  __syncthreads();

  const int n_hc = fH * lpr;                           // (h, channel group) items of d_ctx
  float4 acc[4];                                       // items tid, tid + 256, ... (<= 4: fH * C / 4 <= 1024)
#pragma unroll
  for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  float* dd = d_depth_t + ((size_t)bn * fW + w) * D * fH;
  for (int c0 = 0; c0 < nrun || c0 == 0; c0 += BWD_RCAP) {
    const int nr = nrun - c0 < BWD_RCAP ? nrun - c0 : BWD_RCAP;
    for (int i = tid; i < nr * lpr; i += BWD_THREADS) {
      const int r = i / lpr, j = i - r * lpr;
      ((float4*)s_g)[i] = ((const float4*)out_grad)[(size_t)s_row[c0 + r] * lpr + j];
    }
    __syncthreads();
    // Round 6: the depth bins this pass can touch.  Runs are numbered in depth-bin order (s_pre), so the resident rows [c0, c0 + nr)
    // belong to ONE contiguous range of bins: from the bin holding run c0 (the first pass starts at bin 0) to the bin holding the
    // next pass's first run (inclusive: it may straddle; the last pass ends at D - 1) — bins without a run in between are covered
    // too, for their zeros.  Every pass used to scan all D * fH points and all D bins: an augmented training rig has ~3 runs per
    // (depth bin, column), six passes a column, and the kernel took 1.34 ms against 0.46 on the test-time rig (one to two passes).
    // The non-zero terms of every sum are formed in the same order as before: same bits.
    int d_lo = 0, d_hi = D - 1;
    if (c0 > 0) {                       // first d with s_pre[d + 1] > c0
      int lo = 0, hi = D - 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_pre[mid + 1] > c0) hi = mid; else lo = mid + 1; }
      d_lo = lo;
    }
    if (c0 + nr < nrun) {               // first d with s_pre[d + 1] > c0 + nr
      int lo = 0, hi = D - 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_pre[mid + 1] > c0 + nr) hi = mid; else lo = mid + 1; }
      d_hi = lo;
    }
    // d_depth of the points whose run is resident (dropped points: zero, written by the pass — or the two — that cover their bin)
    for (int p = d_lo * fH + tid; p < (d_hi + 1) * fH; p += BWD_THREADS) {
      const int r = (int)s_ridx[p] - c0;
      if (s_ridx[p] == 0xFFFFu) {
        dd[p] = 0.f;
      } else if (r >= 0 && r < nr) {
        const int h = p % fH;
        const float4* g = (const float4*)s_g + (size_t)r * lpr;
        const float4* c = (const float4*)s_cx + (size_t)h * lpr;
        float a0 = 0.f, a1 = 0.f;
        int j = 0;
        for (; j + 4 <= lpr; j += 4) {   // eight LDS reads in flight per trip
          float4 gv[4], cv[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) { gv[u] = g[j + u]; cv[u] = c[j + u]; }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            a0 = fmaf(gv[u].x, cv[u].x, a0); a1 = fmaf(gv[u].y, cv[u].y, a1);
            a0 = fmaf(gv[u].z, cv[u].z, a0); a1 = fmaf(gv[u].w, cv[u].w, a1);
          }
        }
        for (; j < lpr; ++j) {
          const float4 gv = g[j], cv = c[j];
          a0 = fmaf(gv.x, cv.x, a0); a1 = fmaf(gv.y, cv.y, a1); a0 = fmaf(gv.z, cv.z, a0); a1 = fmaf(gv.w, cv.w, a1);
        }
        dd[p] = a0 + a1;
      }
    }
    // d_ctx: (h, channel group) items, depth bins in order
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int it = tid + u * BWD_THREADS;
      if (it < n_hc) {
        const int h = it / lpr, j = it - h * lpr;
        // four depth bins per trip, branch-free: a bin whose run is not resident (or dropped, or past the end) reads row 0 with
        // weight 0 — its loads are issued with the others instead of behind a branch
        for (int d0 = d_lo; d0 <= d_hi; d0 += 4) {
          int r[4];
          float wgt[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int d = d0 + q <= d_hi ? d0 + q : d_hi;
            r[q] = (int)s_ridx[d * fH + h] - c0;
            wgt[q] = s_dep[d * fH + h];
          }
          float4 gv[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool ok = d0 + q <= d_hi && r[q] >= 0 && r[q] < nr;
            wgt[q] = ok ? wgt[q] : 0.f;
            const float4 ld = ((const float4*)s_g)[(size_t)(ok ? r[q] : 0) * lpr + j];
            // SELECT, not weight 0 alone: a column without a kept point (nr == 0) never loads s_g, and 0 * (stale NaN / Inf) = NaN
            gv[q] = ok ? ld : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[u].x = fmaf(wgt[q], gv[q].x, acc[u].x); acc[u].y = fmaf(wgt[q], gv[q].y, acc[u].y);
            acc[u].z = fmaf(wgt[q], gv[q].z, acc[u].z); acc[u].w = fmaf(wgt[q], gv[q].w, acc[u].w);
          }
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int it = tid + u * BWD_THREADS;
    if (it < n_hc) {
      const int h = it / lpr, j = it - h * lpr;
      ((float4*)d_ctx)[(((size_t)bn * fH + h) * fW + w) * lpr + j] = acc[u];
    }
  }
}

static size_t bwd_cols_lds(int c, int depth_bins, int fh) {
  size_t b = ((size_t)fh * c + (size_t)BWD_RCAP * c + (size_t)depth_bins * fh) * sizeof(float);   // s_cx, s_g, s_dep
  b += (size_t)depth_bins * fh * sizeof(uint32_t) + (size_t)(depth_bins + 1) * sizeof(int);         // s_row, s_pre
  b += (size_t)depth_bins * fh * sizeof(uint16_t);                                                 // s_ridx
  return (b + 15) / 16 * 16;
}

static int cols_shape(int c, int depth_bins, int fh, int fw, ColDims& s, size_t& lds_bytes) {
  if (c <= 0 || (c & 3) || (c >> 2) > 64 || (c >> 2) > COL_THREADS || fh <= 0 || fh > 32 || fw <= 0 || (fw % COL_WB) || depth_bins <= 0) return 0;
  s.D = depth_bins; s.fH = fh; s.fW = fw; s.C = c;
  const int dpad = (depth_bins + 3) / 4 * 4;
  static int dh_max = 0;   // depth bins per tile (tuning: BEVAMD_FUSED_COLS_DH, a multiple of 4; 60 = two tiles for the 118 bins)
  if (dh_max == 0) {
    const char* e = getenv("BEVAMD_FUSED_COLS_DH");
    dh_max = e ? atoi(e) / 4 * 4 : 60;
    if (dh_max < 4 || dh_max > 128) dh_max = 60;
  }
  s.DH = dpad < dh_max ? dpad : dh_max;
  s.ndh = (depth_bins + s.DH - 1) / s.DH;
  s.nwb = fw / COL_WB;
  lds_bytes = ((size_t)fh * COL_WB * c + (size_t)(s.DH / 4) * fh * COL_DEP_PITCH + (size_t)(2 * COL_WB + 1) * s.DH) * sizeof(float);
  return lds_bytes <= 150 * 1024;
}

}  // namespace bevamd

using namespace bevamd;

extern "C" {

/* 1 if the column formulation of the fused pooling supports this shape (c % 4 == 0, fh <= 32, fw % 4 == 0, LDS tile fits);
 * otherwise the caller stays on bevamd_bev_pool_fused_forward[_scheduled]. */
int bevamd_bev_pool_fused_columns_supported(int c, int depth_bins, int fh, int fw) {
  ColDims s;
  size_t lds;
  return cols_shape(c, depth_bins, fh, fw, s, lds);
}

/* 1 if bevamd_bev_pool_fused_backward_columns takes this shape (its own limits: the (h, channel group) items of d_ctx fit four
 * per lane, fewer than 65535 runs per column, the column's rows fit LDS); otherwise the caller keeps the point-wise backward. */
int bevamd_bev_pool_fused_backward_columns_supported(int c, int depth_bins, int fh, int fw) {
  if (c <= 0 || (c & 3) || fh <= 0 || fh > 32 || fw <= 0 || depth_bins <= 0) return 0;
  if (fh * (c / 4) > 4 * BWD_THREADS || (long long)depth_bins * fh >= 65535) return 0;
  return bwd_cols_lds(c, depth_bins, fh) <= 150 * 1024;
}

/* Column plan, step 1 (static per plan): row masks keep / end [cams * depth_bins * fw] and run_first (exclusive scan of the
 * per-column run counts) from cell_of_point [n] (bevamd_bev_pool_cell_of_point), n = cams * depth_bins * fh * fw; total_runs
 * (device uint32) receives the number of runs.  The caller reads it back once (plan time, not per frame) to size step 2.
 * ws: bevamd_bev_pool_fused_columns_workspace_bytes(n / fh, 0). */
size_t bevamd_bev_pool_fused_columns_workspace_bytes(int ncols, int nruns) {
  if (ncols < 0 || nruns < 0) return 0;
  const size_t a = scan_workspace_bytes((size_t)(ncols > 0 ? ncols : 1));
  const size_t r = (size_t)(nruns > 0 ? nruns : 1);
  return align_up(a, 256) + 4 * align_up(r * 4, 256) + align_up(radix_sort_workspace_bytes(r), 256);
}

int bevamd_bev_pool_fused_columns_count(const uint32_t* cell_of_point, int n, int depth_bins, int fh, int fw, int b, int d, int h,
                                        int w, uint32_t* keep, uint32_t* endm, uint32_t* run_first, uint32_t* total_runs,
                                        void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(n > 0 && depth_bins > 0 && fh > 0 && fh <= 32 && fw > 0 && b > 0 && d > 0 && h > 0 && w > 0,
                 "bev_pool_fused_columns_count: bad sizes (fh <= 32)");
  const long long per_cam = (long long)depth_bins * fh * fw;
  BEVAMD_REQUIRE(n % per_cam == 0, "bev_pool_fused_columns_count: n=%d is not a multiple of depth_bins*fh*fw=%lld", n, per_cam);
  BEVAMD_REQUIRE((unsigned long long)b * d * h * w < 0xFFFFFFF0ull, "bev_pool_fused_columns_count: b*d*h*w too large");
  BEVAMD_REQUIRE(cell_of_point && keep && endm && run_first && total_runs, "bev_pool_fused_columns_count: null buffer");
  const int ncols = n / fh;
  if (!ws || ws_bytes < scan_workspace_bytes((size_t)ncols)) {
    set_error("bev_pool_fused_columns_count: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  const uint32_t ncells = (uint32_t)((unsigned long long)b * d * h * w);
  bev_fused_col_masks_kernel<<<dim3(cdiv(ncols, 256)), dim3(256), 0, stream>>>(cell_of_point, ncells, ncols, fh, fw, keep, endm,
                                                                               run_first);
  BEVAMD_LAUNCH_CHECK("bev_fused_col_masks");
  return exclusive_scan_u32(run_first, run_first, (size_t)ncols, total_runs, ws, ws_bytes, stream);
}

/* Column plan, step 2: slot_of_run [nruns] (position of every run in the (frame, cell)-sorted order; stable: runs of a cell
 * stay in (camera, d, w, h) order; round 5: a run that is ALONE in its cell carries 1 << 31 | row of `out` instead, and the forward
 * stores it straight into the output) and prow_start [b*d*h*w + 1] (CSR of the sorted runs over FRAME-MAJOR cells:
 * cell = frame * d*h*w + (x * w + y) * d + z).  nruns = the value step 1 left in total_runs.
 * ws: bevamd_bev_pool_fused_columns_workspace_bytes(n / fh, nruns). */
int bevamd_bev_pool_fused_columns_build(const uint32_t* cell_of_point, const uint32_t* endm, const uint32_t* run_first, int n,
                                        int nruns, int depth_bins, int fh, int fw, int b, int d, int h, int w,
                                        uint32_t* slot_of_run, uint32_t* prow_start, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(n > 0 && nruns >= 0 && depth_bins > 0 && fh > 0 && fh <= 32 && fw > 0 && b > 0 && d > 0 && h > 0 && w > 0,
                 "bev_pool_fused_columns_build: bad sizes");
  BEVAMD_REQUIRE(n % ((long long)depth_bins * fh * fw) == 0, "bev_pool_fused_columns_build: n is not a multiple of depth_bins*fh*fw");
  BEVAMD_REQUIRE((unsigned long long)b * d * h * w < 0xFFFFFFF0ull, "bev_pool_fused_columns_build: b*d*h*w too large");
  BEVAMD_REQUIRE(cell_of_point && endm && run_first && prow_start && (nruns == 0 || slot_of_run), "bev_pool_fused_columns_build: null buffer");
  const int ncols = n / fh;
  const uint32_t ncells = (uint32_t)((unsigned long long)b * d * h * w);
  if (!ws || ws_bytes < bevamd_bev_pool_fused_columns_workspace_bytes(ncols, nruns)) {
    set_error("bev_pool_fused_columns_build: workspace too small");
    return BEVAMD_ERR_WORKSPACE;
  }
  Carver cv(ws, ws_bytes);
  cv.take<char>(align_up(scan_workspace_bytes((size_t)ncols), 256));
  const size_t r = (size_t)(nruns > 0 ? nruns : 1);
  uint32_t* keys_a = cv.take<uint32_t>(r);
  uint32_t* vals_a = cv.take<uint32_t>(r);
  uint32_t* keys_b = cv.take<uint32_t>(r);
  uint32_t* vals_b = cv.take<uint32_t>(r);
  void* sws = cv.base + cv.off;
  const size_t sws_bytes = ws_bytes - cv.off;
  const uint32_t* sorted_keys = keys_b;
  if (nruns > 0) {
    bev_fused_run_keys_kernel<<<dim3(cdiv(ncols, 256)), dim3(256), 0, stream>>>(cell_of_point, endm, run_first, ncols, fh, fw,
                                                                                (uint32_t)b, ncells / (uint32_t)b, keys_a, vals_a);
    BEVAMD_LAUNCH_CHECK("bev_fused_run_keys");
    int rc = radix_sort_pairs_u32(keys_a, vals_a, keys_b, vals_b, (size_t)nruns, bits_for((uint64_t)ncells + 1), sws, sws_bytes, stream);
    if (rc) return rc;
    bev_fused_slot_of_run_kernel<<<dim3(cdiv(nruns, 256)), dim3(256), 0, stream>>>(vals_b, (uint32_t)nruns, slot_of_run);
    BEVAMD_LAUNCH_CHECK("bev_fused_slot_of_run");
  }
  bev_fused_prow_start_kernel<<<dim3(cdiv((long long)ncells + 1, 256)), dim3(256), 0, stream>>>(sorted_keys, (uint32_t)nruns, ncells,
                                                                                               prow_start);
  BEVAMD_LAUNCH_CHECK("bev_fused_prow_start");
  if (nruns > 0 && ncells < RUN_DIRECT) {   // runs alone in their cell: slot -> (1 << 31 | output row), see bev_fused_mark_single_kernel
    bev_fused_mark_single_kernel<<<dim3(cdiv(nruns, 256)), dim3(256), 0, stream>>>(sorted_keys, prow_start, (uint32_t)nruns,
                                                                                   ncells / (uint32_t)b, d, h, w, slot_of_run);
    BEVAMD_LAUNCH_CHECK("bev_fused_mark_single");
  }
  return BEVAMD_OK;
}

/* out [b, d, h, w, c] fp32 (every cell written once) = the fused pooling of bevamd_bev_pool_fused_forward, through the column
 * plan: pass 1 writes partial [nruns, c] fp32 (caller-owned scratch, reusable across calls), pass 2 reduces it per cell.
 * depth [n] fp32, ctx [cams*fh*fw, c] channels-last fp32 (ctx_is_bf16 = 0) or bf16 bits (1). */
int bevamd_bev_pool_fused_forward_columns(const float* depth, const void* ctx, int ctx_is_bf16, const uint32_t* keep,
                                          const uint32_t* endm, const uint32_t* run_first, const uint32_t* slot_of_run,
                                          const uint32_t* prow_start, float* partial, float* out, int n, int nruns, int c,
                                          int depth_bins, int fh, int fw, int b, int d, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(n > 0 && nruns >= 0 && b > 0 && d > 0 && h > 0 && w > 0, "bev_pool_fused_forward_columns: bad sizes");
  ColDims s;
  size_t lds_bytes = 0;
  BEVAMD_REQUIRE(cols_shape(c, depth_bins, fh, fw, s, lds_bytes),
                 "bev_pool_fused_forward_columns: unsupported shape (c %% 4 == 0, fh <= 32, fw %% 4 == 0): c=%d fh=%d fw=%d", c, fh, fw);
  const long long per_cam = (long long)depth_bins * fh * fw;
  BEVAMD_REQUIRE(n % per_cam == 0, "bev_pool_fused_forward_columns: n=%d is not a multiple of depth_bins*fh*fw=%lld", n, per_cam);
  BEVAMD_REQUIRE((unsigned long long)b * d * h * w < 0xFFFFFFF0ull, "bev_pool_fused_forward_columns: b*d*h*w too large");
  BEVAMD_REQUIRE(depth && ctx && keep && endm && run_first && prow_start && out && (nruns == 0 || (slot_of_run && partial)),
                 "bev_pool_fused_forward_columns: null buffer");
  BEVAMD_REQUIRE((((uintptr_t)depth | (uintptr_t)ctx | (uintptr_t)out | (uintptr_t)partial | (uintptr_t)keep) & 15) == 0,
                 "bev_pool_fused_forward_columns: buffers must be 16-byte aligned");
  s.BN = (int)(n / per_cam);
  const uint32_t ncells = (uint32_t)((unsigned long long)b * d * h * w);
  if (nruns > 0) {
    if (lds_bytes > 65536) {
      // the opt-in to > 64 KiB of dynamic LDS is a per-device function attribute: remembered per (flavour, device)
      static int raised[2][64] = {};
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
      dev = dev >= 0 && dev < 64 ? dev : 0;
      int& flag = raised[ctx_is_bf16 ? 1 : 0][dev];
      if (!flag) {
        const void* k = ctx_is_bf16 ? (const void*)&bev_fused_cols_kernel<true> : (const void*)&bev_fused_cols_kernel<false>;
        (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipGetLastError();
        flag = 1;
      }
    }
    const int total = s.BN * s.nwb * s.ndh;
    const dim3 grid(((total + 7) / 8) * 8), block(COL_THREADS);
    if (ctx_is_bf16)
      bev_fused_cols_kernel<true><<<grid, block, lds_bytes, stream>>>(depth, ctx, keep, endm, run_first, slot_of_run, partial, out, s);
    else
      bev_fused_cols_kernel<false><<<grid, block, lds_bytes, stream>>>(depth, ctx, keep, endm, run_first, slot_of_run, partial, out, s);
    BEVAMD_LAUNCH_CHECK("bev_fused_cols");
  }
  const int lpr = c / 4, rpi = 64 / lpr > 0 ? 64 / lpr : 0;
  BEVAMD_REQUIRE(rpi > 0, "bev_pool_fused_forward_columns: c=%d needs more than 64 lanes per row", c);
  bev_fused_reduce_kernel<2><<<dim3(cdiv(cdiv(ncells, 2 * rpi), 4)), dim3(256), 0, stream>>>((const float4*)partial, prow_start, ncells, out,
                                                                                        lpr, rpi, b, d, h, w, c, ncells < RUN_DIRECT ? 1 : 0);
  BEVAMD_LAUNCH_CHECK("bev_fused_reduce");
  return BEVAMD_OK;
}

/* host-only: dynamic LDS bytes of one pass-1 workgroup at this shape (context rows + depth tile + run metadata of 4 image columns
 * x DH depth bins), 0 if the shape is not served.  Two workgroups must fit a compute unit's 160 KiB: <= 81 920 at the flagship. */
size_t bevamd_bev_pool_fused_columns_lds_bytes(int c, int depth_bins, int fh, int fw) {
  ColDims s;
  size_t lds_bytes = 0;
  return cols_shape(c, depth_bins, fh, fw, s, lds_bytes) ? lds_bytes : 0;
}

/* Introspection (tests, tuning): workgroups of pass 1 that fit one compute unit at this shape, from the runtime's occupancy
 * calculator, or MINUS an error code — the tile of the flagship shape (81 520 bytes of LDS) is sized for TWO. */
int bevamd_bev_pool_fused_columns_occupancy(int c, int depth_bins, int fh, int fw) {
  ColDims s;
  size_t lds_bytes = 0;
  if (!cols_shape(c, depth_bins, fh, fw, s, lds_bytes)) {
    set_error("bev_pool_fused_columns_occupancy: unsupported shape c=%d fh=%d fw=%d", c, fh, fw);
    return -BEVAMD_ERR_UNSUPPORTED;
  }
  const void* k = (const void*)&bev_fused_cols_kernel<false>;
  if (lds_bytes > 65536) {
    (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
  }
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, COL_THREADS, lds_bytes) != hipSuccess) {
    (void)hipGetLastError();
    set_error("bev_pool_fused_columns_occupancy: the occupancy query failed");
    return -BEVAMD_ERR_HIP;
  }
  return n;
}

/* Backward of the fused pooling for fp32 context through the column masks (keep / end of bevamd_bev_pool_fused_columns_count):
 * out_grad [b, d, h, w, c] -> d_ctx [cams*fh*fw, c] and d_depth_t [cams, fw, depth_bins, fh] — the depth gradient TRANSPOSED per
 * camera (an image column's values contiguous); the caller views it as [cams, depth_bins, fh, fw].  Same values as
 * bevamd_bev_pool_fused_backward up to fp32 summation order; no atomics, deterministic.  Shapes: c % 4 == 0, fh <= 32,
 * fh * c / 4 <= 1024, depth_bins * fh < 65535 runs per column. */
int bevamd_bev_pool_fused_backward_columns(const float* out_grad, const float* depth, const float* ctx, const uint32_t* keep,
                                           const uint32_t* endm, const uint32_t* cell_of_point, float* d_depth_t, float* d_ctx,
                                           int n, int c, int depth_bins, int fh, int fw, int b, int d, int h, int w, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(n > 0 && c > 0 && (c & 3) == 0 && depth_bins > 0 && fh > 0 && fh <= 32 && fw > 0 && b > 0 && d > 0 && h > 0 && w > 0,
                 "bev_pool_fused_backward_columns: bad sizes (c %% 4 == 0, fh <= 32)");
  const long long per_cam = (long long)depth_bins * fh * fw;
  BEVAMD_REQUIRE(n % per_cam == 0, "bev_pool_fused_backward_columns: n=%d is not a multiple of depth_bins*fh*fw=%lld", n, per_cam);
  BEVAMD_REQUIRE(fh * (c / 4) <= 4 * BWD_THREADS && depth_bins * fh < 65535, "bev_pool_fused_backward_columns: column too large");
  BEVAMD_REQUIRE((unsigned long long)b * d * h * w < 0xFFFFFFF0ull, "bev_pool_fused_backward_columns: b*d*h*w too large");
  BEVAMD_REQUIRE(out_grad && depth && ctx && keep && endm && cell_of_point && d_depth_t && d_ctx, "bev_pool_fused_backward_columns: null buffer");
  BEVAMD_REQUIRE((((uintptr_t)out_grad | (uintptr_t)ctx | (uintptr_t)d_ctx) & 15) == 0, "bev_pool_fused_backward_columns: 16-byte aligned buffers");
  const size_t lds = bwd_cols_lds(c, depth_bins, fh);
  BEVAMD_REQUIRE(lds <= 150 * 1024, "bev_pool_fused_backward_columns: column does not fit LDS (%zu bytes)", lds);
  if (lds > 65536) {
    static int raised[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    dev = dev >= 0 && dev < 64 ? dev : 0;
    if (!raised[dev]) {
      (void)hipFuncSetAttribute((const void*)&bev_fused_bwd_cols_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      (void)hipGetLastError();
      raised[dev] = 1;
    }
  }
  const int bn = (int)(n / per_cam), total = bn * fw;
  bev_fused_bwd_cols_kernel<<<dim3(((total + 7) / 8) * 8), dim3(BWD_THREADS), lds, stream>>>(
      out_grad, depth, ctx, keep, endm, cell_of_point, d_depth_t, d_ctx, bn, depth_bins, fh, fw, c, b, d, h, w);
  BEVAMD_LAUNCH_CHECK("bev_fused_bwd_cols");
  return BEVAMD_OK;
}

}  // extern "C"
