// Hard voxelization for gfx950 — deterministic semantics of the reference, without its O(N^2)
// duplicate scan and its single-thread numbering pass.
//
// Replaces (reference, /root/reference/mmdet3d/ops/voxel/src):
//   voxelization_cuda.cu:25-61    dynamic_voxelize_kernel   (point -> voxel coords)
//   voxelization_cuda.cu:106-147  point_to_voxelidx_kernel  (O(N^2): all previous points)
//   voxelization_cuda.cu:150-180  determin_voxel_num        (<<<1,1>>> serial numbering)
//   voxelization_cuda.cu:64-103   assign_point_to_voxel / assign_voxel_coors
//   voxelization_cuda.cu:231-373  hard_voxelize_gpu (4 cudaDeviceSynchronize + D2H of voxel_num)
// Semantics restated from voxelization_cpu.cpp:46-101 (the serial definition):
//   * voxels are numbered in order of FIRST APPEARANCE in the point list;
//   * only the first max_voxels distinct voxels exist, later ones are dropped with their points;
//   * a voxel keeps its first max_points points, in input order.
//
// Parallel formulation (integer-only, latency/sort bound, not HBM bound):
//   key = linear voxel id (sentinel for out-of-range)          1 kernel
//   stable radix sort (key, point index)                       primitives.hip
//     -> a segment of equal keys lists a voxel's points in input order: slot = j - seg_start,
//        and its head is the voxel's first point
//   mark heads; scatter "is first point of a voxel" back to point order; scan it
//     -> voxel id = number of voxel-opening points before this one  (first-appearance order)
//   one pass writes voxels / coors / num_points (unused slots written as zeros, so the output
//   buffers need no pre-zeroing; the reference's Python side zero-fills 32 MB per call).
#include "common.h"
#include "single_pass.h"

namespace bevamd {

struct VoxGrid {
  float vx, vy, vz;
  float minx, miny, minz;
  int gx, gy, gz;
};

// voxelization_cuda.cu:37,43,50 — c = floor((p - min) / size), fp32 subtract then fp32 divide
// (NOT a multiply by the reciprocal: the quotient decides the voxel index bit-exactly).
__device__ __forceinline__ bool voxel_coord(const float* __restrict__ p, const VoxGrid& g, int& cx, int& cy,
                                            int& cz) {
  float fx = floorf(__fdiv_rn(__fsub_rn(p[0], g.minx), g.vx));
  float fy = floorf(__fdiv_rn(__fsub_rn(p[1], g.miny), g.vy));
  float fz = floorf(__fdiv_rn(__fsub_rn(p[2], g.minz), g.vz));
  bool ok = (fx >= 0.f) && (fx < (float)g.gx) && (fy >= 0.f) && (fy < (float)g.gy) && (fz >= 0.f) &&
            (fz < (float)g.gz);  // NaN compares false -> dropped, like int(NaN) < 0 in the reference
  cx = ok ? (int)fx : -1;
  cy = ok ? (int)fy : -1;
  cz = ok ? (int)fz : -1;
  return ok;
}

__global__ __launch_bounds__(256) void vox_key_kernel(const float* __restrict__ points, int n, int nfeat,
                                                      VoxGrid g, uint32_t ncells, uint32_t* __restrict__ keys,
                                                      uint32_t* __restrict__ vals) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  bool ok = voxel_coord(points + (size_t)i * nfeat, g, cx, cy, cz);
  keys[i] = ok ? (uint32_t)((cx * g.gy + cy) * g.gz + cz) : ncells;
  vals[i] = (uint32_t)i;
}

// dynamic_voxelize (voxelization_cuda.cu:25-61): per-point coords, -1 when out of range.
// The reference writes -1 only to the leading components it had reached when the test failed and
// leaves the others at their previous value; callers test coors[:,0] == -1.  We write (-1,-1,-1),
// which is what voxelization_cpu.cpp:33-38 does.
__global__ __launch_bounds__(256) void dynamic_voxelize_kernel(const float* __restrict__ points, int n,
                                                               int nfeat, VoxGrid g, int* __restrict__ coors) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  int cx, cy, cz;
  voxel_coord(points + (size_t)i * nfeat, g, cx, cy, cz);
  coors[(size_t)i * 3 + 0] = cx;
  coors[(size_t)i * 3 + 1] = cy;
  coors[(size_t)i * 3 + 2] = cz;
}

__global__ __launch_bounds__(256) void vox_heads_kernel(const uint32_t* __restrict__ keys,
                                                        const uint32_t* __restrict__ idx, int n, uint32_t ncells,
                                                        uint32_t* __restrict__ head_flag,
                                                        uint32_t* __restrict__ is_first /*zeroed, point order*/) {
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t k = keys[j];
  bool head = k < ncells && (j == 0 || keys[j - 1] != k);
  head_flag[j] = head ? 1u : 0u;
  if (head) is_first[idx[j]] = 1u;  // stable sort: the head is the voxel's earliest point
}

// per segment: start position, voxel id (= rank of its first point among voxel-opening points)
__global__ __launch_bounds__(256) void vox_segments_kernel(const uint32_t* __restrict__ keys,
                                                           const uint32_t* __restrict__ idx,
                                                           const uint32_t* __restrict__ head_scan,
                                                           const uint32_t* __restrict__ first_scan, int n,
                                                           uint32_t ncells, uint32_t* __restrict__ seg_start,
                                                           uint32_t* __restrict__ seg_vid) {
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t k = keys[j];
  if (k >= ncells) return;
  const bool head = (j == 0 || keys[j - 1] != k);
  const uint32_t seg = head_scan[j] + (head ? 1u : 0u) - 1u;  // head_scan is exclusive
  if (head) {
    seg_start[seg] = (uint32_t)j;
    seg_vid[seg] = first_scan[idx[j]];
  }
  // the last valid row closes the last segment: seg_start[nseg] = number of valid rows
  if (j == n - 1 || keys[j + 1] >= ncells) seg_start[seg + 1] = (uint32_t)(j + 1);
}

// one thread per (sorted row, feature): copies the point into its slot; the head row also writes
// coords / count and zero-fills the voxel's unused slots.
__global__ __launch_bounds__(256) void vox_scatter_kernel(
    const float* __restrict__ points, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ idx,
    const uint32_t* __restrict__ head_flag, const uint32_t* __restrict__ head_scan,
    const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_vid, int n, int nfeat, VoxGrid g,
    uint32_t ncells, int max_points, int max_voxels, float* __restrict__ voxels, int* __restrict__ coors,
    int* __restrict__ num_points_per_voxel) {
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t k = keys[j];
  if (k >= ncells) return;
  const uint32_t hf = head_flag[j];
  const uint32_t seg = head_scan[j] + hf - 1u;
  const uint32_t vid = seg_vid[seg];
  if (vid >= (uint32_t)max_voxels) return;
  const uint32_t start = seg_start[seg];
  const int slot = (int)((uint32_t)j - start);
  float* vbase = voxels + (size_t)vid * max_points * nfeat;
  if (slot < max_points) {
    const float* p = points + (size_t)idx[j] * nfeat;
    float* d = vbase + (size_t)slot * nfeat;
    for (int f = 0; f < nfeat; ++f) d[f] = p[f];
  }
  if (hf) {
    int len = (int)(seg_start[seg + 1] - start);
    int cnt = len < max_points ? len : max_points;
    num_points_per_voxel[vid] = cnt;
    int cz = (int)(k % (uint32_t)g.gz);
    uint32_t t = k / (uint32_t)g.gz;
    int cy = (int)(t % (uint32_t)g.gy);
    int cx = (int)(t / (uint32_t)g.gy);
    coors[(size_t)vid * 3 + 0] = cx;
    coors[(size_t)vid * 3 + 1] = cy;
    coors[(size_t)vid * 3 + 2] = cz;
    float* z = vbase + (size_t)cnt * nfeat;
    for (int e = 0; e < (max_points - cnt) * nfeat; ++e) z[e] = 0.f;
  }
}

__global__ void vox_count_kernel(const uint32_t* __restrict__ nseg, int max_voxels, int* __restrict__ voxel_num) {
  uint32_t s = *nseg;
  *voxel_num = (int)(s < (uint32_t)max_voxels ? s : (uint32_t)max_voxels);
}

// voxelize + mean reduce (bevfusion.py:169-197 with voxelize_reduce): per created voxel,
// feats[vid,:] = sum_{slot<cnt} point / cnt, coords4[vid] = (batch_idx, cx, cy, cz).
// One thread per head row walks its (<= max_points) points in input order.
__global__ __launch_bounds__(256) void vox_mean_kernel(
    const float* __restrict__ points, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ idx,
    const uint32_t* __restrict__ head_flag, const uint32_t* __restrict__ head_scan,
    const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_vid, int n, int nfeat, VoxGrid g,
    uint32_t ncells, int max_points, int max_voxels, int batch_idx, float* __restrict__ feats,
    int* __restrict__ coords4, int* __restrict__ num_points_per_voxel) {
  int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t k = keys[j];
  if (k >= ncells || !head_flag[j]) return;
  const uint32_t seg = head_scan[j];
  const uint32_t vid = seg_vid[seg];
  if (vid >= (uint32_t)max_voxels) return;
  int len = (int)(seg_start[seg + 1] - (uint32_t)j);
  int cnt = len < max_points ? len : max_points;
  const float inv_cnt = (float)cnt;
  if (nfeat <= 8) {
    // point-major walk: one index load per point, all its features accumulated in registers (same per-feature
    // summation order as the reference's sum over the point slots: r ascending)
    float acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = 0.f;
#pragma unroll 2
    for (int r = 0; r < cnt; ++r) {
      const float* p = points + (size_t)idx[j + r] * nfeat;
#pragma unroll
      for (int f = 0; f < 8; ++f)
        if (f < nfeat) acc[f] += p[f];
    }
#pragma unroll
    for (int f = 0; f < 8; ++f)
      if (f < nfeat) feats[(size_t)vid * nfeat + f] = __fdiv_rn(acc[f], inv_cnt);
  } else {
    for (int f = 0; f < nfeat; ++f) {
      float s = 0.f;
      for (int r = 0; r < cnt; ++r) s += points[(size_t)idx[j + r] * nfeat + f];
      feats[(size_t)vid * nfeat + f] = __fdiv_rn(s, inv_cnt);
    }
  }
  if (num_points_per_voxel) num_points_per_voxel[vid] = cnt;
  int cz = (int)(k % (uint32_t)g.gz);
  uint32_t t = k / (uint32_t)g.gz;
  int cy = (int)(t % (uint32_t)g.gy);
  int cx = (int)(t / (uint32_t)g.gy);
  ((int4*)coords4)[vid] = make_int4(batch_idx, cx, cy, cz);
}

static size_t voxelize_ws_bytes(size_t n) {
  size_t a = align_up(n * sizeof(uint32_t), 256);
  size_t a1 = align_up((n + 1) * sizeof(uint32_t), 256);
  size_t s1 = radix_sort_workspace_bytes(n), s2 = scan_dual_workspace_bytes(n);
  // keys_a, vals_a, keys_s, idx_s, head_flag, head_scan, is_first(+scan in place), seg_start(+1), seg_vid, nseg
  return 7 * a + 2 * a1 + 256 + align_up(s1 > s2 ? s1 : s2, 256);
}

struct VoxBuffers {
  uint32_t *keys_a, *vals_a, *keys_s, *idx_s, *head_flag, *head_scan, *first, *seg_start, *seg_vid, *nseg;
  void* sws; size_t sws_bytes;
};

static int make_grid(const float* voxel_size, const float* coors_range, VoxGrid& g) {
  BEVAMD_REQUIRE(voxel_size && coors_range, "voxelize: voxel_size / coors_range are null (host pointers)");
  g.vx = voxel_size[0]; g.vy = voxel_size[1]; g.vz = voxel_size[2];
  g.minx = coors_range[0]; g.miny = coors_range[1]; g.minz = coors_range[2];
  BEVAMD_REQUIRE(g.vx > 0 && g.vy > 0 && g.vz > 0, "voxelize: voxel_size must be positive");
  // voxelization_cuda.cu:255-257: grid = round((max - min) / size), fp32
  g.gx = (int)roundf((coors_range[3] - coors_range[0]) / g.vx);
  g.gy = (int)roundf((coors_range[4] - coors_range[1]) / g.vy);
  g.gz = (int)roundf((coors_range[5] - coors_range[2]) / g.vz);
  BEVAMD_REQUIRE(g.gx > 0 && g.gy > 0 && g.gz > 0, "voxelize: empty grid %d x %d x %d", g.gx, g.gy, g.gz);
  BEVAMD_REQUIRE((unsigned long long)g.gx * g.gy * g.gz < 0xFFFFFFF0ull, "voxelize: grid has >= 2^32 cells");
  return BEVAMD_OK;
}

// shared front half: keys -> sort -> heads -> scans -> segments.  Leaves everything in vb.
static int voxel_segments(const float* points, int n, int nfeat, const VoxGrid& g, uint32_t ncells, VoxBuffers& vb,
                          void* ws, size_t ws_bytes, hipStream_t stream) {
  if (ws == nullptr || ws_bytes < voxelize_ws_bytes((size_t)n)) {
    set_error("voxelize: workspace too small (%zu < %zu)", ws_bytes, voxelize_ws_bytes((size_t)n));
    return BEVAMD_ERR_WORKSPACE;
  }
  Carver cv(ws, ws_bytes);
  vb.keys_a = cv.take<uint32_t>(n);
  vb.vals_a = cv.take<uint32_t>(n);
  vb.keys_s = cv.take<uint32_t>(n);
  vb.idx_s = cv.take<uint32_t>(n);
  vb.head_flag = cv.take<uint32_t>(n);
  vb.head_scan = cv.take<uint32_t>(n);
  vb.first = cv.take<uint32_t>(n);
  vb.seg_start = cv.take<uint32_t>((size_t)n + 1);
  vb.seg_vid = cv.take<uint32_t>((size_t)n + 1);
  vb.nseg = cv.take<uint32_t>(1);
  vb.sws = cv.base + cv.off;
  vb.sws_bytes = ws_bytes - cv.off;

  dim3 grid(cdiv(n, 256)), block(256);
  vox_key_kernel<<<grid, block, 0, stream>>>(points, n, nfeat, g, ncells, vb.keys_a, vb.vals_a);
  BEVAMD_LAUNCH_CHECK("vox_key");
  // the sorted pairs stay wherever the last pass left them (no placement copy)
  int rc = radix_sort_pairs_u32_ex(vb.keys_a, vb.vals_a, vb.keys_s, vb.idx_s, (size_t)n, bits_for((uint64_t)ncells + 1),
                                   vb.sws, vb.sws_bytes, stream, &vb.keys_s, &vb.idx_s);
  if (rc) return rc;
  rc = device_fill_u32(vb.first, (size_t)n, 0u, stream);
  if (rc) return rc;
  vox_heads_kernel<<<grid, block, 0, stream>>>(vb.keys_s, vb.idx_s, n, ncells, vb.head_flag, vb.first);
  BEVAMD_LAUNCH_CHECK("vox_heads");
  rc = exclusive_scan_u32_dual(vb.head_flag, vb.head_scan, vb.nseg, vb.first, vb.first, nullptr, (size_t)n, vb.sws,
                               vb.sws_bytes, stream);
  if (rc) return rc;
  vox_segments_kernel<<<grid, block, 0, stream>>>(vb.keys_s, vb.idx_s, vb.head_scan, vb.first, n, ncells,
                                                  vb.seg_start, vb.seg_vid);
  BEVAMD_LAUNCH_CHECK("vox_segments");
  return BEVAMD_OK;
}

// Batch concatenation without a host sync (the `torch.cat` of bevfusion.py:189-191 needs every sample's voxel count on
// the host): sample b's rows [0, counts[b]) of the padded slabs move to offset sum(counts[:b]) of the packed outputs.
__global__ __launch_bounds__(256) void vox_compact_kernel(const float* __restrict__ feats, const int* __restrict__ coords,
                                                          const int* __restrict__ sizes, const int* __restrict__ counts,
                                                          int batch, int cap, int nfeat, float* __restrict__ out_feats,
                                                          int* __restrict__ out_coords, int* __restrict__ out_sizes,
                                                          int* __restrict__ total) {
  const int b = blockIdx.y;
  int off = 0;
  for (int q = 0; q < b; ++q) off += min(counts[q], cap);
  const int cnt = min(counts[b], cap);
  if (blockIdx.x == 0 && threadIdx.x == 0 && b == batch - 1) *total = off + cnt;
  for (int r = blockIdx.x * 256 + threadIdx.x; r < cnt; r += gridDim.x * 256) {
    const size_t src = (size_t)b * cap + r, dst = (size_t)off + r;
    for (int f = 0; f < nfeat; ++f) out_feats[dst * nfeat + f] = feats[src * nfeat + f];
    ((int4*)out_coords)[dst] = ((const int4*)coords)[src];
    if (sizes && out_sizes) out_sizes[dst] = sizes[src];
  }
}

// ---- batched voxelize + mean: every sweep of the batch in the launches of one ----------------------------------------------
// Sweep b's points occupy [off[b], off[b+1]) of the concatenated key / index arrays (the points themselves stay where they
// are: one pointer per sweep).  The segmented sort orders each range by cell on its own, so "rank of the voxel's first
// point among the voxel-opening points of ITS sweep" — the reference's voxel id — is one flat scan of the first-point
// flags minus the scan value at the sweep's start.  No head / segment arrays: a head row counts its run (<= max_points
// rows) while it sums it.
constexpr int VOX_MAX_BATCH = SORT_MAX_SEGS;
struct VoxBatch {
  const float* pts[VOX_MAX_BATCH];
  uint32_t off[VOX_MAX_BATCH + 1];
  int batch;
};

// The batch kernels run one thread per row of the CONCATENATED arrays, in 256-row chunks dealt to the XCDs in contiguous eighths
// (XCD x = blockIdx.x % 8 takes the x-th eighth): with 8 sweeps per batch every XCD works on one sweep, whose points (6 MB),
// sort buffers and scan (1.2 MB each) then stay in that XCD's L2 instead of being touched from all eight.  The sweep of a row
// is found by comparing against the offsets with a UNIFORM loop index (kernel-argument arrays are read with scalar loads).
struct VoxRow {
  bool in;        // row exists
  int b;          // sweep
  uint32_t J, j;  // position in the concatenated arrays / inside the sweep
  uint32_t lo, n; // first row and row count of the sweep
  const float* pts;
};
__device__ __forceinline__ VoxRow vox_locate(const VoxBatch& vb) {
  VoxRow r;
  const unsigned chunk = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  r.J = chunk * 256u + threadIdx.x;
  r.in = r.J < vb.off[vb.batch];
  r.b = 0;
  r.lo = 0;
  uint32_t hi = vb.off[1];
  r.pts = vb.pts[0];
  for (int q = 1; q < vb.batch; ++q)
    if (r.J >= vb.off[q]) { r.b = q; r.lo = vb.off[q]; hi = vb.off[q + 1]; r.pts = vb.pts[q]; }
  r.j = r.J - r.lo;
  r.n = hi - r.lo;
  return r;
}
static unsigned vox_rows_grid(size_t total) { return (unsigned)(((total + 255) / 256 + 7) / 8 * 8); }

// zero_a / zero_b: the state words of the single-pass kernels that follow (sort passes, scans) — they must be zero when those
// kernels start, and this launch has threads to spare
__global__ __launch_bounds__(256) void vox_key_batch_kernel(VoxBatch vb, int nfeat, VoxGrid g, uint32_t ncells,
                                                            uint32_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                            uint32_t* __restrict__ first, unsigned long long* __restrict__ zero_a,
                                                            size_t words_a, unsigned long long* __restrict__ zero_b,
                                                            size_t words_b) {
  {
    const size_t nthreads = (size_t)gridDim.x * 256, me = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = me; i < words_a; i += nthreads) zero_a[i] = 0ull;
    for (size_t i = me; i < words_b; i += nthreads) zero_b[i] = 0ull;
  }
  const VoxRow r = vox_locate(vb);
  if (!r.in) return;
  int cx, cy, cz;
  bool ok = voxel_coord(r.pts + (size_t)r.j * nfeat, g, cx, cy, cz);
  keys[r.J] = ok ? (uint32_t)((cx * g.gy + cy) * g.gz + cz) : ncells;
  vals[r.J] = r.j;
  first[r.J] = 0u;
}

__global__ __launch_bounds__(256) void vox_heads_batch_kernel(VoxBatch vb, const uint32_t* __restrict__ keys,
                                                              const uint32_t* __restrict__ idx, uint32_t ncells,
                                                              uint32_t* __restrict__ first /*zeroed, point order*/) {
  const VoxRow r = vox_locate(vb);
  if (!r.in) return;
  const uint32_t k = keys[r.J];
  if (k < ncells && (r.j == 0 || keys[r.J - 1] != k)) first[r.lo + idx[r.J]] = 1u;  // stable sort: earliest point
}

// counts[b] = min(voxels of sweep b, max_voxels); rowbase[b] = first output row of sweep b (packed: running sum of the
// counts, else b * max_voxels); total = sum of counts.  first_scan is the exclusive scan, *first_total its grand total.
__global__ void vox_counts_batch_kernel(VoxBatch vb, const uint32_t* __restrict__ first_scan,
                                        const uint32_t* __restrict__ first_total, int max_voxels, int packed,
                                        int* __restrict__ counts, uint32_t* __restrict__ rowbase, int* __restrict__ total) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const uint32_t ntot = vb.off[vb.batch];
  uint32_t run = 0;
  for (int b = 0; b < vb.batch; ++b) {
    const uint32_t lo = vb.off[b] < ntot ? first_scan[vb.off[b]] : *first_total;
    const uint32_t hi = vb.off[b + 1] < ntot ? first_scan[vb.off[b + 1]] : *first_total;
    const uint32_t c = hi - lo < (uint32_t)max_voxels ? hi - lo : (uint32_t)max_voxels;
    counts[b] = (int)c;
    rowbase[b] = packed ? run : (uint32_t)b * (uint32_t)max_voxels;
    run += c;
  }
  if (total) *total = (int)run;
}

// Key-order output (order = 1): the surviving voxels of a sweep — still the first max_voxels by first appearance, the
// reference's cap rule — are written in ascending linear cell index, i.e. in the order the sort already produced, instead of
// first-appearance order.  surv[J] = 1 on the run head of every surviving voxel (sorted order); its exclusive scan minus the
// scan value at the sweep's start is the row.  The encoder's dense BEV output does not depend on the row order of the level-1
// set; rows in linear order are what the staged-rows convolutions and the sorted-key neighbour search need
// (spconv_indice.hip: sp_slab_from_sorted_kernel).
__global__ __launch_bounds__(256) void vox_survivors_batch_kernel(VoxBatch vb, const uint32_t* __restrict__ keys,
                                                                  const uint32_t* __restrict__ idx,
                                                                  const uint32_t* __restrict__ first_scan, uint32_t ncells,
                                                                  int max_voxels, uint32_t* __restrict__ surv) {
  const VoxRow r = vox_locate(vb);
  if (!r.in) return;
  const uint32_t k = keys[r.J];
  uint32_t s = 0u;
  if (k < ncells && (r.j == 0 || keys[r.J - 1] != k)) {
    const uint32_t vid = first_scan[r.lo + idx[r.J]] - first_scan[r.lo];
    s = vid < (uint32_t)max_voxels ? 1u : 0u;
  }
  surv[r.J] = s;
}

// The same flags computed inside the single-pass scan that consumes them (sp::scan_lookback_kernel<SurvLoad>): element J of the
// concatenated sorted arrays.
struct SurvLoad {
  VoxBatch vb;
  const uint32_t* keys;
  const uint32_t* idx;
  const uint32_t* first_scan;
  uint32_t ncells;
  int max_voxels;
  __device__ __forceinline__ uint32_t operator()(size_t J) const {   // branch-free: the scan keeps 32 of these in flight per thread
    uint32_t lo = 0;
    for (int q = 1; q < vb.batch; ++q)
      if (J >= vb.off[q]) lo = vb.off[q];
    const uint32_t k = keys[J];
    const uint32_t kp = keys[J > lo ? J - 1 : J];
    const uint32_t fs = first_scan[lo + idx[J]], f0 = first_scan[lo];
    const bool head = k < ncells && (J == lo || kp != k);
    return head && fs - f0 < (uint32_t)max_voxels ? 1u : 0u;
  }
};

// One thread per sorted row; the first row of every run of equal keys (the voxel's earliest point: the sort is stable) sums
// the run.  The run length comes from the wave's ballot of run boundaries (no dependent walk over the keys), the point indices
// of the run and then the points themselves are fetched as batches of independent, PREDICATED loads (a lane only requests the
// rows it owns) before they are added in input order: two memory round trips per voxel instead of two per point.
// fp32 -> fp16 (dtype 1) / bf16 (dtype 2) bits, round to nearest even — torch's .half() / .bfloat16()
__device__ __forceinline__ uint32_t vox_to_16(float x, int dtype) {
  if (dtype == 1) return (uint32_t)__builtin_bit_cast(unsigned short, (_Float16)x);
  const uint32_t u = __float_as_uint(x);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

__global__ __launch_bounds__(256) void vox_mean_batch_kernel(
    VoxBatch vb, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ idx,
    const uint32_t* __restrict__ first_scan, const uint32_t* __restrict__ surv_scan /*null: first-appearance rows*/,
    const uint32_t* __restrict__ rowbase, int nfeat, VoxGrid g, uint32_t ncells,
    int max_points, int max_voxels, float* __restrict__ feats, int* __restrict__ coords4,
    int* __restrict__ num_points_per_voxel, uint16_t* __restrict__ rows16, int rows16_dtype, int rows16_pitch,
    const uint32_t* __restrict__ first_total, int packed, int* __restrict__ counts, int* __restrict__ total,
    const int* __restrict__ err_scan /*null or the look-back scans' flag*/, const int* __restrict__ err_sort /*... the one-sweep sort's*/) {
  const VoxRow vr = vox_locate(vb);
  const int b = vr.b;
  // rowbase == null: what vox_counts_batch_kernel computes, here — every thread the first output row of ITS sweep (uniform
  // loads), the first thread of the grid the counts and the total (one launch less in front of this kernel)
  uint32_t my_rowbase = 0;
  if (rowbase) {
    my_rowbase = rowbase[b];
  } else {
    const uint32_t ntot = vb.off[vb.batch];
    uint32_t run = 0;
    for (int q = 0; q < vb.batch; ++q) {
      const uint32_t lo = vb.off[q] < ntot ? first_scan[vb.off[q]] : *first_total;
      const uint32_t hi = vb.off[q + 1] < ntot ? first_scan[vb.off[q + 1]] : *first_total;
      const uint32_t c = hi - lo < (uint32_t)max_voxels ? hi - lo : (uint32_t)max_voxels;
      if (q == b) my_rowbase = packed ? run : (uint32_t)q * (uint32_t)max_voxels;
      if (blockIdx.x == 0 && threadIdx.x == 0) counts[q] = (int)c;
      run += c;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (total) *total = (int)run;
      // a single-pass kernel gave up waiting for a predecessor (bounded spin): the voxels are wrong — say so where every caller
      // looks: counts = total = -1 (voxel.py raises on the host route; the encoder's first eager call reads its row counts)
      if ((err_scan && *err_scan) || (err_sort && *err_sort)) {
        for (int q = 0; q < vb.batch; ++q) counts[q] = -1;
        if (total) *total = -1;
      }
    }
  }
  const uint32_t n = vr.n, j = vr.j;
  const bool in = vr.in;
  const uint32_t J = in ? vr.J : 0u;
  const uint32_t k = in ? keys[J] : 0xFFFFFFFFu;
  const bool valid = in && k < ncells;
  const bool head = valid && (j == 0 || keys[J - 1] != k);
  // rows up to the next boundary (next head, first invalid row, end of the sample): inside the wave from the ballot, past its
  // last lane by comparing keys (few runs straddle a wave)
  const unsigned long long bound = __ballot(head || !valid);
  const int lane = threadIdx.x & 63;
  const unsigned long long above = lane < 63 ? bound >> (lane + 1) : 0ull;
  uint32_t len;
  if (above) {
    len = (uint32_t)__ffsll((long long)above);
  } else {
    len = (uint32_t)(64 - lane);
    if (head)
      while (j + len < n && len < (uint32_t)max_points && keys[J + len] == k) ++len;
  }
  if (!head) return;
  const uint32_t i0 = idx[J];
  const uint32_t vid = first_scan[vr.lo + i0] - first_scan[vr.lo];
  if (vid >= (uint32_t)max_voxels) return;
  const size_t row = (size_t)my_rowbase + (surv_scan ? surv_scan[J] - surv_scan[vr.lo] : vid);
  const float* __restrict__ points = vr.pts;
  const int cnt = (int)(len < (uint32_t)max_points ? len : (uint32_t)max_points);
  const float fc = (float)cnt;
  if (nfeat == 5) {   // the nuScenes layout (x, y, z, intensity, time): one 16-byte + one 4-byte load per point
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < cnt; r0 += 4) {
      uint32_t pi[4];
      pi[0] = r0 == 0 ? i0 : idx[J + r0];
#pragma unroll
      for (int u = 1; u < 4; ++u)
        if (r0 + u < cnt) pi[u] = idx[J + r0 + u];
      float v[4][5];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r0 + u < cnt) {
          const float* p = points + (size_t)pi[u] * 5;
          typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
          const f4u q = *(const f4u*)p;
          v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
          v[u][4] = p[4];
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r0 + u < cnt) {
#pragma unroll
          for (int f = 0; f < 5; ++f) acc[f] += v[u][f];   // same per-feature order as the reference's sum over the slots
        }
    }
#pragma unroll
    for (int f = 0; f < 5; ++f) acc[f] = __fdiv_rn(acc[f], fc);
#pragma unroll
    for (int f = 0; f < 5; ++f) feats[row * 5 + f] = acc[f];
    if (rows16 && rows16_pitch == 8) {
      // the encoder's input rows as it stores them (bevamd_spconv_pad_cast_rows: the fp32 mean rounded to 16 bits, zero padding to
      // 8 channels) written by the launch that has the means in registers: one 16-byte store instead of a pass over the rows
      uint32_t h[8];
#pragma unroll
      for (int f = 0; f < 8; ++f) h[f] = f < 5 ? vox_to_16(acc[f], rows16_dtype) : 0u;
      struct alignas(16) Q { uint32_t w[4]; } q = {{h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16)}};
      *(Q*)(rows16 + row * 8) = q;
    }
  } else if (nfeat <= 8) {
    float acc[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) acc[f] = 0.f;
    for (int r = 0; r < cnt; ++r) {
      const float* p = points + (size_t)idx[J + r] * nfeat;
#pragma unroll
      for (int f = 0; f < 8; ++f)
        if (f < nfeat) acc[f] += p[f];
    }
#pragma unroll
    for (int f = 0; f < 8; ++f)
      if (f < nfeat) feats[row * nfeat + f] = __fdiv_rn(acc[f], fc);
  } else {
    for (int f = 0; f < nfeat; ++f) {
      float s = 0.f;
      for (int r = 0; r < cnt; ++r) s += points[(size_t)idx[J + r] * nfeat + f];
      feats[row * nfeat + f] = __fdiv_rn(s, fc);
    }
  }
  if (num_points_per_voxel) num_points_per_voxel[row] = cnt;
  int cz = (int)(k % (uint32_t)g.gz);
  uint32_t t = k / (uint32_t)g.gz;
  int cy = (int)(t % (uint32_t)g.gy);
  int cx = (int)(t / (uint32_t)g.gy);
  ((int4*)coords4)[row] = make_int4(b, cx, cy, cz);
}

static size_t voxelize_batch_ws_bytes(const SortSegs& sg) {
  const size_t n = sg.off[sg.nseg] ? sg.off[sg.nseg] : 1;
  size_t a = align_up(n * sizeof(uint32_t), 256);
  size_t s1 = radix_sort_segmented_workspace_bytes(sg), s2 = scan_workspace_bytes(n);
  // keys_a, vals_a, keys_b, vals_b, first (scanned in place), rowbase + total, the states of the two single-pass scans
  return 5 * a + 3 * 256 + 2 * scan_lookback_state_bytes(n) + align_up(s1 > s2 ? s1 : s2, 256);   // + the scans' error word
}

}  // namespace bevamd

using namespace bevamd;

extern "C" {

size_t bevamd_hard_voxelize_workspace_bytes(int num_points) {
  return voxelize_ws_bytes((size_t)(num_points > 0 ? num_points : 1));
}

int bevamd_hard_voxelize(const float* points, float* voxels, int* coors, int* num_points_per_voxel,
                         const float* voxel_size, const float* coors_range, int max_points, int max_voxels,
                         int num_points, int num_features, int ndim, int deterministic, int* voxel_num_dev,
                         int* voxel_num_host, void* ws, size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  (void)deterministic;  // both modes produce the deterministic result
  BEVAMD_REQUIRE(ndim == 3, "hard_voxelize: NDim=%d unsupported (only 3)", ndim);
  BEVAMD_REQUIRE(num_points >= 0 && num_features >= 3, "hard_voxelize: need num_features >= 3 (got %d)", num_features);
  BEVAMD_REQUIRE(max_points > 0 && max_voxels > 0, "hard_voxelize: max_points/max_voxels must be > 0");
  BEVAMD_REQUIRE(voxel_num_dev != nullptr, "hard_voxelize: voxel_num_dev is null");
  VoxGrid g;
  int rc = make_grid(voxel_size, coors_range, g);
  if (rc) return rc;
  if (num_points == 0) {
    int frc = device_fill_u32((uint32_t*)voxel_num_dev, 1, 0u, stream);
    if (frc) return frc;
    if (voxel_num_host) { BEVAMD_HIP_CHECK(hipStreamSynchronize(stream)); *voxel_num_host = 0; }
    return BEVAMD_OK;
  }
  BEVAMD_REQUIRE(points && voxels && coors && num_points_per_voxel, "hard_voxelize: null buffer");
  const uint32_t ncells = (uint32_t)((unsigned long long)g.gx * g.gy * g.gz);
  VoxBuffers vb;
  rc = voxel_segments(points, num_points, num_features, g, ncells, vb, ws, ws_bytes, stream);
  if (rc) return rc;
  vox_scatter_kernel<<<dim3(cdiv(num_points, 256)), dim3(256), 0, stream>>>(
      points, vb.keys_s, vb.idx_s, vb.head_flag, vb.head_scan, vb.seg_start, vb.seg_vid, num_points, num_features, g,
      ncells, max_points, max_voxels, voxels, coors, num_points_per_voxel);
  BEVAMD_LAUNCH_CHECK("vox_scatter");
  vox_count_kernel<<<1, 1, 0, stream>>>(vb.nseg, max_voxels, voxel_num_dev);
  BEVAMD_LAUNCH_CHECK("vox_count");
  if (voxel_num_host) {
    BEVAMD_HIP_CHECK(hipMemcpyAsync(voxel_num_host, voxel_num_dev, sizeof(int), hipMemcpyDeviceToHost, stream));
    BEVAMD_HIP_CHECK(hipStreamSynchronize(stream));
  }
  return BEVAMD_OK;
}

int bevamd_dynamic_voxelize(const float* points, int* coors, const float* voxel_size, const float* coors_range,
                            int num_points, int num_features, int ndim, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(ndim == 3, "dynamic_voxelize: NDim=%d unsupported (only 3)", ndim);
  BEVAMD_REQUIRE(num_points >= 0 && num_features >= 3, "dynamic_voxelize: need num_features >= 3");
  VoxGrid g;
  int rc = make_grid(voxel_size, coors_range, g);
  if (rc) return rc;
  if (num_points == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(points && coors, "dynamic_voxelize: null buffer");
  dynamic_voxelize_kernel<<<dim3(cdiv(num_points, 256)), dim3(256), 0, stream>>>(points, num_points, num_features, g,
                                                                                 coors);
  BEVAMD_LAUNCH_CHECK("dynamic_voxelize");
  return BEVAMD_OK;
}

int bevamd_voxelize_mean(const float* points, float* feats, int* coords4, int* num_points_per_voxel,
                         const float* voxel_size, const float* coors_range, int max_points, int max_voxels,
                         int num_points, int num_features, int batch_idx, int* voxel_num_dev, void* ws,
                         size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(num_points >= 0 && num_features >= 3, "voxelize_mean: need num_features >= 3");
  BEVAMD_REQUIRE(max_points > 0 && max_voxels > 0, "voxelize_mean: max_points/max_voxels must be > 0");
  BEVAMD_REQUIRE(voxel_num_dev != nullptr, "voxelize_mean: voxel_num_dev is null");
  VoxGrid g;
  int rc = make_grid(voxel_size, coors_range, g);
  if (rc) return rc;
  if (num_points == 0) {
    return device_fill_u32((uint32_t*)voxel_num_dev, 1, 0u, stream);
  }
  BEVAMD_REQUIRE(points && feats && coords4, "voxelize_mean: null buffer");
  const uint32_t ncells = (uint32_t)((unsigned long long)g.gx * g.gy * g.gz);
  VoxBuffers vb;
  rc = voxel_segments(points, num_points, num_features, g, ncells, vb, ws, ws_bytes, stream);
  if (rc) return rc;
  vox_mean_kernel<<<dim3(cdiv(num_points, 256)), dim3(256), 0, stream>>>(
      points, vb.keys_s, vb.idx_s, vb.head_flag, vb.head_scan, vb.seg_start, vb.seg_vid, num_points, num_features, g,
      ncells, max_points, max_voxels, batch_idx, feats, coords4, num_points_per_voxel);
  BEVAMD_LAUNCH_CHECK("vox_mean");
  vox_count_kernel<<<1, 1, 0, stream>>>(vb.nseg, max_voxels, voxel_num_dev);
  BEVAMD_LAUNCH_CHECK("vox_count");
  return BEVAMD_OK;
}

size_t bevamd_voxelize_mean_batch_workspace_bytes(const int* num_points, int batch_size) {
  SortSegs sg;
  if (!num_points || sort_segs_init(sg, num_points, batch_size) != BEVAMD_OK) return 0;
  return voxelize_batch_ws_bytes(sg);
}

int bevamd_voxelize_mean_batch_rows16(const float* const* points, const int* num_points, int batch_size, int num_features,
                                      const float* voxel_size, const float* coors_range, int max_points, int max_voxels,
                                      int packed, int order, float* feats, int* coords4, int* num_points_per_voxel,
                                      int* counts_dev, int* total_dev, void* rows16, int rows16_dtype, int rows16_pitch, void* ws,
                                      size_t ws_bytes, void* stream_);
int bevamd_voxelize_mean_batch_ex(const float* const* points, const int* num_points, int batch_size, int num_features,
                                  const float* voxel_size, const float* coors_range, int max_points, int max_voxels,
                                  int packed, int order, float* feats, int* coords4, int* num_points_per_voxel,
                                  int* counts_dev, int* total_dev, void* ws, size_t ws_bytes, void* stream_);

int bevamd_voxelize_mean_batch(const float* const* points, const int* num_points, int batch_size, int num_features,
                               const float* voxel_size, const float* coors_range, int max_points, int max_voxels,
                               int packed, float* feats, int* coords4, int* num_points_per_voxel, int* counts_dev,
                               int* total_dev, void* ws, size_t ws_bytes, void* stream_) {
  return bevamd_voxelize_mean_batch_ex(points, num_points, batch_size, num_features, voxel_size, coors_range, max_points,
                                       max_voxels, packed, 0, feats, coords4, num_points_per_voxel, counts_dev, total_dev, ws,
                                       ws_bytes, stream_);
}

int bevamd_voxelize_mean_batch_ex(const float* const* points, const int* num_points, int batch_size, int num_features,
                                  const float* voxel_size, const float* coors_range, int max_points, int max_voxels,
                                  int packed, int order, float* feats, int* coords4, int* num_points_per_voxel,
                                  int* counts_dev, int* total_dev, void* ws, size_t ws_bytes, void* stream_) {
  return bevamd_voxelize_mean_batch_rows16(points, num_points, batch_size, num_features, voxel_size, coors_range, max_points,
                                           max_voxels, packed, order, feats, coords4, num_points_per_voxel, counts_dev, total_dev,
                                           nullptr, 0, 0, ws, ws_bytes, stream_);
}

int bevamd_voxelize_mean_batch_rows16(const float* const* points, const int* num_points, int batch_size, int num_features,
                                      const float* voxel_size, const float* coors_range, int max_points, int max_voxels,
                                      int packed, int order, float* feats, int* coords4, int* num_points_per_voxel,
                                      int* counts_dev, int* total_dev, void* rows16, int rows16_dtype, int rows16_pitch, void* ws,
                                      size_t ws_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(!rows16 || ((rows16_dtype == 1 || rows16_dtype == 2) && rows16_pitch == 8 && num_features == 5 &&
                             ((uintptr_t)rows16 & 15) == 0),
                 "voxelize_mean_batch_rows16: 16-bit rows are written for 5 point features into pitch-8 rows (dtype 1 fp16 | 2 bf16)");
  BEVAMD_REQUIRE(order == 0 || order == 1, "voxelize_mean_batch: order %d (0 = first appearance, 1 = linear cell index)", order);
  BEVAMD_REQUIRE(points && num_points, "voxelize_mean_batch: points / num_points are null (host arrays)");
  BEVAMD_REQUIRE(batch_size >= 1 && batch_size <= VOX_MAX_BATCH, "voxelize_mean_batch: batch_size %d (1..%d supported)",
                 batch_size, VOX_MAX_BATCH);
  BEVAMD_REQUIRE(num_features >= 3, "voxelize_mean_batch: need num_features >= 3");
  BEVAMD_REQUIRE(max_points > 0 && max_voxels > 0, "voxelize_mean_batch: max_points/max_voxels must be > 0");
  BEVAMD_REQUIRE(counts_dev != nullptr, "voxelize_mean_batch: counts_dev is null");
  VoxGrid g;
  int rc = make_grid(voxel_size, coors_range, g);
  if (rc) return rc;
  SortSegs sg;
  rc = sort_segs_init(sg, num_points, batch_size);
  if (rc) return rc;
  VoxBatch vbt;
  vbt.batch = batch_size;
  int nmax = 0;
  for (int b = 0; b < batch_size; ++b) {
    BEVAMD_REQUIRE(num_points[b] == 0 || points[b] != nullptr, "voxelize_mean_batch: points[%d] is null", b);
    vbt.pts[b] = points[b];
    vbt.off[b] = sg.off[b];
    if (num_points[b] > nmax) nmax = num_points[b];
  }
  vbt.off[batch_size] = sg.off[batch_size];
  const size_t n = sg.off[batch_size];
  if (n == 0) {
    rc = device_fill_u32((uint32_t*)counts_dev, (size_t)batch_size, 0u, stream);
    if (rc == BEVAMD_OK && total_dev) rc = device_fill_u32((uint32_t*)total_dev, 1, 0u, stream);
    return rc;
  }
  BEVAMD_REQUIRE(feats && coords4, "voxelize_mean_batch: null output buffer");
  if (ws == nullptr || ws_bytes < voxelize_batch_ws_bytes(sg)) {
    set_error("voxelize_mean_batch: workspace too small (%zu < %zu)", ws_bytes, voxelize_batch_ws_bytes(sg));
    return BEVAMD_ERR_WORKSPACE;
  }
  const uint32_t ncells = (uint32_t)((unsigned long long)g.gx * g.gy * g.gz);
  Carver cv(ws, ws_bytes);
  uint32_t* keys_a = cv.take<uint32_t>(n);
  uint32_t* vals_a = cv.take<uint32_t>(n);
  uint32_t* keys_b = cv.take<uint32_t>(n);
  uint32_t* vals_b = cv.take<uint32_t>(n);
  uint32_t* first = cv.take<uint32_t>(n);
  uint32_t* rowbase = cv.take<uint32_t>(VOX_MAX_BATCH);
  uint32_t* first_total = cv.take<uint32_t>(1);
  // the two look-back scans' states + ONE more word behind them: their error flag (bit 0: a predecessor's word never arrived,
  // single_pass.h), zeroed with the states by the key kernel and read by the mean kernel
  unsigned long long* scan_state = (unsigned long long*)cv.take<char>(2 * scan_lookback_state_bytes(n) + 8);
  const size_t scan_state_words = 2 * scan_lookback_state_bytes(n) / 8;
  int* scan_err = (int*)(scan_state + scan_state_words);
  void* sws = cv.base + cv.off;
  const size_t sws_bytes = ws_bytes - cv.off;
  const int nbits = bits_for((uint64_t)ncells + 1);
  const bool single = single_pass_for(n);
  // single-pass kernels (one-sweep sort passes, look-back scans; BEVAMD_SINGLE_PASS=0: the multi-launch ones): 22 -> 9 launches.
  // Their state words are zeroed by the key kernel.
  unsigned long long* sort_state = nullptr;
  size_t sort_words = 0;
  if (single) radix_sort_segmented_state(sg, nbits, sws, &sort_state, &sort_words);

  const dim3 grid(vox_rows_grid(n)), block(256);
  vox_key_batch_kernel<<<grid, block, 0, stream>>>(vbt, num_features, g, ncells, keys_a, vals_a, first, sort_state, sort_words,
                                                   scan_state, single ? scan_state_words + 1 : 0);
  BEVAMD_LAUNCH_CHECK("vox_key_batch");
  uint32_t *keys_s, *idx_s;
  rc = radix_sort_pairs_u32_segmented(keys_a, vals_a, keys_b, vals_b, sg, nbits, sws, sws_bytes, stream, &keys_s, &idx_s,
                                      /*state_zeroed=*/sort_words != 0);
  if (rc) return rc;
  vox_heads_batch_kernel<<<grid, block, 0, stream>>>(vbt, keys_s, idx_s, ncells, first);
  BEVAMD_LAUNCH_CHECK("vox_heads_batch");
  uint32_t* surv = order == 1 ? (keys_s == keys_a ? keys_b : keys_a) : nullptr;   // the sort's other key buffer is free from here on
  if (single) {
    unsigned long long* state_b = scan_state + scan_state_words / 2;
    rc = exclusive_scan_u32_lookback(first, first, n, first_total, scan_state, scan_err, stream);
    if (rc) return rc;
    if (order == 1) {
      sp::scan_lookback_launch(SurvLoad{vbt, keys_s, idx_s, first, ncells, max_voxels}, surv, n, nullptr, state_b, scan_err, stream);
      BEVAMD_LAUNCH_CHECK("vox_survivors_scan");
    }
    vox_mean_batch_kernel<<<grid, block, 0, stream>>>(vbt, keys_s, idx_s, first, surv, nullptr, num_features, g, ncells,
                                                      max_points, max_voxels, feats, coords4, num_points_per_voxel,
                                                      (uint16_t*)rows16, rows16_dtype, rows16_pitch, first_total, packed,
                                                      counts_dev, total_dev, scan_err, sort_state ? (const int*)sort_state + 32 : nullptr);
    BEVAMD_LAUNCH_CHECK("vox_mean_batch");
    return BEVAMD_OK;
  }
  rc = exclusive_scan_u32(first, first, n, first_total, sws, sws_bytes, stream);
  if (rc) return rc;
  vox_counts_batch_kernel<<<1, 64, 0, stream>>>(vbt, first, first_total, max_voxels, packed, counts_dev, rowbase, total_dev);
  BEVAMD_LAUNCH_CHECK("vox_counts_batch");
  if (order == 1) {
    vox_survivors_batch_kernel<<<grid, block, 0, stream>>>(vbt, keys_s, idx_s, first, ncells, max_voxels, surv);
    BEVAMD_LAUNCH_CHECK("vox_survivors_batch");
    rc = exclusive_scan_u32(surv, surv, n, nullptr, sws, sws_bytes, stream);
    if (rc) return rc;
  }
  vox_mean_batch_kernel<<<grid, block, 0, stream>>>(vbt, keys_s, idx_s, first, surv, rowbase, num_features, g, ncells,
                                                    max_points, max_voxels, feats, coords4, num_points_per_voxel,
                                                    (uint16_t*)rows16, rows16_dtype, rows16_pitch, first_total, packed,
                                                    counts_dev, total_dev, nullptr, nullptr);
  BEVAMD_LAUNCH_CHECK("vox_mean_batch");
  return BEVAMD_OK;
}

int bevamd_voxel_compact(const float* feats, const int* coords4, const int* sizes, const int* counts, int batch_size,
                         int max_voxels, int num_features, float* out_feats, int* out_coords4, int* out_sizes,
                         int* total_dev, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(batch_size > 0 && batch_size <= 65535 && max_voxels > 0 && num_features > 0, "voxel_compact: bad sizes");
  BEVAMD_REQUIRE(feats && coords4 && counts && out_feats && out_coords4 && total_dev, "voxel_compact: null buffer");
  long long blocks = ((long long)max_voxels + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  vox_compact_kernel<<<dim3((unsigned)blocks, batch_size), dim3(256), 0, stream>>>(
      feats, coords4, sizes, counts, batch_size, max_voxels, num_features, out_feats, out_coords4, out_sizes, total_dev);
  BEVAMD_LAUNCH_CHECK("vox_compact");
  return BEVAMD_OK;
}

}  // extern "C"
