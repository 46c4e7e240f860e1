/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, gcc) of the reference's hard voxelization.
 * Follows /root/reference/mmdet3d/ops/voxel/src/voxelization_cpu.cpp:8-101 (the serial definition
 * that the CUDA path at voxelization_cuda.cu:106-180 reproduces in parallel).
 *
 * The reference's own CPU code is broken on non-cubic grids (it allocates the lookup grid
 * [gz,gy,gx] and indexes it [x][y][z], voxelization_cpu.cpp:129-130 vs :75,83 — SURVEY.md D4), so a
 * restatement is needed at the real 1440x1440x40 grid.  The coordinate->voxel lookup here is a hash
 * map instead of a dense grid; the rest is line-for-line the same algorithm.
 *
 * Parity pin: tests/golden/voxel_ref_*.npz hold outputs of the reference's hard_voxelize_cpu
 * (oracle/_ref/voxel_layer, compiled from /root/reference) on cubic grids, where it is memory-safe.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* voxelization_cpu.cpp:8-44 (dynamic_voxelize_kernel): c = floor((p - min) / size) per axis in
 * fp32; (-1,-1,-1) if any axis is out of [0, grid).  grid = round((max-min)/size) (:121-124). */
void oracle_dynamic_voxelize(const float* points, int64_t n, int64_t nfeat, const float* voxel_size,
                             const float* coors_range, int32_t* coors) {
  int grid[3];
  for (int a = 0; a < 3; ++a) grid[a] = (int)roundf((coors_range[3 + a] - coors_range[a]) / voxel_size[a]);
  for (int64_t i = 0; i < n; ++i) {
    int c[3];
    int failed = 0;
    for (int a = 0; a < 3; ++a) {
      volatile float diff = points[i * nfeat + a] - coors_range[a];
      volatile float q = diff / voxel_size[a];
      float fl = floorf(q);
      /* the reference casts floor() to int and range-checks; compare in float so that NaN and
       * huge values (undefined int conversion) are simply "out of range" */
      if (!(fl >= 0.0f && fl < (float)grid[a])) { failed = 1; break; }
      c[a] = (int)fl;
    }
    for (int a = 0; a < 3; ++a) coors[3 * i + a] = failed ? -1 : c[a];
  }
}

/* open-addressing hash map: key = linear voxel id, value = voxel index */
typedef struct { uint64_t* keys; int32_t* vals; uint64_t cap; } vmap;
static void vmap_init(vmap* m, uint64_t n) {
  uint64_t cap = 16; while (cap < 2 * n + 16) cap <<= 1;
  m->cap = cap; m->keys = (uint64_t*)malloc(cap * sizeof(uint64_t)); m->vals = (int32_t*)malloc(cap * sizeof(int32_t));
  memset(m->keys, 0xFF, cap * sizeof(uint64_t));
}
static void vmap_free(vmap* m) { free(m->keys); free(m->vals); }
static int32_t* vmap_slot(vmap* m, uint64_t key) {
  uint64_t h = (key * 0x9E3779B97F4A7C15ull) & (m->cap - 1);
  while (m->keys[h] != UINT64_MAX && m->keys[h] != key) h = (h + 1) & (m->cap - 1);
  if (m->keys[h] == UINT64_MAX) { m->keys[h] = key; m->vals[h] = -1; }
  return &m->vals[h];
}

/* voxelization_cpu.cpp:46-101 (hard_voxelize_kernel).  voxels [max_voxels,max_points,nfeat],
 * coors [max_voxels,3], num_points_per_voxel [max_voxels] must be ZEROED by the caller (as
 * voxelize.py:52-54 does).  Returns voxel_num. */
int32_t oracle_hard_voxelize(const float* points, int64_t n, int64_t nfeat, const float* voxel_size,
                             const float* coors_range, int32_t max_points, int32_t max_voxels, float* voxels,
                             int32_t* coors, int32_t* num_points_per_voxel) {
  int32_t* tc = (int32_t*)malloc((size_t)(n > 0 ? n : 1) * 3 * sizeof(int32_t));
  oracle_dynamic_voxelize(points, n, nfeat, voxel_size, coors_range, tc);
  int64_t grid[3];
  for (int a = 0; a < 3; ++a) grid[a] = (int64_t)roundf((coors_range[3 + a] - coors_range[a]) / voxel_size[a]);
  vmap m; vmap_init(&m, (uint64_t)n);
  int32_t voxel_num = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t* c = tc + 3 * i;
    if (c[0] == -1) continue;
    uint64_t key = ((uint64_t)c[0] * (uint64_t)grid[1] + (uint64_t)c[1]) * (uint64_t)grid[2] + (uint64_t)c[2];
    int32_t* slot = vmap_slot(&m, key);
    int32_t voxelidx = *slot;
    if (voxelidx == -1) {
      voxelidx = voxel_num;
      if (max_voxels != -1 && voxel_num >= max_voxels) continue; /* dropped; slot stays -1 */
      voxel_num += 1;
      *slot = voxelidx;
      for (int k = 0; k < 3; ++k) coors[3 * (int64_t)voxelidx + k] = c[k];
    }
    int32_t num = num_points_per_voxel[voxelidx];
    if (max_points == -1 || num < max_points) {
      for (int64_t k = 0; k < nfeat; ++k)
        voxels[((int64_t)voxelidx * max_points + num) * nfeat + k] = points[i * nfeat + k];
      num_points_per_voxel[voxelidx] += 1;
    }
  }
  vmap_free(&m); free(tc);
  return voxel_num;
}

/* models/fusion_models/bevfusion.py:192-195: feats = voxels.sum(dim=1) / count  (fp32).
 * Summation in slot order; the unused slots are zeros and do not change the sum. */
void oracle_voxel_mean(const float* voxels, const int32_t* num_points_per_voxel, int64_t m, int64_t max_points,
                       int64_t nfeat, float* feats) {
  for (int64_t v = 0; v < m; ++v)
    for (int64_t f = 0; f < nfeat; ++f) {
      volatile float s = 0.0f;
      for (int64_t r = 0; r < num_points_per_voxel[v]; ++r) s = s + voxels[(v * max_points + r) * nfeat + f];
      feats[v * nfeat + f] = s / (float)num_points_per_voxel[v];
    }
}
