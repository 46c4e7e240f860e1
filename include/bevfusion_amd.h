/*
 * bevfusion_amd — C ABI of the MI355X (gfx950) BEVFusion hot path.
 *
 * This is the drop-in boundary: every entry point below replaces one native
 * function that the reference (mit-han-lab/bevfusion, /root/reference) exposes to
 * Python through pybind11.  Signatures use plain pointers and sizes only — no
 * torch types — so the library can be bound from ctypes, pybind11 or any FFI.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its comment says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the HIP default stream);
 *     all work is enqueued on it and nothing synchronises unless stated;
 *   - return value: BEVAMD_OK or an error code; bevamd_last_error() returns a
 *     thread-local, human-readable message for the last non-zero return;
 *   - scratch memory is caller-provided: `*_workspace_bytes()` gives the size, the
 *     caller allocates (e.g. torch.empty(uint8)) and passes `ws, ws_bytes`;
 *   - tensors are dense, row-major, contiguous, 16-byte aligned (torch allocations are).
 */
#ifndef BEVFUSION_AMD_H_
#define BEVFUSION_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BEVAMD_OK 0
#define BEVAMD_ERR_INVALID_ARG 1
#define BEVAMD_ERR_WORKSPACE 2
#define BEVAMD_ERR_HIP 3
#define BEVAMD_ERR_UNSUPPORTED 4

const char* bevamd_last_error(void);

/* ------------------------------------------------------------------------- *
 * iou3d  (reference: mmdet3d/ops/iou3d) — rotated boxes in the BEV plane, [x1, y1, x2, y2, angle] fp32
 * ------------------------------------------------------------------------- */

/* Replace iou3d_cuda.boxes_overlap_bev_gpu / boxes_iou_bev_gpu (iou3d.cpp:57-94 -> iou3d_kernel.cu:231-262):
 * ans [num_a, num_b] fp32 = overlap area / IoU of every pair.  Every element is written. */
int bevamd_iou3d_boxes_overlap_bev(const float* boxes_a, int num_a, const float* boxes_b, int num_b,
                                   float* ans_overlap, void* stream);
int bevamd_iou3d_boxes_iou_bev(const float* boxes_a, int num_a, const float* boxes_b, int num_b, float* ans_iou,
                               void* stream);

/* Replace iou3d_cuda.nms_gpu (normal = 0, rotated IoU) and nms_normal_gpu (normal = 1, axis-aligned IoU)
 * (iou3d.cpp:96-180): boxes [num_boxes, 5] sorted by descending score; keep [num_boxes] int64 receives the indices
 * of the kept boxes in order, num_out_dev [1] their number.  The suppression mask and the greedy sweep stay on the
 * device (the reference copies the mask to the host and sweeps there); num_out_host (optional) adds the one
 * synchronisation the reference API implies. */
size_t bevamd_iou3d_nms_workspace_bytes(int num_boxes);
int bevamd_iou3d_nms(const float* boxes, int num_boxes, float thresh, int normal, long long* keep, int* num_out_dev,
                     int* num_out_host, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * bev_pool  (reference: mmdet3d/ops/bev_pool)
 * ------------------------------------------------------------------------- */

/* Replaces bev_pool_ext.bev_pool_forward
 *   (mmdet3d/ops/bev_pool/src/bev_pool_cpu.cpp:22-47 -> bev_pool_cuda.cu:20-42,86-91).
 * x                [n, c]   fp32, rows already sorted by rank
 * geom_feats       [n, 4]   int32 (x, y, z, b) per sorted row
 * interval_lengths [n_intervals] int32      (argument order as in the reference:
 * interval_starts  [n_intervals] int32       lengths BEFORE starts)
 * out              [b, d, h, w, c] fp32 — zero-filled by this call (the reference
 *                  allocates torch::zeros), then out[b,z,x,y,:] = sum of the interval.
 */
int bevamd_bev_pool_forward(const float* x, const int* geom_feats, const int* interval_lengths,
                            const int* interval_starts, float* out, int n, int c, int n_intervals,
                            int b, int d, int h, int w, void* stream);

/* Same with bf16 features (raw uint16 bit patterns), fp32 accumulate, fp32 out.
 * BASELINE config 2 ("bf16 on 1 MI355X"); no reference counterpart (reference is fp32-only). */
int bevamd_bev_pool_forward_bf16(const uint16_t* x, const int* geom_feats, const int* interval_lengths,
                                 const int* interval_starts, float* out, int n, int c, int n_intervals,
                                 int b, int d, int h, int w, void* stream);

/* Replaces bev_pool_ext.bev_pool_backward
 *   (bev_pool_cpu.cpp:60-87 -> bev_pool_cuda.cu:61-84,93-98).
 * out_grad [b,d,h,w,c] fp32 contiguous;  x_grad [n, c] fp32.
 * skip_zero_fill = 0 reproduces the reference (x_grad = zeros, then interval rows
 * written); pass 1 when the intervals are known to cover [0, n). */
int bevamd_bev_pool_backward(const float* out_grad, const int* geom_feats, const int* interval_lengths,
                             const int* interval_starts, float* x_grad, int n, int c, int n_intervals,
                             int b, int d, int h, int w, int skip_zero_fill, void* stream);

/* Precompute = the PyTorch prologue of the reference op, on device, without host syncs:
 *   rank = x*(W*D*B) + y*(D*B) + z*B + b          (bev_pool.py:86-91)
 *   stable sort by rank                            (bev_pool.py:92-93; ties in input order)
 *   interval starts / lengths                      (bev_pool.py:41-46)
 * coords [n,4] (x,y,z,b), int32 or int64 (coords_are_int64).  Rows outside
 * [0,h)x[0,w)x[0,d)x[0,b) are dropped (sorted behind every valid row) — a superset of the
 * reference, whose caller filters first (vtransforms/base.py:160-169).
 * Outputs:
 *   ranks_sorted [n] u32 (dropped rows carry the sentinel b*d*h*w), order [n] u32
 *     (order[j] = input row of sorted row j),
 *   cell_start [b*d*h*w + 2] u32 — CSR over rank-ordered cells: rows of cell r are sorted rows
 *     [cell_start[r], cell_start[r+1]); cell_start[b*d*h*w] = number of kept rows,
 *   optional (NULL to skip) reference-shaped arrays: interval_starts / interval_lengths
 *     (capacity min(n, b*d*h*w)), n_intervals_dev [1], geom_sorted [n,4] int32. */
size_t bevamd_bev_pool_prepare_workspace_bytes(int n, int b, int d, int h, int w);
int bevamd_bev_pool_prepare(const void* coords, int coords_are_int64, int n, int b, int d, int h, int w,
                            uint32_t* ranks_sorted, uint32_t* order, uint32_t* cell_start,
                            int* interval_starts, int* interval_lengths, int* n_intervals_dev,
                            int* geom_sorted, void* ws, size_t ws_bytes, void* stream);

/* Same, straight from the fp32 frustum geometry [n,3] in the lidar frame: replaces
 * BaseTransform.bev_pool's index computation, batch-index concat and range mask
 * (vtransforms/base.py:149-169): idx = trunc((p - (bx - dx/2)) / dx), b = i / (n / batch).
 * bx_minus_half_dx, dx: HOST float[3]. */
int bevamd_bev_pool_prepare_from_geom(const float* geom_xyz, int n, int b, int d, int h, int w,
                                      const float* bx_minus_half_dx, const float* dx,
                                      uint32_t* ranks_sorted, uint32_t* order, uint32_t* cell_start,
                                      int* interval_starts, int* interval_lengths, int* n_intervals_dev,
                                      int* geom_sorted, void* ws, size_t ws_bytes, void* stream);

/* Native forward: reads the UNSORTED feature rows through `order` (the sorted copy
 * `feats[indices]` of bev_pool.py:93 is never materialised) and walks the cell CSR, so ONE
 * launch writes every cell of out [b,d,h,w,c] exactly once (empty cells as zeros): no memset,
 * no interval list, no host sync where the reference has torch.where (bev_pool.py:42).
 * x_is_bf16: 0 fp32, 1 bf16 (fp32 accumulate, fp32 out). */
int bevamd_bev_pool_forward_cells(const void* x, int x_is_bf16, const uint32_t* order,
                                  const uint32_t* cell_start, float* out, int n, int c, int b, int d,
                                  int h, int w, void* stream);

/* Fused depth (x) context -> BEV: DepthLSSTransform.get_cam_feats' outer product
 * (models/vtransforms/depth_lss.py:92-97) folded into the pooling (base.py:141-176), so the [N', C] camera feature
 * volume (638 MB fp32 per frame) is never materialised:
 *   out[cell, :] = sum over the frustum points p of the cell of  depth[p] * ctx[pixel(p), :]
 * depth [n] fp32 = softmax output flattened as [cams, depth_bins, fh, fw] (the point order of the geometry the plan was
 * built from); ctx [cams*fh*fw, c] channels-last, fp32 (ctx_is_bf16 = 0) or bf16 bits (1); order / cell_start from
 * bevamd_bev_pool_prepare[_from_geom]; out [b, d, h, w, c] fp32, every cell written once. */
int bevamd_bev_pool_fused_forward(const float* depth, const void* ctx, int ctx_is_bf16, const uint32_t* order,
                                  const uint32_t* cell_start, float* out, int n, int c, int depth_bins, int fh,
                                  int fw, int b, int d, int h, int w, void* stream);

/* Camera-sector schedule of the fused pooling (static per plan): perm [b*d*h*w] orders the cells by (frame, camera of the
 * cell's first point, cell), xcd_start [9] cuts that order into 8 chunks of equal work; bevamd_bev_pool_fused_forward_scheduled
 * walks chunk x on XCD x, so an XCD reads about one camera's context rows at a time (its own L2 holds them) instead of
 * sweeping the whole BEV.  Same per-cell summation order: bit-identical to bevamd_bev_pool_fused_forward. */
size_t bevamd_bev_pool_fused_schedule_workspace_bytes(int ncells);
int bevamd_bev_pool_fused_schedule(const uint32_t* order, const uint32_t* cell_start, int n, int depth_bins, int fh, int fw,
                                   int b, int d, int h, int w, uint32_t* perm, uint32_t* xcd_start, void* ws, size_t ws_bytes,
                                   void* stream);
int bevamd_bev_pool_fused_forward_scheduled(const float* depth, const void* ctx, int ctx_is_bf16, const uint32_t* order,
                                            const uint32_t* cell_start, const uint32_t* perm, const uint32_t* xcd_start,
                                            float* out, int n, int c, int depth_bins, int fh, int fw, int b, int d, int h, int w,
                                            void* stream);

/* The same op by image COLUMNS (round 4; csrc/bev_pool_fused_cols.hip).  The BEV grid has one z cell, so the fh frustum points
 * of an image column (camera, depth bin, feature column) fall into one cell or a few consecutive RUNS of rows.  Pass 1 loads
 * each context row and depth value once, coalesced, and forms one partial row per run (sum over the run's rows of
 * depth * ctx) — 20-30x fewer rows than points; pass 2 sums every cell's consecutive partial rows and writes every cell once.
 * Same result as bevamd_bev_pool_fused_forward up to fp32 summation order (<= 1e-4 against float64), deterministic.
 * Plan (static per calibration, built from cell_of_point of bevamd_bev_pool_cell_of_point):
 *   _columns_count : keep / end [cams*depth_bins*fw] row masks (bit h: row kept by the range mask / row closes a run),
 *                    run_first [cams*depth_bins*fw] (runs before the column), total_runs (device uint32);
 *   _columns_build : slot_of_run [nruns] (position of a run in the stable (frame, cell) order; a run alone in its cell: 1 << 31 |
 *                    its row of out, stored by pass 1 directly — round 5), prow_start [b*d*h*w + 1]
 *                    (CSR over frame-major cells).  nruns = *total_runs read back once by the caller (plan time).
 * Shapes: c % 4 == 0, fh <= 32, fw % 4 == 0 (bevamd_bev_pool_fused_columns_supported); plans with about as many runs as points
 * (a camera rolled by 90 degrees) should stay on the cell-centric kernels above. */
int bevamd_bev_pool_fused_columns_supported(int c, int depth_bins, int fh, int fw);
int bevamd_bev_pool_fused_backward_columns_supported(int c, int depth_bins, int fh, int fw);   /* the backward's own limits */
/* Introspection: workgroups of the column forward's pass 1 per compute unit at this shape (runtime occupancy calculator; needs a
 * device), or minus a BEVAMD_ERR_* code.  The flagship tile is sized for two. */
int bevamd_bev_pool_fused_columns_occupancy(int c, int depth_bins, int fh, int fw);
/* host-only: dynamic LDS bytes of one workgroup of that pass at this shape (0: shape not served) */
size_t bevamd_bev_pool_fused_columns_lds_bytes(int c, int depth_bins, int fh, int fw);
size_t bevamd_bev_pool_fused_columns_workspace_bytes(int ncols, int nruns);
int bevamd_bev_pool_fused_columns_count(const uint32_t* cell_of_point, int n, int depth_bins, int fh, int fw, int b, int d, int h,
                                        int w, uint32_t* keep, uint32_t* end, uint32_t* run_first, uint32_t* total_runs,
                                        void* ws, size_t ws_bytes, void* stream);
int bevamd_bev_pool_fused_columns_build(const uint32_t* cell_of_point, const uint32_t* end, const uint32_t* run_first, int n,
                                        int nruns, int depth_bins, int fh, int fw, int b, int d, int h, int w,
                                        uint32_t* slot_of_run, uint32_t* prow_start, void* ws, size_t ws_bytes, void* stream);
int bevamd_bev_pool_fused_forward_columns(const float* depth, const void* ctx, int ctx_is_bf16, const uint32_t* keep,
                                          const uint32_t* end, const uint32_t* run_first, const uint32_t* slot_of_run,
                                          const uint32_t* prow_start, float* partial, float* out, int n, int nruns, int c,
                                          int depth_bins, int fh, int fw, int b, int d, int h, int w, void* stream);

/* Backward of the fused pooling through the column masks (fp32 context): one workgroup per image column (camera, w) keeps the
 * gradient rows of the column's runs, its fh context rows and its masked depth values in LDS — every global byte moves once
 * (the point-wise kernels below fetch two 4*c-byte rows per POINT).  d_depth_t is [cams, fw, depth_bins, fh]: the depth gradient
 * with an image column's values contiguous; view it as [cams, depth_bins, fh, fw] by permuting.  d_ctx [cams*fh*fw, c]. */
int bevamd_bev_pool_fused_backward_columns(const float* out_grad, const float* depth, const float* ctx, const uint32_t* keep,
                                           const uint32_t* end, const uint32_t* cell_of_point, float* d_depth_t, float* d_ctx,
                                           int n, int c, int depth_bins, int fh, int fw, int b, int d, int h, int w, void* stream);

/* Backward of the fused op (fp32 context): d_depth [n] = sum_c out_grad[cell(p), c] * ctx[pixel(p), c] (0 for dropped
 * points), d_ctx [cams*fh*fw, c] = sum over the depth bins of a pixel of depth[p] * out_grad[cell(p), :]; both fully
 * written, no atomics.  cell_of_point [n] (rank per point in POINT order) comes from bevamd_bev_pool_cell_of_point. */
int bevamd_bev_pool_cell_of_point(const uint32_t* order, const uint32_t* ranks_sorted, int n, uint32_t* cell_of_point,
                                  void* stream);
int bevamd_bev_pool_fused_backward(const float* out_grad, const float* depth, const float* ctx,
                                   const uint32_t* cell_of_point, float* d_depth, float* d_ctx, int n, int c,
                                   int depth_bins, int fh, int fw, int b, int d, int h, int w, void* stream);

/* Tuning hook of the same kernel family (bench sweeps only): variant 0 = shipped default (15 for fp32 rows, 14 for bf16),
 * 1/2 = one wave per cell (4/8 loads in flight), 3..7 = workgroup-cooperative flavours, 14/15 = 1/2 with the tail rows of a
 * cell as one batch of predicated loads, 16..19 (+ 100 * R) = XCD-striped walks, 23/25 = 2 x 4 cell tiles
 * (csrc/bev_pool.hip::launch_cells_vec; every one-wave-per-cell variant returns the same bits). */
int bevamd_bev_pool_forward_cells_tuned(const void* x, int x_is_bf16, const uint32_t* order,
                                        const uint32_t* cell_start, float* out, int n, int c, int b,
                                        int d, int h, int w, int variant, void* stream);

#ifdef BEVAMD_PROFILING   /* profiling builds only: the rejected XCD-striped walk (variants 16-19) and its host-side map */
/* host-only (tests of the striped variants 16..19 above): groups[p] = the 4-cell group that workgroup p of the padded line `line`
 * takes (-1: padding); returns the padded line length — a multiple of 8, workgroup p runs on XCD p % 8 — or a negative error code. */
int bevamd_bev_pool_striped_line(int row_groups, int stripe_groups, int line, int rot_lines, int* groups, int max_n);
#endif

/* The same gradient written in POINT order (x_grad[i] = out_grad[cell of point i], zeros for dropped points): consecutive
 * 16-byte pieces of x_grad by consecutive threads — a streaming write — with the scattered side on the cached out_grad reads.
 * cell_of_point [n] comes from bevamd_bev_pool_cell_of_point (static per plan).  c % 4 == 0, 16-byte aligned buffers.
 * Same values as bevamd_bev_pool_backward_rows / bev_pool_ext.bev_pool_backward (bev_pool_cuda.cu:61-84), bit for bit. */
int bevamd_bev_pool_backward_points(const float* out_grad, const uint32_t* cell_of_point, float* x_grad, int n, int c, int b,
                                    int d, int h, int w, void* stream);

/* Native backward (row-parallel): x_grad[order[j], :] = out_grad[cell(ranks_sorted[j]), :],
 * zeros for dropped rows.  Every row of x_grad [n,c] is written once. */
int bevamd_bev_pool_backward_rows(const float* out_grad, const uint32_t* order,
                                  const uint32_t* ranks_sorted, float* x_grad, int n, int c, int b,
                                  int d, int h, int w, void* stream);

/* ------------------------------------------------------------------------- *
 * view-transform glue  (reference: mmdet3d/models/vtransforms/base.py — Python in the reference)
 * ------------------------------------------------------------------------- */

/* Replaces the per-sample body of BaseDepthTransform.forward (base.py:283-329): LiDAR points -> one depth image per
 * camera.  points [num_points, num_features] fp32 (xyz first); lidar_aug_inv_rot [3,3] = inverse(lidar_aug_matrix[:3,:3]),
 * lidar_aug_trans [3] = lidar_aug_matrix[:3,3]; lidar2image, img_aug [ncam,4,4]; all DEVICE fp32.
 * depth [ncam, 1, ih, iw] fp32 is fully written (zeros where no point lands).  A pixel hit by several points takes the
 * LAST point in input order (deterministic; = the reference's sequential assignment).  The three GEMMs of the reference are
 * k-ascending fma chains (BLAS sgemm): pixel sets and depths are bit-exact against the reference's own function body on CPU
 * torch given the same inverse.  ws: bevamd_depth_raster_workspace_bytes(ncam, ih, iw) (one u64 per pixel; on return
 * (winning point index + 1) << 32 | depth bits, 0 where nothing landed). */
size_t bevamd_depth_raster_workspace_bytes(int ncam, int ih, int iw);
int bevamd_depth_raster(const float* points, int num_points, int num_features, const float* lidar_aug_inv_rot,
                        const float* lidar_aug_trans, const float* lidar2image, const float* img_aug, int ncam,
                        int ih, int iw, float* depth, void* ws, size_t ws_bytes, void* stream);

/* The per-sample loop of BaseDepthTransform.forward (base.py:283-329) in ONE launch pair for the whole batch: points[b] are
 * DEVICE pointers in a HOST array, num_points[b] host ints; lidar_aug_inv_rot [batch,3,3]; lidar_aug_trans: row b at
 * lidar_aug_trans + b*trans_stride floats (3 for a packed [batch,3]); lidar2image / img_aug [batch,ncam,4,4];
 * depth [batch,ncam,1,ih,iw]; ws: batch * bevamd_depth_raster_workspace_bytes(ncam, ih, iw).  Per sample identical to
 * bevamd_depth_raster (same arithmetic, last point in input order wins). */
int bevamd_depth_raster_batch(const float* const* points, const int* num_points, int batch, int num_features,
                              const float* lidar_aug_inv_rot, const float* lidar_aug_trans, int trans_stride,
                              const float* lidar2image, const float* img_aug, int ncam, int ih, int iw, float* depth,
                              void* ws, size_t ws_bytes, void* stream);
/* The same over a PERSISTENT map: ws must be all zero when the call starts and is all zero again once its kernels have run (the
 * unpack pass clears what it reads): two launches instead of three.  For a caller that keeps one zero-initialised workspace per
 * (device, size) and issues its rasters on one stream. */
int bevamd_depth_raster_batch_zero_ws(const float* const* points, const int* num_points, int batch, int num_features,
                                      const float* lidar_aug_inv_rot, const float* lidar_aug_trans, int trans_stride,
                                      const float* lidar2image, const float* img_aug, int ncam, int ih, int iw, float* depth,
                                      void* ws, size_t ws_bytes, void* stream);

/* Replaces BaseTransform.get_geometry (base.py:92-135): frustum [frustum_points, 3] (u, v, d) -> geom
 * [batch*cams, frustum_points, 3] in the lidar frame.  post_rot_inv [batch*cams,3,3] = inverse(img_aug[:3,:3]),
 * post_trans [batch*cams,3], combine [batch*cams,3,3] = camera2lidar_rot @ inverse(intrinsics), camera2lidar_trans
 * [batch*cams,3], extra_rot [batch,3,3] / extra_trans [batch,3] (LiDAR augmentation, either may be NULL). DEVICE fp32.
 * fp32 op for op in the reference's order (products and sums rounded separately, as ATen's per-point bmm does): bit-exact
 * against the reference's own function body on CPU torch given the same two inverse-derived matrices. */
int bevamd_lss_geometry(const float* frustum, int frustum_points, const float* post_rot_inv, const float* post_trans,
                        const float* combine, const float* camera2lidar_trans, const float* extra_rot,
                        const float* extra_trans, int batch_size, int cams_per_sample, float* geom, void* stream);

/* out[i] = inverse of the 3x3 matrix at m + i*mat_stride + r*row_stride + c (floats; the top-left block of a [.., 4, 4]
 * tensor is mat_stride 16, row_stride 4), i < count; out [count, 3, 3] packed.  Replaces `torch.inverse` at
 * base.py:106 (post_rots), :118 (intrins), :292 (lidar_aug_matrix[:3,:3]) on the device path: adjugate / determinant in
 * fp64, one rounding to fp32; no workspace, no host sync (LAPACK's getrf/getrs result may differ in the last 1-2 ulp). */
int bevamd_mat3_inverse(const float* m, long long mat_stride, long long row_stride, int count, float* out,
                        void* stream);
/* ... and col[i][r] = m[i*mat_stride + r*row_stride + 3] (row_stride >= 4), packed [count, 3]: inverse rotation and translation of
 * [count, 4, 4] augmentation matrices (base.py:289-292) from one launch. */
int bevamd_mat3_inverse_with_column(const float* m, long long mat_stride, long long row_stride, int count, float* out, float* col,
                                    void* stream);

/* The per-camera matrices bevamd_lss_geometry consumes, from the raw calibration (base.py:106, 118):
 * post_rot_inv[i] = inverse(post_rots[i]); combine[i] = camera2lidar_rots[i] @ inverse(intrins[i]) (product in the order
 * of ATen's small-bmm kernel), i < ncam_total.  The three inputs share one (mat_stride, row_stride) addressing. */
int bevamd_lss_camera_matrices(const float* post_rots, const float* camera2lidar_rots, const float* intrins,
                               long long mat_stride, long long row_stride, int ncam_total, float* post_rot_inv,
                               float* combine, void* stream);

/* ------------------------------------------------------------------------- *
 * voxelization  (reference: mmdet3d/ops/voxel)
 * ------------------------------------------------------------------------- */

/* Replaces voxel_layer.hard_voxelize
 *   (mmdet3d/ops/voxel/src/voxelization.cpp:6-11 -> voxelization.h:58-81 ->
 *    voxelization_cuda.cu:231-373 hard_voxelize_gpu; semantics of voxelization_cpu.cpp:46-101).
 * points [num_points, num_features] fp32 (xyz first); outputs sized by the caller for max_voxels:
 * voxels [max_voxels, max_points, num_features] fp32, coors [max_voxels, 3] int32 (x, y, z),
 * num_points_per_voxel [max_voxels] int32.  Rows [0, voxel_num) are fully written (unused point
 * slots as zeros), so the buffers need NOT be pre-zeroed; rows >= voxel_num are left untouched.
 * voxel_size: HOST float[3]; coors_range: HOST float[6] (xyz min, xyz max).
 * Voxels are numbered in order of first appearance in `points`; at most max_voxels voxels and
 * max_points points per voxel (input order) are kept.  `deterministic` is accepted for signature
 * parity; the result is always the deterministic one.
 * voxel_num_dev [1] (device) always receives the count.  If voxel_num_host != NULL the call also
 * synchronises `stream` and stores the count there — the reference returns a host int
 * (voxelization_cuda.cu:369-372); pass NULL to stay asynchronous. */
size_t bevamd_hard_voxelize_workspace_bytes(int num_points);
int bevamd_hard_voxelize(const float* points, float* voxels, int* coors, int* num_points_per_voxel,
                         const float* voxel_size, const float* coors_range, int max_points,
                         int max_voxels, int num_points, int num_features, int ndim, int deterministic,
                         int* voxel_num_dev, int* voxel_num_host, void* ws, size_t ws_bytes, void* stream);

/* Replaces voxel_layer.dynamic_voxelize (voxelization_cuda.cu:25-61, 485-...): coors [num_points,3]
 * int32 = floor((p - min) / size), or (-1,-1,-1) for points outside the range. */
int bevamd_dynamic_voxelize(const float* points, int* coors, const float* voxel_size,
                            const float* coors_range, int num_points, int num_features, int ndim,
                            void* stream);

/* Fused BEVFusion.voxelize step for one sample (models/fusion_models/bevfusion.py:169-197 with
 * voxelize_reduce): hard voxelization + mean over the kept points + batch-index column, without
 * materialising voxels[max_voxels, max_points, F]:
 *   feats [max_voxels, num_features] = sum(points of voxel, input order) / count
 *   coords4 [max_voxels, 4] int32 = (batch_idx, x, y, z);  num_points_per_voxel optional (NULL ok). */
int bevamd_voxelize_mean(const float* points, float* feats, int* coords4, int* num_points_per_voxel,
                         const float* voxel_size, const float* coors_range, int max_points,
                         int max_voxels, int num_points, int num_features, int batch_idx,
                         int* voxel_num_dev, void* ws, size_t ws_bytes, void* stream);

/* The same step for a whole batch in the launches of one (bevfusion.py:176-191: the per-sample loop, the F.pad batch
 * column and the torch.cat): points / num_points are HOST arrays of batch_size device pointers / point counts
 * (batch_size <= 64).  One segmented radix sort orders every sweep by cell; results are those of batch_size
 * bevamd_voxelize_mean calls with batch_idx = 0..batch_size-1.
 *   packed = 0: sample b's rows start at row b * max_voxels (padded slabs, what bevamd_voxel_compact consumes);
 *   packed = 1: rows are packed sample after sample (the torch.cat) — the buffers still hold batch_size * max_voxels rows.
 * counts_dev [batch_size] receives min(voxels of the sample, max_voxels), total_dev [1] (optional) their sum.  No host
 * synchronisation. */
size_t bevamd_voxelize_mean_batch_workspace_bytes(const int* num_points, int batch_size);
int bevamd_voxelize_mean_batch(const float* const* points, const int* num_points, int batch_size, int num_features,
                               const float* voxel_size, const float* coors_range, int max_points, int max_voxels,
                               int packed, float* feats, int* coords4, int* num_points_per_voxel, int* counts_dev,
                               int* total_dev, void* ws, size_t ws_bytes, void* stream);

/* The same with a choice of row order.  order = 0: first appearance (the reference's voxel numbering, bevfusion.py:176-191 /
 * voxelization_cuda.cu:231-373; what bevamd_voxelize_mean_batch does).  order = 1: the SAME surviving set per sample (the first
 * max_voxels voxels by first appearance — the reference's cap rule) and the same per-voxel results, written in ascending linear
 * cell index (x, y, z; z fastest) inside each sample.  The SparseEncoder's dense BEV output does not depend on the row order of
 * its input set; rows in linear order let level 1 use the staged-rows convolutions and sorted-key neighbour search. */
int bevamd_voxelize_mean_batch_ex(const float* const* points, const int* num_points, int batch_size, int num_features,
                                  const float* voxel_size, const float* coors_range, int max_points, int max_voxels,
                                  int packed, int order, float* feats, int* coords4, int* num_points_per_voxel,
                                  int* counts_dev, int* total_dev, void* ws, size_t ws_bytes, void* stream);

/* The same, additionally writing the voxel means as the 16-bit, zero-padded rows the SparseEncoder's first convolution reads
 * (rows16 [batch_size * max_voxels, 8]; rows16_dtype 1 fp16 | 2 bf16; the fp32 mean rounded once, as the cast of
 * functional.py:24 custom_fwd(cast_inputs=torch.half) does) from the launch that holds the means in registers — the separate
 * pad-and-cast pass over the rows (bevamd_spconv_pad_cast_rows) disappears from the LiDAR branch.  5 point features only. */
int bevamd_voxelize_mean_batch_rows16(const float* const* points, const int* num_points, int batch_size, int num_features,
                                      const float* voxel_size, const float* coors_range, int max_points, int max_voxels,
                                      int packed, int order, float* feats, int* coords4, int* num_points_per_voxel,
                                      int* counts_dev, int* total_dev, void* rows16, int rows16_dtype, int rows16_pitch, void* ws,
                                      size_t ws_bytes, void* stream);

/* Batch concatenation of BEVFusion.voxelize (bevfusion.py:189-191) without a host sync: the padded per-sample slabs
 * feats [batch, max_voxels, num_features], coords4 [batch, max_voxels, 4], sizes [batch, max_voxels] (optional) with the
 * device counts [batch] written by bevamd_voxelize_mean are packed sample after sample into out_* (capacity
 * batch*max_voxels rows); total_dev [1] receives the number of packed rows. */
int bevamd_voxel_compact(const float* feats, const int* coords4, const int* sizes, const int* counts, int batch_size,
                         int max_voxels, int num_features, float* out_feats, int* out_coords4, int* out_sizes,
                         int* total_dev, void* stream);

/* ------------------------------------------------------------------------- *
 * spconv: rulebook + sparse convolution  (reference: mmdet3d/ops/spconv)
 * ------------------------------------------------------------------------- *
 * Geometry arrays (in_shape, out_shape, ksize, stride, padding, dilation) are HOST int[3].
 * indices: [n, 4] int32 (batch, x, y, z) as in SparseConvTensor.indices (structure.py:21-37).
 * Offsets are numbered offset = (kx*Ky + ky)*Kz + kz, k = in - out*stride + pad (geometry.h:67-69).
 * dtype codes: 0 = fp32, 1 = fp16, 2 = bf16. */

/* The native rulebook is OUTPUT-STATIONARY: nbr [kernel_volume, nbr_stride] int32,
 * nbr[k][o] = input row that feeds output row o through offset k, or -1.
 *
 * bevamd_spconv_build_rulebook replaces sparse_conv_ext.get_indice_pairs_3d
 *   (spconv/src/all.cc:21-51 -> spconv_ops.h:27-141 getIndicePair<3>; kernels indice.cu.h:22-203).
 * subm != 0: out rows == in rows (out_indices may be NULL or == indices); stride is forced to 1 and
 *   padding to ksize/2 as spconv_ops.h:78-81 does.
 * subm == 0: out_indices [out_cap, 4] receives the active outputs in ASCENDING linear index
 *   (b, x, y, z) — the row order of the reference's CUDA path (torch::_unique, spconv_ops.h:130).
 *   out_cap / nbr_stride must be >= bevamd_spconv_max_outputs(...) unless the caller knows better.
 * dilation (HOST int[3], NULL = 1): in = out*stride - padding + k*dilation (geometry.h:24-85); SubM keeps
 *   padding = ksize/2 whatever the dilation, as the reference does (spconv_ops.h:78-81).
 * transpose != 0 (ignored for subm): out = in*stride - padding + k*dilation (geometry.h:86-141
 *   getValidOutPosTranspose); out_shape is the caller's get_deconv_output_size.
 * 2D convolutions run as 3D ones with a unit z axis (ksize 1, stride 1, padding 0); 4D is not implemented.
 * num_out_dev [1] always receives the row count; num_out_host (optional) makes the call
 * synchronise and return it to the host. */
size_t bevamd_spconv_rulebook_workspace_bytes(int n, int batch_size, const int* out_shape, int subm);
int bevamd_spconv_max_outputs(int n, const int* ksize, const int* stride, int subm);
int bevamd_spconv_max_outputs_ex(int n, const int* ksize, const int* stride, const int* dilation, int subm,
                                 int transpose);
int bevamd_spconv_build_rulebook(const int* indices, int n, int batch_size, const int* in_shape,
                                 const int* out_shape, const int* ksize, const int* stride,
                                 const int* padding, const int* dilation, int subm, int transpose,
                                 int* out_indices, int out_cap, int* nbr, int nbr_stride, int* num_out_dev,
                                 int* num_out_host, void* ws, size_t ws_bytes, void* stream);

/* Sync-free building blocks of the same rulebook (what the inference path of SparseEncoder chains): every
 * count may stay on the device — `n_cap` / `m_cap` bound the launches and buffers, `n_dev` / `m_dev`
 * (optional device ints) hold the live counts.
 *   hash index  : open-addressing table of a voxel set in ANY row order (bevamd_spconv_hash_index_bytes);
 *   rank index  : (bitmap, popcount prefix) words of a voxel set whose rows are in ascending linear index
 *                 — produced by bevamd_spconv_downsample for its outputs (bevamd_spconv_rank_index_bytes);
 *   downsample  : active outputs of a strided convolution = getIndicePair's non-subm branch
 *                 (spconv_ops.h:100-139 + torch::_unique), rows ascending, count in num_out_dev (clamped to out_cap);
 *                 with nbr != NULL it also writes the convolution's neighbour table nbr[K][nbr_stride], generated
 *                 from the input side (each input row scatters into the <= prod(ceil(k/s)) outputs it feeds);
 *   neighbors   : nbr[k][o] through an index of the INPUT set (index_kind 0 = hash, 1 = rank;
 *                 in_index_n_cap = the n_cap the hash index was built with). */
size_t bevamd_spconv_hash_index_bytes(int n_cap);
size_t bevamd_spconv_rank_index_bytes(int batch_size, const int* shape);
int bevamd_spconv_hash_index_build(const int* indices, int n_cap, const int* n_dev, int batch_size,
                                   const int* shape, void* index, size_t index_bytes, void* stream);
int bevamd_spconv_downsample(const int* indices, int n_cap, const int* n_dev, int batch_size,
                             const int* in_shape, const int* out_shape, const int* ksize, const int* stride,
                             const int* padding, int* out_indices, int out_cap, int* num_out_dev,
                             void* out_index, size_t out_index_bytes, int* nbr, int nbr_stride, void* stream);
/* The same output set, count and rank index for an input set in ascending linear index that owns a lookup structure (src_kind 0:
 * the x-plane directory of its sorted-key index, int32 [batch * in_shape[0] + 1]; 1: its rank-index words), in two launches
 * instead of four and without the byte map of the general entry (csrc/spconv_indice.hip: sp_rank_tiles_from_rows_kernel).
 * No neighbour table: bevamd_spconv_neighbors builds it from the output side.  Replaces the same reference lines. */
int bevamd_spconv_downsample_sorted(const int* indices, int n_cap, const int* n_dev, int batch_size, const int* in_shape,
                                    const int* out_shape, const int* ksize, const int* stride, const int* padding,
                                    int src_kind, const void* src, int* out_indices, int out_cap, int* num_out_dev,
                                    void* out_index, size_t out_index_bytes, void* stream);
int bevamd_spconv_neighbors(const int* out_indices, int m_cap, const int* m_dev, int batch_size,
                            const int* in_shape, const int* out_shape, const int* ksize, const int* stride,
                            const int* padding, int subm, int index_kind, const void* in_index,
                            int in_index_n_cap, int* nbr, int nbr_stride, void* stream);

/* Dense BEV tail of SparseEncoder.forward (models/backbones/sparse_encoder.py:126-131: SparseConvTensor.dense()
 * structure.py:49-59 -> permute(0,1,4,2,3) -> view): out [batch, channels*Z, X, Y], same element size as features,
 * out[b][c*Z+z][x][y] = features[row(b,x,y,z)][c], zeros elsewhere; every element is written once (no pre-zeroing).
 * Rows are found through an index of the voxel set (kind 0 = hash, 1 = rank). shape: HOST int[3] = (X, Y, Z). */
int bevamd_spconv_dense_bev(const void* features, int elem_bytes, int pitch, int channels, int index_kind,
                            const void* index, int index_n_cap, int batch_size, const int* shape, void* out,
                            void* stream);

/* The reference's rulebook arrays (spconv_ops.h:56-59): indice_pairs [K, 2, pairs_len] int32 (-1
 * padded; [k][0] = input rows, [k][1] = output rows), indice_num [K].  Pairs of one offset are listed
 * by ascending output row (the reference's CUDA order is atomicAdd order, i.e. unspecified). */
size_t bevamd_spconv_pairs_workspace_bytes(int m, int kernel_volume);
int bevamd_spconv_pairs_from_nbr(const int* nbr, int nbr_stride, int m, int kernel_volume,
                                 int* indice_pairs, int pairs_len, int* indice_num, void* ws,
                                 size_t ws_bytes, void* stream);
/* ...and back, for callers that hold reference-shaped pairs (the drop-in indice_conv entry points).
 * inverse != 0 swaps the two pair columns (spconv_ops.h:317,348). */
int bevamd_spconv_nbr_from_pairs(const int* indice_pairs, int pairs_len, const int* indice_num,
                                 int kernel_volume, int inverse, int* nbr, int nbr_stride, void* stream);
/* input-stationary view: nbr_t[k][nbr[k][o]] = o (nbr_t pre-filled with -1 by the call). */
int bevamd_spconv_transpose_nbr(const int* nbr, int nbr_stride, int m, int kernel_volume, int* nbr_t,
                                int nbr_t_stride, void* stream);

/* Filters [kx,ky,kz,cin,cout] (conv.py:100) -> MFMA-friendly image [cout_pad][K][cin_pad], zero
 * padded.  transpose_io != 0 prepares W^T (cin and cout swap roles) for the input-gradient pass. */
size_t bevamd_spconv_prepared_filter_elems(int dtype, int kernel_volume, int cin, int cout, int transpose_io);
int bevamd_spconv_prepare_filters(const void* filters, int dtype, int kernel_volume, int cin, int cout,
                                  int transpose_io, void* prepared, void* stream);

/* Replaces sparse_conv_ext.indice_conv_{fp32,half} (all.cc:28-31 -> spconv_ops.h:260-361) and the
 * gather / mm_out / scatter-add loop it runs per offset:
 *   out[o, :] = epilogue( sum_k features[nbr[k][o], :] @ W[k] ),  fp32 accumulation, one launch.
 * Optional epilogue (NULL to skip): bias [cout] (features' dtype), folded BatchNorm bn_scale /
 * bn_shift [cout] fp32, residual [num_out, cout] (features' dtype), relu flag.
 * Rows: num_out, or *num_out_dev when non-NULL (num_out then bounds the launch).  cout <= 128. */
int bevamd_spconv_conv_forward(const void* features, int dtype, const void* prepared, const int* nbr,
                               int nbr_stride, int num_out, const int* num_out_dev, int kernel_volume,
                               int cin, int cout, void* out, const void* bias, const float* bn_scale,
                               const float* bn_shift, const void* residual, int relu, void* stream);

/* fp32 sparse convolution on the bf16 matrix cores by three-way operand splitting (round 5).  Replaces
 * sparse_conv_ext.indice_conv_fp32 and the input-gradient half of indice_conv_backward_fp32 (spconv/src/all.cc:28-31 ->
 * spconv_ops.h:260-456) for fp32 rows of exactly 16 | 32 | 64 | 128 channels, cout <= 128: x = hi + mid + lo (bf16 pieces by
 * truncation, exact), six v_mfma_f32_16x16x32_bf16 per product with fp32 accumulation — error of the order of one fp32 rounding.
 * image: bevamd_spconv_make_filter_image3 of the fp32 filters [K, cin, cout] (transpose_io = 1: the input-gradient pass on the
 * transposed table); nbr / epilogue operands as bevamd_spconv_conv_forward (all fp32).  csrc/spconv_tile_f32x3.hip. */
int bevamd_spconv_f32x3_supported(int cin, int cout);
size_t bevamd_spconv_filter_image3_elems(int kernel_volume, int cin, int cout, int transpose_io);   /* uint16 elements */
int bevamd_spconv_make_filter_image3(const float* filters, int kernel_volume, int cin, int cout, int transpose_io, void* image,
                                     void* stream);
int bevamd_spconv_conv_forward_f32x3(const float* features, int feat_stride, int num_in, const void* image, const int* nbr,
                                     int nbr_stride, int num_out, const int* num_out_dev, int kernel_volume, int cin, int cout,
                                     float* out, int out_stride, const float* bias, const float* bn_scale, const float* bn_shift,
                                     const float* residual, int residual_stride, int relu, void* stream);

/* Tiled forward for 16-bit features (dtype 1 = fp16, 2 = bf16; channels <= 128) — the kernel the modules
 * use at inference.  Same operator as bevamd_spconv_conv_forward (sparse_conv_ext.indice_conv_half /
 * fused_indice_conv_half, all.cc:28-37 -> spconv_ops.h:260-361), with
 *   - `image`: the filter in MFMA-fragment order, made once per weight by bevamd_spconv_make_filter_image
 *     (bevamd_spconv_filter_image_elems elements of the feature dtype);
 *   - features [num_in, feat_stride]: the pitch must be a multiple of 8 elements and cover cin rounded up to
 *     8 / 16 / 32 / 64 / 128, the padding channels zero;
 *   - the epilogue y = relu?( bn_scale * (conv + bias) + bn_shift + residual ), every operand optional
 *     (eval-mode BatchNorm1d folded to fp32 scale/shift, SparseBasicBlock's identity add, sparse_block.py:88-107);
 *   - num_out_dev (optional, device int): actual row count, num_out then only bounds the launch;
 *   - variant 0 = automatic; other values select a kernel for tuning (see spconv_tile_impl.h). */
int bevamd_spconv_tiled_supported(int dtype, int cin, int cout);
size_t bevamd_spconv_filter_image_elems(int kernel_volume, int cin, int cout, int transpose_io);
int bevamd_spconv_make_filter_image(const void* filters, int dtype, int kernel_volume, int cin, int cout,
                                    int transpose_io, void* image, void* stream);
/* Filter images of n <= 48 convolutions in ONE launch (the training path: forward and input-gradient images of every layer of a
 * step, straight from the fp32 master weights).  filters[i] [K, cin, cout] (fp32 when flags[i] & 4, else `dtype`); images[i] of
 * bevamd_spconv_filter_image_elems(K, cin, cout, flags[i] & 1) elements of `dtype`; flags[i] & 1 = transpose_io, & 2 = kernel
 * offsets mirrored (k -> K - 1 - k: the input gradient of a symmetric SubM rulebook walks the same table with the mirrored
 * transposed filter).  No reference counterpart (spconv_ops.h:322-334 hands torch::mm the filter as it is). */
int bevamd_spconv_make_filter_images(int n, const void* const* filters, void* const* images, const int* kernel_volume, const int* cin,
                                     const int* cout, const int* flags, int dtype, void* stream);
int bevamd_spconv_conv_forward_tiled(const void* features, int dtype, int feat_stride, int num_in,
                                     const void* image, const int* nbr, int nbr_stride, int num_out,
                                     const int* num_out_dev, int kernel_volume, int cin, int cout, void* out,
                                     int out_stride, const void* bias, const float* bn_scale,
                                     const float* bn_shift, const void* residual, int residual_stride,
                                     int relu, int variant, void* stream);
/* The same call with the rulebook of a 3x3x3 convolution given as slab metadata (hdr / slots of bevamd_spconv_slab_build*,
 * block_rows = 128 | 256) instead of the int32 neighbour table: the gather kernels decode `first row of the plane + 16-bit slot`
 * while they load a tile's table.  Same kernels and results, half the rulebook bytes, no table to clear and scatter
 * (spconv_ops.h:27-141 writes 2 x 27 x N int32 per rulebook). */
int bevamd_spconv_conv_forward_tiled_slots(const void* features, int dtype, int feat_stride, int num_in, const void* image,
                                           const void* hdr, const void* slots, int block_rows, int num_out,
                                           const int* num_out_dev, int cin, int cout, void* out, int out_stride,
                                           const void* bias, const float* bn_scale, const float* bn_shift, const void* residual,
                                           int residual_stride, int relu, int variant, void* stream);
/* dst [n, pitch] (dtype 1 = fp16, 2 = bf16) = src [n, c] fp32 rounded to nearest, columns c .. pitch-1 zero: the encoder's input
 * rows in the padded pitch the 16-bit kernels read (the @auto_fp16 cast of SparseEncoder.forward, sparse_encoder.py:99). */
int bevamd_spconv_pad_cast_rows(const float* src, int n, int c, int pitch, int dtype, void* dst, void* stream);

/* Slab (staged-rows) forward for 3x3x3 SUBMANIFOLD convolutions over voxel sets whose rows are in ascending linear index
 * ((b*X + x)*Y + y)*Z + z — every set a strided convolution produced (the reference's CUDA row order, spconv_ops.h:130).
 * Same operator, epilogue and results as bevamd_spconv_conv_forward_tiled (sparse_conv_ext.indice_conv_half, all.cc:30-33 ->
 * spconv_ops.h:260-361 with subM = 1), cin == cout in {32, 64, 128}; bit-identical to it for cin <= 64, except the
 * filter-stationary 32-channel variants (codes 4xxxxxx, csrc/spconv_slab_fstat.h: v_mfma_32x32x16 reduces 16 channels per
 * instruction, the fp32 sums associate differently; equal within 2 units of the 16-bit result's last place).  Instead of gathering
 * 19-27 neighbour rows per output row through the texture path, a workgroup copies the ~9 contiguous input ranges its block
 * of rows reads (one per kernel line (kx, ky)) into LDS and feeds the MFMAs from there (csrc/spconv_slab.h).
 * Block sizes: 128 | 256 rows with raw 16-bit slots (0xFFFF = no neighbour); 64 rows = the BAKED format of the filter-
 * stationary kernels: a slot is the LDS byte offset of staged row s, (s + 1) * 64 | (((s + 1) >> 2) & 3) << 4, 0 = none; a
 * range of more than 1022 rows keeps raw slots and sets bit 30 of its row count (csrc/spconv_slab_meta.h).
 * Epilogue arithmetic of every 16-bit kernel (csrc/spconv_tile.h: finish_pair): the conv result, bias add, folded BatchNorm,
 * residual add and ReLU each round once, as the stored 16-bit tensors of the unfused reference pipeline do — fp16: packed half
 * adds (the reference's own half add), fp32 fma rounded to half, ReLU = "sign bit set -> +0" (a NaN with a clear sign bit stays
 * a NaN).
 *   bevamd_spconv_slab_block_rows(cin, variant)   rows per block of a variant (0 = default variant; returns 0 if not built)
 *   bevamd_spconv_slab_variants(cin, codes, n)     the variant codes built for cin (tuning sweeps)
 *   bevamd_spconv_slab_grid_ok(shape, block_rows)  1 if ranges on this [X, Y, Z] grid always fit the 16-bit slots
 *   bevamd_spconv_slab_{hdr,slot}_bytes            metadata sizes for m_cap rows
 *   bevamd_spconv_slab_build                       nbr [27, nbr_stride] (bevamd_spconv_neighbors, subm = 1) -> per block and
 *                                                  kernel line (first input row, row count) + 16-bit slot table; once per
 *                                                  voxel set, shared by every SubM convolution over it; status (optional
 *                                                  device int32): bit 0 set if a range overflowed (unsorted rows)
 *   bevamd_spconv_conv_forward_slab                the convolution; `image` as for the tiled entry point */
void bevamd_spconv_slab_set_profile_buffer(void* buf);  /* only meaningful in -DBEVAMD_PROFILING builds */
int bevamd_spconv_slab_ablation_mask(void);               /* compile-time ablation mask of this build (0 = shipped kernels) */
int bevamd_spconv_slab_block_rows(int cin, int variant);
int bevamd_spconv_slab_variants(int cin, int* codes, int max_n);
int bevamd_spconv_slab_grid_ok(const int* shape, int block_rows);
size_t bevamd_spconv_slab_hdr_bytes(int m_cap, int block_rows);
size_t bevamd_spconv_slab_slot_bytes(int m_cap, int block_rows);
int bevamd_spconv_slab_build(const int* nbr, int nbr_stride, int m_cap, const int* m_dev, int block_rows, void* hdr,
                             void* slots, int* status, void* stream);

/* The same metadata without the neighbour table: every row looks its 27 neighbour cells up in the voxel set's own index
 * (index_kind / index / index_n_cap as for bevamd_spconv_neighbors; indices [m, 4] = (b, x, y, z) on grid `shape`).
 * Identical hdr / slots to bevamd_spconv_neighbors(subm = 1) + bevamd_spconv_slab_build. */
int bevamd_spconv_slab_build_from_index(const int* indices, int m_cap, const int* m_dev, int batch_size, const int* shape,
                                        int index_kind, const void* index, int index_n_cap, int block_rows, void* hdr,
                                        void* slots, int* status, void* stream);

/* Sorted-key index: a voxel set whose rows are in ascending linear index (b, x, y, z) is its own lookup structure.
 * bevamd_spconv_sorted_index_build writes keys [n_cap] uint32 and the x-plane directory [batch * shape[0] + 1] int32 into
 * `index` (bevamd_spconv_sorted_index_bytes); a lookup is a binary search inside one x-plane's segment.  Replaces the dense
 * int32 grid of getIndicePair (spconv_ops.h:27-141, indice.cu.h:147-203) and the hash index for such sets.
 * status (optional int32, device): bit 1 (value 2) is set when the rows are NOT strictly ascending. */
size_t bevamd_spconv_sorted_index_bytes(int n_cap, int batch_size, const int* shape);
int bevamd_spconv_sorted_index_build(const int* indices, int n_cap, const int* n_dev, int batch_size, const int* shape,
                                     void* index, size_t index_bytes, int* status, void* stream);
/* Slab metadata (hdr / slots as bevamd_spconv_slab_build) of a 3x3x3 convolution from the sorted-key index of its INPUT set:
 * subm != 0: the submanifold convolution over the set itself (out_indices = the set; stride / padding / out_shape ignored);
 * subm == 0: the strided convolution with active outputs out_indices [m_cap, 4] on out_shape, rows in ascending linear index
 * (as bevamd_spconv_downsample emits them).  No neighbour table is built (spconv_ops.h:27-141 writes 2 x 27 x N int32). */
int bevamd_spconv_slab_build_from_sorted(const int* out_indices, int m_cap, const int* m_dev, int batch_size,
                                         const int* in_shape, const int* out_shape, const int* stride, const int* padding,
                                         int subm, const void* in_index, int in_n_cap, int block_rows, void* hdr, void* slots,
                                         int* status, void* stream);
int bevamd_spconv_conv_forward_slab(const void* features, int dtype, int feat_stride, int num_in, const void* image,
                                    const void* hdr, const void* slots, int block_rows, int num_out,
                                    const int* num_out_dev, int cin, int cout, void* out, int out_stride,
                                    const void* bias, const float* bn_scale, const float* bn_shift,
                                    const void* residual, int residual_stride, int relu, int variant, void* stream);

/* Filter-gradient half of sparse_conv_ext.indice_conv_backward_{fp32,half} (spconv_ops.h:363-456):
 *   filter_grad[k] = sum over pairs of features[i]^T @ out_grad[o]   (fp32 accumulation).
 * The input-gradient half is bevamd_spconv_conv_forward on (out_grad, prepared W^T, nbr_t). */
size_t bevamd_spconv_wgrad_workspace_bytes(int kernel_volume, int cin, int cout);
int bevamd_spconv_conv_wgrad(const void* features, const void* out_grad, int dtype, const int* nbr,
                             int nbr_stride, int num_out, int kernel_volume, int cin, int cout,
                             void* filter_grad, void* ws, size_t ws_bytes, void* stream);

/* The same filter gradient (spconv_ops.h:363-456, the filter half of indice_conv_backward_half) for a 3x3x3 SUBMANIFOLD
 * convolution over rows in ascending linear index, with the rulebook given as slab metadata (hdr / slots of
 * bevamd_spconv_slab_build* with block_rows = bevamd_spconv_wgrad_slab_block_rows(cin): the table as LDS byte addresses in the
 * order the lanes of the transposing reads want them, csrc/spconv_slab_meta.h FMT_WG64 / FMT_WG32) instead of the neighbour table: the neighbour rows of a block
 * of output rows are contiguous ranges, staged in LDS once and gathered by the transposing LDS reads (csrc/spconv_wgrad_slab.h).
 * 16-bit features; cin == cout in {16, 32, 64, 128}; pitches in elements, multiples of 8.  Deterministic. */
void bevamd_spconv_wgrad_slab_set_profile_buffer(void* buf);   /* only meaningful in -DBEVAMD_WGS_PROF experiment builds */
int bevamd_spconv_wgrad_slab_supported(int dtype, int cin, int cout);
int bevamd_spconv_wgrad_slab_block_rows(int cin);   /* block_rows code (128 | format << 16) of the metadata a cin -> cin layer reads */
size_t bevamd_spconv_wgrad_slab_workspace_bytes(int cin, int cout);
int bevamd_spconv_conv_wgrad_slab(const void* features, int feat_stride, int num_in, const void* out_grad, int og_stride,
                                  int dtype, const void* hdr, const void* slots, int block_rows, int num_out, int cin,
                                  int cout, void* filter_grad, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * device primitives (exposed for tests; used by every precompute path)
 * ------------------------------------------------------------------------- */
size_t bevamd_scan_workspace_bytes(size_t n);
int bevamd_exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total, void* ws,
                              size_t ws_bytes, void* stream);
/* the same scan in ONE launch (tiles chained by decoupled look-back, csrc/single_pass.h); `state` is scratch */
size_t bevamd_scan_single_pass_state_bytes(size_t n);
int bevamd_exclusive_scan_u32_single_pass(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total, void* state,
                                          size_t state_bytes, void* stream);
size_t bevamd_radix_sort_workspace_bytes(size_t n);
/* stable; sorts on the low nbits of the key; keys_in/vals_in are clobbered */
int bevamd_radix_sort_pairs_u32(uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out,
                                uint32_t* vals_out, size_t n, int nbits, void* ws, size_t ws_bytes,
                                void* stream);

/* nseg (<= 64) independent arrays laid end to end — counts is a HOST array of their lengths — each stably sorted on its
 * own by the launches of one sort: one digit-count pass over the keys + ONE launch per radix pass (one-sweep: tile offsets by
 * look-back; BEVAMD_SORT_ONESWEEP=0 / BEVAMD_SINGLE_PASS=0: histogram + scan + scatter launches per pass) */
size_t bevamd_radix_sort_segmented_workspace_bytes(const int* counts, int nseg);
/* host-only: the tiles of the segments (tile_begin[nseg + 1]) and how the one-sweep passes deal them to their lanes
 * (lane_begin[9]: whole segments per lane); returns the lane count (8 from 8 segments on, else 1), negative on error */
int bevamd_radix_sort_segmented_lanes(const int* counts, int nseg, unsigned* tile_begin, unsigned* lane_begin);
/* test hook (no reference counterpart): fill the LDS of every CU with `pattern`, so that a test can prove a kernel never consumes
 * LDS words it did not write (0 * stale NaN in the column backward of the fused pooling). */
int bevamd_debug_lds_poison(uint32_t pattern, void* stream);
int bevamd_radix_sort_pairs_u32_segmented(uint32_t* keys_in, uint32_t* vals_in, uint32_t* keys_out, uint32_t* vals_out,
                                          const int* counts, int nseg, int nbits, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------- *
 * training-mode BatchNorm1d over sparse feature rows (csrc/sparse_bn.hip)
 * Replaces, on the training path of the sparse blocks, the torch.nn.BatchNorm1d / nn.ReLU / residual-add modules the
 * reference applies to SparseConvTensor.features (ops/sparse_block.py:88-107; sparse_encoder.py:39: BN1d eps 1e-3,
 * momentum 0.01): batch statistics in fp32, running statistics updated like torch, every tensor of the unfused pipeline rounded
 * once.  dtype: 0 fp32 | 1 fp16 | 2 bf16 rows; statistics / parameters / parameter gradients fp32; strides in elements.
 * ------------------------------------------------------------------------- */
size_t bevamd_sparse_bn_workspace_bytes(int c);   /* zero it once; every launch leaves its ticket word zero again */
int bevamd_sparse_bn_stats(const void* x, int dtype, long long n, int c, long long stride, float eps, float momentum, float* mean,
                           float* invstd, float* running_mean, float* running_var, void* ws, size_t ws_bytes, void* stream);
int bevamd_sparse_bn_apply(const void* x, int dtype, long long n, int c, long long stride, const float* mean, const float* invstd,
                           const float* weight, const float* bias, const void* residual, long long res_stride, int relu, void* y,
                           long long y_stride, void* stream);
int bevamd_sparse_bn_backward(const void* dy, long long dy_stride, const void* y, long long y_stride, const void* x, long long stride,
                              int dtype, long long n, int c, int relu, const float* mean, const float* invstd, const float* weight,
                              float* sum_dz, float* sum_dz_xhat, void* dx, long long dx_stride, void* d_residual,
                              long long dres_stride, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BEVFUSION_AMD_H_ */
