/* C ABI of the OPTIONAL library libbevfusion_amd_ext.so: the exports of the reference's pybind modules that lie OUTSIDE the hot
 * path of SURVEY.md §8 (sparse max pooling of sparse_conv_ext, dynamic scatter of voxel_layer).  They were built after the hot
 * path met its bar and live in their own library (bevfusion_amd/csrc/ext/), so that libbevfusion_amd.so carries only what the
 * path runs; the ext library links against it for the shared primitives (scan, sort, error string).  Same conventions as
 * bevfusion_amd.h. */
#ifndef BEVFUSION_AMD_EXT_H_
#define BEVFUSION_AMD_EXT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Dynamic scatter.  Replace voxel_layer.dynamic_point_to_voxel_forward / _backward
 *   (voxel/src/voxelization.cpp:6-11 -> voxelization.h:108-140 -> scatter_points_cuda.cu:197-330).
 * bevamd_dynamic_scatter_index: coors [num_points, ndim] int32 (ndim 1..4); rows with a negative entry are dropped.
 *   out_coors [num_points, ndim] receives the distinct rows in ascending lexicographic order (what
 *   at::unique_dim(sorted) yields), coors_map [num_points] the voxel of every point (-1 for dropped points),
 *   reduce_count [num_points] the points per voxel; order [num_points] (sorted position -> point id; points of a voxel
 *   are consecutive, ascending point id) and seg_start [num_points + 1] (first sorted position of every voxel, then
 *   the number of kept points) feed bevamd_dynamic_scatter_reduce.  Only the first *num_voxels rows of the per-voxel
 *   arrays are written.  num_voxels_dev [1] always receives the count; num_voxels_host (optional) makes the call
 *   synchronise and return it.  (The call reads the column maxima back once to size its sort keys.)
 * bevamd_dynamic_scatter_reduce: reduced [num_voxels, num_feats] = sum / mean / max (reduce_type 0 / 1 / 2, the
 *   reference's reduce_t) of feats [num_points, num_feats] fp32 over each voxel's points, in ascending point order —
 *   no float atomics, bit-reproducible (the reference's atomicAdd order is not).
 * bevamd_dynamic_scatter_backward: grad_feats [num_points, num_feats] from grad_reduced [num_voxels, num_feats]:
 *   sum: g[voxel]; mean: g[voxel] / count; max: the lowest-numbered point attaining the maximum takes g, the others 0
 *   (reduce_from_ws: int32 [num_voxels * num_feats] scratch, max only; feats / reduced may be NULL otherwise). */
size_t bevamd_dynamic_scatter_workspace_bytes(int num_points);
int bevamd_dynamic_scatter_index(const int* coors, int num_points, int ndim, int* out_coors, int* coors_map,
                                 int* reduce_count, int* order, int* seg_start, int* num_voxels_dev,
                                 int* num_voxels_host, void* ws, size_t ws_bytes, void* stream);
int bevamd_dynamic_scatter_reduce(const float* feats, int num_feats, const int* order, const int* seg_start,
                                  int num_voxels, int reduce_type, float* reduced, void* stream);
int bevamd_dynamic_scatter_backward(float* grad_feats, const float* grad_reduced, const float* feats,
                                    const float* reduced, const int* coors_map, const int* reduce_count,
                                    int num_points, int num_voxels, int num_feats, int reduce_type,
                                    int* reduce_from_ws, void* stream);


/* Sparse max pooling.  Replace sparse_conv_ext.indice_maxpool_{fp32,half} and indice_maxpool_backward_{fp32,half}
 *   (spconv/src/all.cc:39-46 -> pool_ops.h:25-97 indiceMaxPool / indiceMaxPoolBackward; arithmetic of
 *    maxpool_cpu.cc:22-66).
 * forward : out[o][c] = max(0, max over offsets k with nbr[k][o] >= 0 of features[nbr[k][o]][c]) — the reference's
 *           output starts from zeros and only takes strictly larger inputs.  features [num_in, feat_stride],
 *           out [num_out, out_stride] (strides in elements, >= channels).
 * backward: in_grad[i][c] = sum over offsets k (ascending) with o = nbr_t[k][i] >= 0 and
 *           out_features[o][c] == features[i][c] of out_grad[o][c]; all four tensors contiguous [rows, channels];
 *           nbr_t is the input-stationary table (bevamd_spconv_transpose_nbr / nbr_from_pairs with inverse = 1).
 * dtype: 0 fp32, 1 fp16, 2 bf16.  One launch each, no atomics, results bit-reproducible. */
int bevamd_spconv_maxpool_forward(const void* features, int dtype, int feat_stride, const int* nbr, int nbr_stride,
                                  int num_out, int kernel_volume, int channels, void* out, int out_stride,
                                  void* stream);
int bevamd_spconv_maxpool_backward(const void* features, const void* out_features, const void* out_grad, int dtype,
                                   const int* nbr_t, int nbr_t_stride, int num_in, int kernel_volume, int channels,
                                   void* in_grad, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BEVFUSION_AMD_EXT_H_ */
