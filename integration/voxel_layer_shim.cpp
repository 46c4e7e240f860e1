// pybind11 module `voxel_layer` over libbevfusion_amd.so — replaces mmdet3d/ops/voxel/src/voxelization.cpp:6-11 for the two
// entry points on BEVFusion's hot path (voxelization.h:58-81 hard_voxelize, :83-97 dynamic_voxelize).
// The caller pre-allocates the outputs and slices them by the returned count (ops/voxel/voxelize.py:52-71); the count is a
// host int in the reference's API, so this call synchronises the stream once, like the reference (voxelization_cuda.cu:369-372).
#include "shim_common.h"

int hard_voxelize(const at::Tensor& points, at::Tensor& voxels, at::Tensor& coors, at::Tensor& num_points_per_voxel,
                  const std::vector<float> voxel_size, const std::vector<float> coors_range, const int max_points,
                  const int max_voxels, const int NDim = 3, const bool deterministic = true) {
  BEVAMD_CHECK_CUDA(points);
  BEVAMD_CHECK_CONTIG(points);
  TORCH_CHECK(points.scalar_type() == at::kFloat, "points must be float32 (BEVFusion.voxelize is @force_fp32)");
  TORCH_CHECK(NDim == 3 && voxel_size.size() == 3 && coors_range.size() == 6, "hard_voxelize: NDim must be 3");
  const int n = points.size(0), nfeat = points.size(1);
  auto opts = points.options();
  const size_t ws_bytes = bevamd_hard_voxelize_workspace_bytes(n);
  auto ws = torch::empty({(int64_t)ws_bytes}, opts.dtype(at::kByte));
  auto count_dev = torch::empty({1}, opts.dtype(at::kInt));
  int voxel_num = 0;
  BEVAMD_CALL(bevamd_hard_voxelize(points.data_ptr<float>(), voxels.data_ptr<float>(), coors.data_ptr<int>(),
                                   num_points_per_voxel.data_ptr<int>(), voxel_size.data(), coors_range.data(), max_points,
                                   max_voxels, n, nfeat, NDim, deterministic ? 1 : 0, count_dev.data_ptr<int>(), &voxel_num,
                                   ws.data_ptr(), ws_bytes, bevamd_current_stream()));
  return voxel_num;
}

void dynamic_voxelize(const at::Tensor& points, at::Tensor& coors, const std::vector<float> voxel_size,
                      const std::vector<float> coors_range, const int NDim = 3) {
  BEVAMD_CHECK_CUDA(points);
  BEVAMD_CHECK_CONTIG(points);
  TORCH_CHECK(NDim == 3 && voxel_size.size() == 3 && coors_range.size() == 6, "dynamic_voxelize: NDim must be 3");
  BEVAMD_CALL(bevamd_dynamic_voxelize(points.data_ptr<float>(), coors.data_ptr<int>(), voxel_size.data(), coors_range.data(),
                                      (int)points.size(0), (int)points.size(1), NDim, bevamd_current_stream()));
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("hard_voxelize", &hard_voxelize, "hard voxelize");
  m.def("dynamic_voxelize", &dynamic_voxelize, "dynamic voxelization");
}
