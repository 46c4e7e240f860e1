// pybind11 module `bev_pool_ext` over libbevfusion_amd.so — replaces mmdet3d/ops/bev_pool/src/bev_pool_cpu.cpp:22-94.
// Argument order is the reference's: interval_LENGTHS before interval_STARTS (bev_pool_cpu.cpp:22-28).
#include "shim_common.h"

at::Tensor bev_pool_forward(const at::Tensor x, const at::Tensor geom_feats, const at::Tensor interval_lengths,
                            const at::Tensor interval_starts, int b, int d, int h, int w) {
  BEVAMD_CHECK_CUDA(x);
  BEVAMD_CHECK_CONTIG(x);
  BEVAMD_CHECK_CONTIG(geom_feats);
  const int n = x.size(0), c = x.size(1), n_intervals = interval_lengths.size(0);
  auto out = torch::empty({b, d, h, w, c}, x.options());   // every cell is written by the kernel (zeros included)
  BEVAMD_CALL(bevamd_bev_pool_forward(x.data_ptr<float>(), geom_feats.data_ptr<int>(), interval_lengths.data_ptr<int>(),
                                      interval_starts.data_ptr<int>(), out.data_ptr<float>(), n, c, n_intervals, b, d, h, w,
                                      bevamd_current_stream()));
  return out;
}

at::Tensor bev_pool_backward(const at::Tensor out_grad, const at::Tensor geom_feats, const at::Tensor interval_lengths,
                             const at::Tensor interval_starts, int b, int d, int h, int w) {
  BEVAMD_CHECK_CUDA(out_grad);
  BEVAMD_CHECK_CONTIG(out_grad);
  const int n = geom_feats.size(0), c = out_grad.size(4), n_intervals = interval_lengths.size(0);
  auto x_grad = torch::empty({n, c}, out_grad.options());
  BEVAMD_CALL(bevamd_bev_pool_backward(out_grad.data_ptr<float>(), geom_feats.data_ptr<int>(), interval_lengths.data_ptr<int>(),
                                       interval_starts.data_ptr<int>(), x_grad.data_ptr<float>(), n, c, n_intervals, b, d, h, w,
                                       /*skip_zero_fill=*/0, bevamd_current_stream()));
  return x_grad;
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("bev_pool_forward", &bev_pool_forward, "bev_pool_forward");
  m.def("bev_pool_backward", &bev_pool_backward, "bev_pool_backward");
}
