// pybind11 module `sparse_conv_ext` over libbevfusion_amd.so — replaces mmdet3d/ops/spconv/src/all.cc:21-51 for the 3D entry
// points BEVFusion's SparseEncoder reaches (ops/spconv/ops.py:85-211): get_indice_pairs_3d, indice_conv_{fp32,half},
// fused_indice_conv_{fp32,half}, indice_conv_backward_{fp32,half}.
//
// The native rulebook is output-stationary (nbr[k][o]); the reference-shaped (indice_pairs [K,2,N], indice_num [K]) arrays are
// derived from it (bevamd_spconv_pairs_from_nbr) and converted back on entry of the convolution calls
// (bevamd_spconv_nbr_from_pairs), so the functions compose exactly like the reference's.  Strided get_indice_pairs synchronises
// once (the returned out_indices tensor is sized by the number of active outputs), as the reference does (spconv_ops.h:130-133).
#include "shim_common.h"

static int volume(const std::vector<int>& k) { return k[0] * k[1] * k[2]; }

std::vector<at::Tensor> get_indice_pairs_3d(at::Tensor indices, int64_t batchSize, std::vector<int64_t> outSpatialShape,
                                            std::vector<int64_t> spatialShape, std::vector<int64_t> kernelSize,
                                            std::vector<int64_t> stride, std::vector<int64_t> padding,
                                            std::vector<int64_t> dilation, std::vector<int64_t> outPadding, int64_t _subM,
                                            int64_t _transpose) {
  BEVAMD_CHECK_CUDA(indices);
  BEVAMD_CHECK_CONTIG(indices);
  TORCH_CHECK(indices.scalar_type() == at::kInt && indices.size(1) == 4, "indices must be int32 [N, 4] (batch, x, y, z)");
  const auto in_shape = bevamd_ints(spatialShape), out_shape = bevamd_ints(outSpatialShape), ks = bevamd_ints(kernelSize);
  const auto st = bevamd_ints(stride), pd = bevamd_ints(padding), dl = bevamd_ints(dilation);
  const int n = indices.size(0), K = volume(ks), subm = _subM != 0, transpose = _transpose != 0;
  auto iopt = indices.options();
  const int out_cap = subm ? n : bevamd_spconv_max_outputs_ex(n, ks.data(), st.data(), dl.data(), subm, transpose);
  auto out_indices = subm ? indices : torch::empty({out_cap, 4}, iopt);
  auto nbr = torch::empty({K, out_cap > 0 ? out_cap : 1}, iopt);
  auto num_out_dev = torch::empty({1}, iopt);
  const size_t ws_bytes = bevamd_spconv_rulebook_workspace_bytes(n, (int)batchSize, out_shape.data(), subm);
  auto ws = torch::empty({(int64_t)ws_bytes}, iopt.dtype(at::kByte));
  int num_out = 0;
  void* stream = bevamd_current_stream();
  BEVAMD_CALL(bevamd_spconv_build_rulebook(indices.data_ptr<int>(), n, (int)batchSize, in_shape.data(), out_shape.data(), ks.data(),
                                           st.data(), pd.data(), dl.data(), subm, transpose, subm ? nullptr : out_indices.data_ptr<int>(),
                                           out_cap, nbr.data_ptr<int>(), (int)nbr.size(1), num_out_dev.data_ptr<int>(), &num_out,
                                           ws.data_ptr(), ws_bytes, stream));
  // reference-shaped pairs: [K, 2, N] (-1 padded) + [K]
  auto pairs = torch::empty({K, 2, n}, iopt);
  auto pair_num = torch::empty({K}, iopt);
  const size_t pws_bytes = bevamd_spconv_pairs_workspace_bytes(num_out, K);
  auto pws = torch::empty({(int64_t)pws_bytes}, iopt.dtype(at::kByte));
  BEVAMD_CALL(bevamd_spconv_pairs_from_nbr(nbr.data_ptr<int>(), (int)nbr.size(1), num_out, K, pairs.data_ptr<int>(), n,
                                           pair_num.data_ptr<int>(), pws.data_ptr(), pws_bytes, stream));
  return {subm ? indices : out_indices.slice(0, 0, num_out), pairs, pair_num};
}

// nbr[k][o] for `rows` output rows from reference-shaped pairs (inverse swaps the pair columns, spconv_ops.h:317,348)
static at::Tensor nbr_from_pairs(const at::Tensor& pairs, const at::Tensor& pair_num, int rows, int inverse) {
  const int K = pairs.size(0);
  auto nbr = torch::empty({K, rows > 0 ? rows : 1}, pairs.options());
  BEVAMD_CALL(bevamd_spconv_nbr_from_pairs(pairs.data_ptr<int>(), (int)pairs.size(2), pair_num.data_ptr<int>(), K, inverse,
                                           nbr.data_ptr<int>(), (int)nbr.size(1), bevamd_current_stream()));
  return nbr;
}

static at::Tensor conv_on_table(const at::Tensor& features, const at::Tensor& filters, const at::Tensor& nbr, int rows,
                                const at::Tensor* bias, int transpose_io) {
  const int dt = bevamd_dtype_code(features);
  const int nd = filters.dim(), cin = filters.size(nd - 2), cout = filters.size(nd - 1), K = nbr.size(0);
  const int c_out = transpose_io ? cin : cout;
  auto prepared = torch::empty({(int64_t)bevamd_spconv_prepared_filter_elems(dt, K, cin, cout, transpose_io)}, features.options());
  void* stream = bevamd_current_stream();
  BEVAMD_CALL(bevamd_spconv_prepare_filters(filters.data_ptr(), dt, K, cin, cout, transpose_io, prepared.data_ptr(), stream));
  auto out = torch::empty({rows, c_out}, features.options());
  BEVAMD_CALL(bevamd_spconv_conv_forward(features.data_ptr(), dt, prepared.data_ptr(), nbr.data_ptr<int>(), (int)nbr.size(1), rows,
                                         nullptr, K, transpose_io ? cout : cin, c_out, out.data_ptr(),
                                         bias ? bias->data_ptr() : nullptr, nullptr, nullptr, nullptr, 0, stream));
  return out;
}

template <typename T>
at::Tensor indice_conv(at::Tensor features, at::Tensor filters, at::Tensor indicePairs, at::Tensor indiceNum,
                       int64_t numActOut, int64_t _inverse, int64_t _subM) {
  BEVAMD_CHECK_CUDA(features);
  BEVAMD_CHECK_CONTIG(features);
  BEVAMD_CHECK_CONTIG(filters);
  auto nbr = nbr_from_pairs(indicePairs, indiceNum.to(indicePairs.device()), (int)numActOut, (int)_inverse);
  (void)_subM;   // an undilated SubM table already holds the identity at the centre offset (spconv_ops.h:272-276's shortcut)
  return conv_on_table(features, filters, nbr, (int)numActOut, nullptr, 0);
}

template <typename T>
at::Tensor fused_indice_conv(at::Tensor features, at::Tensor filters, at::Tensor bias, at::Tensor indicePairs,
                             at::Tensor indiceNum, int64_t numActOut, int64_t _inverse, int64_t _subM) {
  BEVAMD_CHECK_CUDA(features);
  auto nbr = nbr_from_pairs(indicePairs, indiceNum.to(indicePairs.device()), (int)numActOut, (int)_inverse);
  (void)_subM;
  return conv_on_table(features, filters, nbr, (int)numActOut, &bias, 0);
}

template <typename T>
std::vector<at::Tensor> indice_conv_backward(at::Tensor features, at::Tensor filters, at::Tensor outGrad,
                                             at::Tensor indicePairs, at::Tensor indiceNum, int64_t _inverse, int64_t _subM) {
  BEVAMD_CHECK_CUDA(features);
  BEVAMD_CHECK_CONTIG(features);
  BEVAMD_CHECK_CONTIG(outGrad);
  (void)_subM;
  const int num_in = features.size(0), num_out = outGrad.size(0);
  auto num = indiceNum.to(indicePairs.device());
  auto nbr = nbr_from_pairs(indicePairs, num, num_out, (int)_inverse);          // forward table: output <- input
  auto nbr_t = nbr_from_pairs(indicePairs, num, num_in, _inverse ? 0 : 1);     // input-stationary table
  auto in_grad = conv_on_table(outGrad, filters, nbr_t, num_in, nullptr, 1);  // dgrad = the forward kernel on W^T
  const int dt = bevamd_dtype_code(features);
  const int nd = filters.dim(), cin = filters.size(nd - 2), cout = filters.size(nd - 1), K = nbr.size(0);
  auto f_grad = torch::empty_like(filters);
  const size_t ws_bytes = bevamd_spconv_wgrad_workspace_bytes(K, cin, cout);
  auto ws = torch::empty({(int64_t)ws_bytes}, features.options().dtype(at::kByte));
  BEVAMD_CALL(bevamd_spconv_conv_wgrad(features.data_ptr(), outGrad.data_ptr(), dt, nbr.data_ptr<int>(), (int)nbr.size(1), num_out, K,
                                       cin, cout, f_grad.data_ptr(), ws.data_ptr(), ws_bytes, bevamd_current_stream()));
  return {in_grad, f_grad};
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("get_indice_pairs_3d", &get_indice_pairs_3d, "get_indice_pairs_3d");
  m.def("indice_conv_fp32", &indice_conv<float>, "indice_conv_fp32");
  m.def("indice_conv_half", &indice_conv<at::Half>, "indice_conv_half");
  m.def("fused_indice_conv_fp32", &fused_indice_conv<float>, "fused_indice_conv_fp32");
  m.def("fused_indice_conv_half", &fused_indice_conv<at::Half>, "fused_indice_conv_half");
  m.def("indice_conv_backward_fp32", &indice_conv_backward<float>, "indice_conv_backward_fp32");
  m.def("indice_conv_backward_half", &indice_conv_backward<at::Half>, "indice_conv_backward_half");
}
