// Shared by the pybind11 shims: stream lookup and argument checks in the reference's own style.
#pragma once
#include <torch/extension.h>

#include <vector>

#include "bevfusion_amd.h"

#if __has_include(<c10/hip/HIPStream.h>)
#include <c10/hip/HIPStream.h>
static inline void* bevamd_current_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }
#else  // syntax-only builds in a container whose torch headers carry no HIP stream header
static inline void* bevamd_current_stream() { return nullptr; }
#endif

#define BEVAMD_CHECK_CUDA(x) TORCH_CHECK((x).is_cuda(), #x " must be a GPU tensor (the HIP path has no CPU fallback)")
#define BEVAMD_CHECK_CONTIG(x) TORCH_CHECK((x).is_contiguous(), #x " must be contiguous")
#define BEVAMD_CALL(expr)                      \
  do {                                         \
    int rc__ = (expr);                         \
    TORCH_CHECK(rc__ == 0, bevamd_last_error()); \
  } while (0)

static inline int bevamd_dtype_code(const at::Tensor& t) {
  if (t.scalar_type() == at::kFloat) return 0;
  if (t.scalar_type() == at::kHalf) return 1;
  if (t.scalar_type() == at::kBFloat16) return 2;
  TORCH_CHECK(false, "unsupported dtype ", t.scalar_type());
  return -1;
}
static inline std::vector<int> bevamd_ints(const std::vector<int64_t>& v) { return std::vector<int>(v.begin(), v.end()); }
