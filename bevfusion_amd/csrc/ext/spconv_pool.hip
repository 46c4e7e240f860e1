// Sparse max pooling over a rulebook, gfx950.
//
// Replaces (reference, /root/reference/mmdet3d/ops/spconv):
//   include/spconv/pool_ops.h:25-58      indiceMaxPool          (zero-initialised output, one launch per offset)
//   include/spconv/pool_ops.h:60-97      indiceMaxPoolBackward
//   src/maxpool_cuda.cu / include/spconv/maxpool.h             maxPoolFwd* / maxPoolBwd* kernels
//   src/maxpool_cpu.cc:22-66             the CPU functors that fix the arithmetic:
//        forward   out[o][c] = in[i][c]            if out[o][c] <  in[i][c]     (out starts at 0)
//        backward  din[i][c] += dout[o][c]         if out[o][c] == in[i][c]
//
// Native formulation: the rulebook is the output-stationary table nbr[k][o] (forward) and its input-stationary
// transpose nbr_t[k][i] (backward), so each result element is produced by exactly one thread walking the K offsets in
// ascending order — no atomics, no per-offset launches, and the same accumulation order as the reference's k loop
// (an input row occurs at most once per offset), which makes the backward bit-reproducible in fp16 too.
// Purely HBM/L2-bound element work: threads run along the channels of a row so loads and stores coalesce.
#include "common.h"

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

namespace bevamd {

template <typename T> struct PoolNum;
template <> struct PoolNum<float> {
  static __device__ __forceinline__ float up(float v) { return v; }
  static __device__ __forceinline__ float down(float v) { return v; }
};
template <> struct PoolNum<__half> {
  static __device__ __forceinline__ float up(__half v) { return __half2float(v); }
  static __device__ __forceinline__ __half down(float v) { return __float2half(v); }
};
template <> struct PoolNum<__hip_bfloat16> {
  static __device__ __forceinline__ float up(__hip_bfloat16 v) { return __bfloat162float(v); }
  static __device__ __forceinline__ __hip_bfloat16 down(float v) { return __float2bfloat16(v); }
};

template <typename T>
__global__ __launch_bounds__(256) void sp_maxpool_fwd_kernel(const T* __restrict__ feat, int feat_stride,
                                                             const int* __restrict__ nbr, int nbr_stride, int num_out, int K,
                                                             int C, T* __restrict__ out, int out_stride) {
  const long long total = (long long)num_out * C;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int o = (int)(e / C), c = (int)(e - (long long)o * C);
    float best = 0.f;  // the reference starts from torch::zeros (pool_ops.h:33)
    for (int k = 0; k < K; ++k) {
      const int i = nbr[(size_t)k * nbr_stride + o];
      if (i < 0) continue;
      const float v = PoolNum<T>::up(feat[(size_t)i * feat_stride + c]);
      if (best < v) best = v;  // same comparison as maxpool_cpu.cc:36 (a NaN input never wins)
    }
    out[(size_t)o * out_stride + c] = PoolNum<T>::down(best);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void sp_maxpool_bwd_kernel(const T* __restrict__ feat, const T* __restrict__ out_feat,
                                                             const T* __restrict__ out_grad, const int* __restrict__ nbr_t,
                                                             int nbr_t_stride, int num_in, int K, int C,
                                                             T* __restrict__ in_grad) {
  const long long total = (long long)num_in * C;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int i = (int)(e / C), c = (int)(e - (long long)i * C);
    const float mine = PoolNum<T>::up(feat[e]);
    T acc = PoolNum<T>::down(0.f);
    for (int k = 0; k < K; ++k) {
      const int o = nbr_t[(size_t)k * nbr_t_stride + i];
      if (o < 0) continue;
      const size_t oe = (size_t)o * C + c;
      if (PoolNum<T>::up(out_feat[oe]) == mine)  // maxpool_cpu.cc:60; rounded to T after every add like `+=` on T
        acc = PoolNum<T>::down(PoolNum<T>::up(acc) + PoolNum<T>::up(out_grad[oe]));
    }
    in_grad[e] = acc;
  }
}

static unsigned pool_grid(long long total) {
  long long b = (total + 255) / 256;
  return (unsigned)(b < 1 ? 1 : b > 8192 ? 8192 : b);
}

}  // namespace bevamd

using namespace bevamd;

extern "C" {

int bevamd_spconv_maxpool_forward(const void* features, int dtype, int feat_stride, const int* nbr, int nbr_stride,
                                  int num_out, int kernel_volume, int channels, void* out, int out_stride, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(dtype >= 0 && dtype <= 2, "spconv_maxpool_forward: dtype %d (0 fp32, 1 fp16, 2 bf16)", dtype);
  BEVAMD_REQUIRE(num_out >= 0 && kernel_volume > 0 && channels > 0, "spconv_maxpool_forward: bad sizes");
  BEVAMD_REQUIRE(feat_stride >= channels && out_stride >= channels && nbr_stride >= num_out,
                 "spconv_maxpool_forward: a stride is smaller than the row it addresses");
  if (num_out == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(features && nbr && out, "spconv_maxpool_forward: null buffer");
  const dim3 grid(pool_grid((long long)num_out * channels)), block(256);
#define BEVAMD_POOL_FWD(T) \
  sp_maxpool_fwd_kernel<T><<<grid, block, 0, stream>>>((const T*)features, feat_stride, nbr, nbr_stride, num_out, \
                                                        kernel_volume, channels, (T*)out, out_stride)
  if (dtype == 0) BEVAMD_POOL_FWD(float);
  else if (dtype == 1) BEVAMD_POOL_FWD(__half);
  else BEVAMD_POOL_FWD(__hip_bfloat16);
#undef BEVAMD_POOL_FWD
  BEVAMD_LAUNCH_CHECK("sp_maxpool_fwd");
  return BEVAMD_OK;
}

int bevamd_spconv_maxpool_backward(const void* features, const void* out_features, const void* out_grad, int dtype,
                                   const int* nbr_t, int nbr_t_stride, int num_in, int kernel_volume, int channels,
                                   void* in_grad, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  BEVAMD_REQUIRE(dtype >= 0 && dtype <= 2, "spconv_maxpool_backward: dtype %d (0 fp32, 1 fp16, 2 bf16)", dtype);
  BEVAMD_REQUIRE(num_in >= 0 && kernel_volume > 0 && channels > 0 && nbr_t_stride >= num_in,
                 "spconv_maxpool_backward: bad sizes");
  if (num_in == 0) return BEVAMD_OK;
  BEVAMD_REQUIRE(features && out_features && out_grad && nbr_t && in_grad, "spconv_maxpool_backward: null buffer");
  const dim3 grid(pool_grid((long long)num_in * channels)), block(256);
#define BEVAMD_POOL_BWD(T) \
  sp_maxpool_bwd_kernel<T><<<grid, block, 0, stream>>>((const T*)features, (const T*)out_features, (const T*)out_grad, \
                                                        nbr_t, nbr_t_stride, num_in, kernel_volume, channels, (T*)in_grad)
  if (dtype == 0) BEVAMD_POOL_BWD(float);
  else if (dtype == 1) BEVAMD_POOL_BWD(__half);
  else BEVAMD_POOL_BWD(__hip_bfloat16);
#undef BEVAMD_POOL_BWD
  BEVAMD_LAUNCH_CHECK("sp_maxpool_bwd");
  return BEVAMD_OK;
}

}  // extern "C"
