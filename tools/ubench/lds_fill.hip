// Micro-benchmark: how fast can a CU pull an L2-resident block into LDS?
//   mode 0: LDS-DMA (buffer_load_dwordx4 ... lds), mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128, mode 2: global loads only (no LDS)
// Every workgroup (NW waves) re-reads the same `chunk_kb` KiB region `iters` times (region << L2), `depth` requests in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_fill.hip -o /tmp/lds_fill && /tmp/lds_fill
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void fill_kernel(const u32x4* __restrict__ src, int chunk_pieces, int iters, u32x4* __restrict__ sink) {
  extern __shared__ u32x4 lds[];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)chunk_pieces * 1024u, 0x00020000);
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    for (int p0 = w; p0 < chunk_pieces; p0 += nw * DEPTH) {
      u32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const int p = p0 + d * nw < chunk_pieces ? p0 + d * nw : p0;
        if (MODE == 0)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + (p % 16) * 64 + w * 1024), 16, lane * 16, p * 1024, 0, 0);
        else
          v[d] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, p * 1024, 0);
      }
      if (MODE == 1) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) lds[w * 1024 + ((p0 + d) % 16) * 64 + lane] = v[d];
      }
      if (MODE == 2) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) { acc.x ^= v[d].x; acc.y += v[d].y; }
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (MODE != 2) acc = lds[threadIdx.x];
  if (acc.x == 0x12345678u && acc.y == 77u) sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int MODE, int DEPTH>
double run(const u32x4* src, u32x4* sink, int chunk_kb, int nw, int wg_per_cu, int iters) {
  const int blocks = 256 * wg_per_cu;
  const size_t lds = (size_t)nw * 16 * 1024;
  hipFuncSetAttribute((const void*)fill_kernel<MODE, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  fill_kernel<MODE, DEPTH><<<blocks, nw * 64, lds>>>(src, chunk_kb, 2, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  fill_kernel<MODE, DEPTH><<<blocks, nw * 64, lds>>>(src, chunk_kb, iters, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = (double)blocks * iters * chunk_kb * 1024.0;
  return bytes / (ms * 1e-3) / 1e12;   // TB/s chip-wide
}

int main() {
  u32x4 *src, *sink;
  hipMalloc(&src, 4 << 20); hipMemset(src, 1, 4 << 20);
  hipMalloc(&sink, 64 << 20);
  const int iters = 400;
  printf("# TB/s chip-wide (divide by 256 CUs x ~2.4 GHz for B/clk/CU: 1 TB/s = 1.63 B/clk/CU); every workgroup re-reads the same L2-resident region\n");
  for (int chunk_kb : {8, 64, 216}) {
    for (int nw : {4, 8}) {
      for (int wpc : {1, 2, 4}) {
        if ((size_t)nw * 16 * 1024 * wpc > 160 * 1024) continue;
        printf("chunk %3d KiB, %d waves/WG, %d WG/CU:  dma d2 %6.2f  dma d4 %6.2f  dma d8 %6.2f | load+ds_write d4 %6.2f d8 %6.2f | load only d4 %6.2f d8 %6.2f\n", chunk_kb, nw, wpc,
               run<0, 2>(src, sink, chunk_kb, nw, wpc, iters), run<0, 4>(src, sink, chunk_kb, nw, wpc, iters), run<0, 8>(src, sink, chunk_kb, nw, wpc, iters),
               run<1, 4>(src, sink, chunk_kb, nw, wpc, iters), run<1, 8>(src, sink, chunk_kb, nw, wpc, iters),
               run<2, 4>(src, sink, chunk_kb, nw, wpc, iters), run<2, 8>(src, sink, chunk_kb, nw, wpc, iters));
      }
    }
  }
  return 0;
}
