#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void probe(int* out, int pitch_elems) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, l15 = l & 15, l4 = l >> 4;
  // lane t of a 16-lane group supplies the address of row (l4*8 + t/4), columns (t%4)*4..+3
  const short* p = lds + (l4 * 8 + (l15 >> 2)) * pitch_elems + (l15 & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  probe<<<1, 64>>>(d, 64);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (r%d,c%d)", h[l*4+j] / 64, h[l*4+j] % 64); printf("\n"); }
  return 0;
}
