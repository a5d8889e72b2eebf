#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
  __shared__ __align__(16) unsigned short lds[64 * 64];   // [row][col] pitch 64 elements, value = row * 256 + col
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)((i / 64) * 256 + (i % 64));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, i = l & 15;
  // hypothesis: lane i of a 16-lane group points at row (i >> 2), columns 4 * (i & 3) .. + 3 of a 4 x 16 tile; group g uses columns 16 g ..
  const unsigned short* p = lds + (i >> 2) * 64 + g * 16 + (i & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short *d, h[256];
  hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" (r%d,c%2d)", h[l*4+j] >> 8, h[l*4+j] & 255); printf("\n"); }
  return 0;
}
