// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  A = 2^-20 (subnormal in fp16) everywhere, B = 1: expect 16 * 2^-20.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a, float b) {
  f16x8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = (_Float16)a; B[i] = (_Float16)b; }
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)A[0]; }
}
int main() {
  float* d; hipMalloc(&d, 8);
  const float vals[4][2] = {{9.5367431640625e-07f, 1.f}, {1.f, 9.5367431640625e-07f}, {3.0517578125e-05f, 1.f}, {6.103515625e-05f, 1.f}};
  for (auto& v : vals) {
    k<<<1, 64>>>(d, v[0], v[1]);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("a=%g b=%g: mfma=%g (expect %g), a as f16=%g\n", v[0], v[1], h[0], 16.0 * v[0] * v[1], h[1]);
  }
  return 0;
}
