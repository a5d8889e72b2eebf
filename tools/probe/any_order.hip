// Does hipExtLaunchKernel(..., hipExtAnyOrderLaunch) let a kernel start while its predecessor in the SAME stream still runs (gfx950)?
// Two launches of a ~100 us spin kernel on 64 workgroups each: serialized = ~200 us, overlapped = ~100 us.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = wall_clock64() - t0;
}
int main() {
  unsigned long long* d;
  hipMalloc(&d, 64);
  hipStream_t st;
  hipStreamCreate(&st);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0, st);
      hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, 10000ull, d);       // 100 us at 100 MHz
      if (mode == 0) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, 10000ull, d + 1);
      else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, 10000ull, d + 1);
      else { hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, 10000ull, d + 1);
             hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, 10000ull, d + 2); }
      hipEventRecord(e1, st);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d (%s): %.1f us\n", mode, mode == 0 ? "plain, plain" : mode == 1 ? "plain, any-order" : "plain, any-order, plain", ms * 1e3);
    }
  }
  return 0;
}
