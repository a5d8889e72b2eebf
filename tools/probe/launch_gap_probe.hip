// What a DEPENDENT kernel launch costs on the device: a chain of N small kernels on one stream, issued (a) launch by launch,
// (b) as a captured hipGraph replayed, (c) the same with a 64-byte kernarg struct.  Prints microseconds per launch (HIP events).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/launch_gap_probe tools/probe/launch_gap_probe.hip && /tmp/launch_gap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct Args { float* p; long long n; int a, b, c, d; float e, f; void* q[4]; };
__global__ void k_tiny(Args A) { if (threadIdx.x == 0 && blockIdx.x == 0) A.p[0] += 1.f; }
__global__ void k_stream(Args A) {      // a 1 MB read-modify-write: something for the cache write-back at the kernel boundary to do
  const long long i = blockIdx.x * 256ll + threadIdx.x;
  if (i < A.n) A.p[i] += 1.f;
}
static float run_direct(hipStream_t st, int N, bool big, Args A, int grid) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < N; ++i) { if (big) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, st, A); else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, A); }
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f / N;
}
static float run_graph(hipStream_t st, int N, bool big, Args A, int grid) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < N; ++i) { if (big) hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, st, A); else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, A); }
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) {
    CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return ms * 1e3f / N;
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  Args A = {}; A.n = 1 << 18; CK(hipMalloc(&A.p, A.n * 4)); CK(hipMemset(A.p, 0, A.n * 4));
  const int N = 300, grid = (int)(A.n / 256);
  printf("dependent launches on one stream, us per launch (N = %d)\n", N);
  printf("  tiny kernel, launch by launch : %.2f\n", run_direct(st, N, false, A, grid));
  printf("  tiny kernel, hipGraph replay  : %.2f\n", run_graph(st, N, false, A, grid));
  printf("  1 MB kernel, launch by launch : %.2f\n", run_direct(st, N, true, A, grid));
  printf("  1 MB kernel, hipGraph replay  : %.2f\n", run_graph(st, N, true, A, grid));
  return 0;
}
