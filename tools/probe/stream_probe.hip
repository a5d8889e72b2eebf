// Probe: what does a (rows, 256) bf16 -> (rows, 256) bf16 streaming pass cost at the token-GEMM launch sizes, by structure?
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip && ./stream_probe [rows]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1); } } while (0)

// 1. one uint4 per thread
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i];
}
// 2. U uint4 per thread, all loads first
template <int U>
__global__ __launch_bounds__(256) void k_copy_u(const uint4* __restrict__ a, uint4* __restrict__ b, long long n) {
  const long long base = (long long)blockIdx.x * 256 * U + threadIdx.x;
  uint4 q[U];
#pragma unroll
  for (int u = 0; u < U; ++u) q[u] = base + u * 256 < n ? a[base + u * 256] : make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) b[base + u * 256] = q[u];
}
// 3. persistent: G workgroups of 512 threads, tiles of T uint4 per thread, D tiles in flight
template <int T, int D>
__global__ __launch_bounds__(512) void k_copy_p(const uint4* __restrict__ a, uint4* __restrict__ b, long long tiles) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const u4* A = (const u4*)a;
  u4* B = (u4*)b;
  u4 q[D][T];
  long long t = blockIdx.x;
  const long long step = gridDim.x;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (t + d * step < tiles)
#pragma unroll
      for (int u = 0; u < T; ++u) q[d][u] = A[((t + d * step) * T + u) * 512 + threadIdx.x];
  while (t < tiles) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (t < tiles) {
#pragma unroll
        for (int u = 0; u < T; ++u) B[(t * T + u) * 512 + threadIdx.x] = q[d][u];
        if (t + D * step < tiles)
#pragma unroll
          for (int u = 0; u < T; ++u) q[d][u] = A[((t + D * step) * T + u) * 512 + threadIdx.x];
        t += step;
      }
    }
  }
}
// 4. persistent through LDS with two barriers per tile (the token GEMM's skeleton without the product)
template <int T, int D>
__global__ __launch_bounds__(512) void k_copy_lds(const uint4* __restrict__ a, uint4* __restrict__ b, long long tiles) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  __shared__ u4 sm[2][T * 512];
  const u4* A = (const u4*)a;
  u4* B = (u4*)b;
  u4 q[D][T];
  long long t = blockIdx.x;
  const long long step = gridDim.x;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (t + d * step < tiles)
#pragma unroll
      for (int u = 0; u < T; ++u) q[d][u] = A[((t + d * step) * T + u) * 512 + threadIdx.x];
  while (t < tiles) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (t < tiles) {
#pragma unroll
        for (int u = 0; u < T; ++u) sm[0][u * 512 + threadIdx.x] = q[d][u];
        __syncthreads();
        if (t + D * step < tiles)
#pragma unroll
          for (int u = 0; u < T; ++u) q[d][u] = A[((t + D * step) * T + u) * 512 + threadIdx.x];
        u4 r[T];
#pragma unroll
        for (int u = 0; u < T; ++u) r[u] = sm[0][u * 512 + (threadIdx.x ^ 64)];
#pragma unroll
        for (int u = 0; u < T; ++u) sm[1][u * 512 + threadIdx.x] = r[u];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < T; ++u) B[(t * T + u) * 512 + threadIdx.x] = sm[1][u * 512 + (threadIdx.x ^ 64)];
        t += step;
      }
    }
  }
}
// 5. read-only and write-only
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ a, uint4* __restrict__ b, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) { uint4 q = a[i]; if (q.x == 0x12345678u && q.y == 0x9abcdef0u) b[i] = q; }
}
__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ b, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = make_uint4(1, 2, 3, 4);
}

template <class F>
static float timeit(F f, int it = 50) {
  hipEvent_t s, e;
  CK(hipEventCreate(&s)); CK(hipEventCreate(&e));
  for (int i = 0; i < 5; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(s));
  for (int i = 0; i < it; ++i) f();
  CK(hipEventRecord(e));
  CK(hipEventSynchronize(e));
  float ms;
  CK(hipEventElapsedTime(&ms, s, e));
  return ms * 1e3f / it;
}

int main(int argc, char** argv) {
  const long long rows = argc > 1 ? atoll(argv[1]) : 43008;
  for (int width = 256; width <= 512; width *= 2) {
    const long long bytes = rows * width * 2, n = bytes / 16;
    uint4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    auto rep = [&](const char* name, float us, double moved) { printf("rows %lld x %d bf16 (%.1f MB)  %-34s %7.2f us  %5.2f TB/s\n", rows, width, bytes / 1e6, name, us, moved / us / 1e6); };
    rep("copy 1 x uint4 / thread", timeit([&] { hipLaunchKernelGGL(k_copy, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n); }), 2.0 * bytes);
    rep("copy 4 x uint4 / thread", timeit([&] { hipLaunchKernelGGL(k_copy_u<4>, dim3((n + 1023) / 1024), dim3(256), 0, 0, a, b, n); }), 2.0 * bytes);
    rep("copy 8 x uint4 / thread", timeit([&] { hipLaunchKernelGGL(k_copy_u<8>, dim3((n + 2047) / 2048), dim3(256), 0, 0, a, b, n); }), 2.0 * bytes);
    rep("read only", timeit([&] { hipLaunchKernelGGL(k_read, dim3((n + 255) / 256), dim3(256), 0, 0, a, b, n); }), 1.0 * bytes);
    rep("write only", timeit([&] { hipLaunchKernelGGL(k_write, dim3((n + 255) / 256), dim3(256), 0, 0, b, n); }), 1.0 * bytes);
    for (int G = 256; G <= 1024; G *= 2) {
      char nm[64];
      snprintf(nm, 64, "persistent %d WGs, 16 KB x 2 deep", G);
      rep(nm, timeit([&] { hipLaunchKernelGGL((k_copy_p<2, 2>), dim3(G), dim3(512), 0, 0, a, b, n / 1024); }), 2.0 * bytes);
      snprintf(nm, 64, "persistent %d WGs, 16 KB x 4 deep", G);
      rep(nm, timeit([&] { hipLaunchKernelGGL((k_copy_p<2, 4>), dim3(G), dim3(512), 0, 0, a, b, n / 1024); }), 2.0 * bytes);
      snprintf(nm, 64, "persistent %d WGs, 32 KB x 2 deep", G);
      rep(nm, timeit([&] { hipLaunchKernelGGL((k_copy_p<4, 2>), dim3(G), dim3(512), 0, 0, a, b, n / 2048); }), 2.0 * bytes);
      snprintf(nm, 64, "via LDS %d WGs, 16 KB x 2 deep", G);
      rep(nm, timeit([&] { hipLaunchKernelGGL((k_copy_lds<2, 2>), dim3(G), dim3(512), 0, 0, a, b, n / 1024); }), 2.0 * bytes);
    }
    rep("empty launch", timeit([&] { hipLaunchKernelGGL(k_copy, dim3(1), dim3(256), 0, 0, a, b, 0); }), 0.0);
    CK(hipFree(a)); CK(hipFree(b));
  }
  return 0;
}
