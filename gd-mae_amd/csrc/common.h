// Common device helpers for the gfx950 (CDNA4, wave64) kernels of the GD-MAE hot path.
// Everything here is written for 64-wide wavefronts; there is no 32-lane path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define GD_WAVE 64

extern "C" void gd_set_error(int code, const char* file, int line, const char* msg);

#define GD_CHECK(x)                                                      \
  do {                                                                   \
    hipError_t e_ = (x);                                                 \
    if (e_ != hipSuccess) {                                              \
      gd_set_error((int)e_, __FILE__, __LINE__, hipGetErrorString(e_));  \
      return (int)e_;                                                    \
    }                                                                    \
  } while (0)
#define GD_LAUNCH_CHECK() GD_CHECK(hipGetLastError())
#define GD_REQUIRE(cond, msg)                              \
  do {                                                     \
    if (!(cond)) {                                         \
      gd_set_error(-1, __FILE__, __LINE__, msg);           \
      return -1;                                           \
    }                                                      \
  } while (0)

// ------------------------------------------------------------------------------------------
// Measurement slots (bench.py roofline leg): HIP-event brackets around the launches of a kernel family ON THE PRODUCT PATH, on
// the stream they are launched on, together with the algorithmic work of the bracketed launches.  Off by default (one load
// of a flag per launch); gdmae_kernel_timing(1) starts collecting, gdmae_kernel_timing_read sums a slot (capi.hip).
// ------------------------------------------------------------------------------------------
enum {
  GD_T_ATTN_FWD = 0, GD_T_ATTN_BWD = 1, GD_T_TOK_GEMM = 2, GD_T_DW_GROUPED = 3, GD_T_CONV_TILES = 4, GD_T_GRAD_TAPS = 5,
  GD_T_SPCONV_FWD = 6, GD_T_SPCONV_BWD = 7, GD_T_DEC_CONV_BWD = 8, GD_T_VFE = 9, GD_T_PLAN = 10, GD_T_LAYER_TAIL = 11,
  GD_T_FFN = 12, GD_T_SLOTS = 16
};
extern int g_gd_timing_on;
void* gd_timing_begin(int slot, hipStream_t st);
void gd_timing_end(void* handle, hipStream_t st, double bytes, double flops);
struct GdTimed {
  void* h;
  hipStream_t st;
  double bytes, flops;
  GdTimed(int slot, hipStream_t s, double b, double f = 0.0) : h(g_gd_timing_on ? gd_timing_begin(slot, s) : nullptr), st(s), bytes(b), flops(f) {}
  ~GdTimed() {
    if (h) gd_timing_end(h, st, bytes, flops);
  }
};

static inline int gd_div_up(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline size_t gd_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// bump allocator over a caller-provided workspace
struct GdArena {
  char* base;
  size_t cap, off;
  GdArena(void* p, size_t c) : base((char*)p), cap(c), off(0) {}
  template <typename T>
  T* take(size_t n) {
    size_t bytes = gd_align(n * sizeof(T));
    T* r = (T*)(base + off);
    off += bytes;
    return r;
  }
  bool ok() const { return off <= cap; }
};

// ------------------------------------------------------------------------------------------
// 128-bit packed counter (two u64 lanes) used to scan several counters at once
// ------------------------------------------------------------------------------------------
struct U128 {
  unsigned long long a, b;
};
__host__ __device__ inline U128 operator+(U128 x, U128 y) { return U128{x.a + y.a, x.b + y.b}; }

template <typename T>
__device__ inline T gd_zero();
template <>
__device__ inline int gd_zero<int>() { return 0; }
template <>
__device__ inline unsigned long long gd_zero<unsigned long long>() { return 0ull; }
template <>
__device__ inline U128 gd_zero<U128>() { return U128{0ull, 0ull}; }
template <>
__device__ inline float gd_zero<float>() { return 0.f; }

__device__ inline int gd_shfl_up(int v, int d) { return __shfl_up(v, d, GD_WAVE); }
__device__ inline unsigned long long gd_shfl_up(unsigned long long v, int d) { return __shfl_up(v, d, GD_WAVE); }
__device__ inline U128 gd_shfl_up(U128 v, int d) { return U128{__shfl_up(v.a, d, GD_WAVE), __shfl_up(v.b, d, GD_WAVE)}; }
__device__ inline int gd_shfl(int v, int l) { return __shfl(v, l, GD_WAVE); }
__device__ inline unsigned long long gd_shfl(unsigned long long v, int l) { return __shfl(v, l, GD_WAVE); }
__device__ inline U128 gd_shfl(U128 v, int l) { return U128{__shfl(v.a, l, GD_WAVE), __shfl(v.b, l, GD_WAVE)}; }

// inclusive scan across the 64 lanes of a wave
template <typename T>
__device__ inline T gd_wave_inclusive_scan(T v) {
  const int lane = threadIdx.x & (GD_WAVE - 1);
#pragma unroll
  for (int d = 1; d < GD_WAVE; d <<= 1) {
    T o = gd_shfl_up(v, d);
    if (lane >= d) v = v + o;
  }
  return v;
}

// Block-wide exclusive scan for blockDim.x = BLOCK (multiple of 64, <= 1024).
// Returns the exclusive prefix of `v` in thread order; `total` receives the block sum.
template <typename T, int BLOCK>
__device__ inline T gd_block_exclusive_scan(T v, T& total, T* smem /* BLOCK/64 + 1 entries */) {
  constexpr int NW = BLOCK / GD_WAVE;
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wid = threadIdx.x / GD_WAVE;
  T inc = gd_wave_inclusive_scan(v);
  if (lane == GD_WAVE - 1) smem[wid] = inc;
  __syncthreads();
  if (threadIdx.x == 0) {
    T run = gd_zero<T>();
    for (int w = 0; w < NW; ++w) {
      T t = smem[w];
      smem[w] = run;
      run = run + t;
    }
    smem[NW] = run;
  }
  __syncthreads();
  // exclusive = wave base + inclusive of the previous lane (T need not have operator-)
  T prev = gd_shfl_up(inc, 1);
  T excl = (lane == 0) ? smem[wid] : (smem[wid] + prev);
  total = smem[NW];
  __syncthreads();
  return excl;
}

__device__ inline float gd_wave_sum(float v) {
#pragma unroll
  for (int d = GD_WAVE / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, GD_WAVE);
  return v;
}
__device__ inline float gd_wave_min(float v) {
#pragma unroll
  for (int d = GD_WAVE / 2; d > 0; d >>= 1) v = fminf(v, __shfl_xor(v, d, GD_WAVE));
  return v;
}
__device__ inline float gd_wave_max(float v) {
#pragma unroll
  for (int d = GD_WAVE / 2; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d, GD_WAVE));
  return v;
}

// ------------------------------------------------------------------------------------------
// Device-wide exclusive scan:  out(i) gets  sum_{j<i} load(j)
//   LoadF:  __device__ T operator()(long long i) const
//   StoreF: __device__ void operator()(long long i, T exclusive_prefix, T value) const
//   total (optional device pointer) receives the grand total.
// n <= GD_SCAN_SINGLE_MAX runs as ONE launch of one 1024-thread workgroup (sequential tiles with a
// carry); larger n uses reduce / spine / apply (3 launches, 4096 elements per workgroup).
// ------------------------------------------------------------------------------------------
#define GD_SCAN_BLOCK 256
#define GD_SCAN_ITEMS 16
#define GD_SCAN_TILE (GD_SCAN_BLOCK * GD_SCAN_ITEMS)
#define GD_SCAN_SINGLE_MAX (1 << 16)
#define GD_SCAN_MAX_BLOCKS 65536

template <typename T, typename LoadF, typename StoreF>
__global__ __launch_bounds__(1024) void gd_scan_single_kernel(long long n, LoadF load, StoreF store, T* total) {
  __shared__ T smem[1024 / GD_WAVE + 1];
  T carry = gd_zero<T>();
  for (long long base = 0; base < n; base += 1024) {
    long long i = base + threadIdx.x;
    T v = (i < n) ? load(i) : gd_zero<T>();
    T tot;
    T ex = gd_block_exclusive_scan<T, 1024>(v, tot, smem);
    if (i < n) store(i, carry + ex, v);
    carry = carry + tot;
  }
  if (threadIdx.x == 0 && total) *total = carry;
}

template <typename T, typename LoadF>
__global__ __launch_bounds__(GD_SCAN_BLOCK) void gd_scan_reduce_kernel(long long n, LoadF load, T* block_sums) {
  __shared__ T smem[GD_SCAN_BLOCK / GD_WAVE + 1];
  const long long base = (long long)blockIdx.x * GD_SCAN_TILE;
  T acc = gd_zero<T>();
#pragma unroll
  for (int k = 0; k < GD_SCAN_ITEMS; ++k) {
    long long i = base + (long long)threadIdx.x * GD_SCAN_ITEMS + k;
    if (i < n) acc = acc + load(i);
  }
  T tot;
  gd_block_exclusive_scan<T, GD_SCAN_BLOCK>(acc, tot, smem);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

template <typename T>
__global__ __launch_bounds__(1024) void gd_scan_spine_kernel(int nb, T* block_sums, T* total) {
  __shared__ T smem[1024 / GD_WAVE + 1];
  T carry = gd_zero<T>();
  for (int base = 0; base < nb; base += 1024) {
    int i = base + threadIdx.x;
    T v = (i < nb) ? block_sums[i] : gd_zero<T>();
    T tot;
    T ex = gd_block_exclusive_scan<T, 1024>(v, tot, smem);
    if (i < nb) block_sums[i] = carry + ex;
    carry = carry + tot;
  }
  if (threadIdx.x == 0 && total) *total = carry;
}

template <typename T, typename LoadF, typename StoreF>
__global__ __launch_bounds__(GD_SCAN_BLOCK) void gd_scan_apply_kernel(long long n, LoadF load, StoreF store,
                                                                     const T* block_sums) {
  __shared__ T smem[GD_SCAN_BLOCK / GD_WAVE + 1];
  const long long base = (long long)blockIdx.x * GD_SCAN_TILE;
  T vals[GD_SCAN_ITEMS];
  T acc = gd_zero<T>();
#pragma unroll
  for (int k = 0; k < GD_SCAN_ITEMS; ++k) {
    long long i = base + (long long)threadIdx.x * GD_SCAN_ITEMS + k;
    vals[k] = (i < n) ? load(i) : gd_zero<T>();
    acc = acc + vals[k];
  }
  T tot;
  T ex = gd_block_exclusive_scan<T, GD_SCAN_BLOCK>(acc, tot, smem);
  T run = block_sums[blockIdx.x] + ex;
#pragma unroll
  for (int k = 0; k < GD_SCAN_ITEMS; ++k) {
    long long i = base + (long long)threadIdx.x * GD_SCAN_ITEMS + k;
    if (i < n) store(i, run, vals[k]);
    run = run + vals[k];
  }
}

// block_sums: workspace of >= gd_scan_ws_elems(n) elements of T
static inline size_t gd_scan_ws_elems(long long n) { return (size_t)gd_div_up(n, GD_SCAN_TILE) + 1; }

template <typename T, typename LoadF, typename StoreF>
static inline int gd_device_scan(long long n, LoadF load, StoreF store, T* total, T* block_sums, hipStream_t st) {
  if (n <= GD_SCAN_SINGLE_MAX) {
    hipLaunchKernelGGL((gd_scan_single_kernel<T, LoadF, StoreF>), dim3(1), dim3(1024), 0, st, n, load, store, total);
    GD_LAUNCH_CHECK();
    return 0;
  }
  int nb = gd_div_up(n, GD_SCAN_TILE);
  GD_REQUIRE(nb <= GD_SCAN_MAX_BLOCKS * 1024, "scan too large");
  hipLaunchKernelGGL((gd_scan_reduce_kernel<T, LoadF>), dim3(nb), dim3(GD_SCAN_BLOCK), 0, st, n, load, block_sums);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL((gd_scan_spine_kernel<T>), dim3(1), dim3(1024), 0, st, nb, block_sums, total);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL((gd_scan_apply_kernel<T, LoadF, StoreF>), dim3(nb), dim3(GD_SCAN_BLOCK), 0, st, n, load, store,
                     block_sums);
  GD_LAUNCH_CHECK();
  return 0;
}
