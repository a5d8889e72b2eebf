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
  GD_T_FFN = 12, GD_T_ROWS_GEMM = 13, GD_T_SLOTS = 16
};
// fp32 -> bf16, round to nearest even.  gfx950 converts in hardware, two values per instruction (v_cvt_pk_bf16_f32); the integer
// sequence the kernels carried before (compare, add 0x7FFF + lsb, shift: ~13 VALU operations per pair) was a measurable share of
// every bf16 epilogue.  Finite values and infinities round exactly as before; a NaN stays a (quiet) NaN.
typedef __bf16 gd_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float gd_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned gd_pack_bf16(float lo, float hi) {          // lo in bits 0 - 15
  const gd_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gd_bf16x2_t));
}
__device__ __forceinline__ unsigned short gd_to_bf16(float f) { return (unsigned short)(gd_pack_bf16(f, 0.f) & 0xFFFFu); }
// fp32 -> fp16 (v_cvt_pk_f16_f32: two values per instruction, round to nearest even).  The decoder's
// FORWARD products (deconvolution rows, tile convolution) take their operands in fp16 instead of bf16: same matrix-core rate
// (v_mfma_f32_32x32x16_f16), 11 instead of 8 significand bits.  What the bf16 mode's loss deviates by from the fp32 mode at full size is
// dominated by the rounding of these WEIGHTS - the same error at every site, so it does not average over the pillars the way activation
// rounding does (tools/weight_rounding_full_size.py) - and the operands there are O(1) BatchNorm / LayerNorm outputs and O(0.01 - 1)
// weights, far inside fp16's range; gradients (backward products) stay bf16.
typedef _Float16 gd_f16x2_t __attribute__((ext_vector_type(2)));
// No range clamp: a value beyond 65504 becomes inf and a NaN stays a NaN, so that a diverged run fails loudly in its loss instead of
// training on saturated rows (the operands here never come near the limit: weights and normalised activations).
__device__ __forceinline__ unsigned gd_pack_f16(float lo, float hi) {           // lo in bits 0 - 15
  const gd_f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gd_f16x2_t));
}
// ReLU + conversion for BatchNorm outputs: relu and the clamp to fp16's finite range are one v_med3_f32 (like the `h > 0 ? h : 0` it
// replaces, it turns a NaN into a finite value: a NaN batch statistic shows up in the BatchNorm's own outputs, not here)
__device__ __forceinline__ unsigned gd_pack_f16_relu(float lo, float hi) {
  const gd_f32x2_t v = {__builtin_amdgcn_fmed3f(lo, 0.f, 65504.f), __builtin_amdgcn_fmed3f(hi, 0.f, 65504.f)};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, gd_f16x2_t));
}
// 8 bf16 values -> 8 fp16 values (exact inside fp16's normal range: 8 significand bits into 11)
__device__ __forceinline__ uint4 gd_bf16x8_to_f16(const uint4& u) {
  uint4 q;
  q.x = gd_pack_f16(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u));
  q.y = gd_pack_f16(__uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
  q.z = gd_pack_f16(__uint_as_float(u.z << 16), __uint_as_float(u.z & 0xFFFF0000u));
  q.w = gd_pack_f16(__uint_as_float(u.w << 16), __uint_as_float(u.w & 0xFFFF0000u));
  return q;
}

extern int g_gd_timing_on;
void* gd_timing_begin(int slot, hipStream_t st);
void gd_timing_end(void* handle, hipStream_t st, double bytes, double flops, double side);
// bytes: operand / result rows in the compute dtype + weight images, each counted once (the figure the roofline fraction is
// computed from); side: what the launch moves besides by design (fp32 statistics rows, per-workgroup partial rows, fp32 rows at a
// stage boundary, duplicate copies such as y + pos)
struct GdTimed {
  void* h;
  hipStream_t st;
  double bytes, flops, side;
  GdTimed(int slot, hipStream_t s, double b, double f = 0.0, double sd = 0.0)
      : h(g_gd_timing_on ? gd_timing_begin(slot, s) : nullptr), st(s), bytes(b), flops(f), side(sd) {}
  ~GdTimed() {
    if (h) gd_timing_end(h, st, bytes, flops, side);
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

// Sum over the LPR (16 / 32 / 64) consecutive lanes of a lane group, result in every lane of the group, WITHOUT the LDS crossbar:
// `__shfl_xor` is a ds_bpermute (an LDS-pipe operation with its own wait) - the LayerNorm row passes of the fused layer kernels issued 56
// of them per thread, one dependent round trip each.  DPP row operations (quad permutes, half-row and row mirrors) cover the first
// four steps at VALU rate, v_permlane16_swap / v_permlane32_swap (gfx950) the last two.  Every lane of a group ends with the same
// bits (each step adds two partial sums that are the same pair in both lanes).
template <int CTRL>
__device__ __forceinline__ float gd_dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int LPR>
__device__ __forceinline__ float gd_group_sum(float v) {
  static_assert(LPR >= 16 && (LPR & (LPR - 1)) == 0, "lane group");      // > 64 (instantiated for shapes that never launch) acts as 64
  v += gd_dpp_mov<0xB1>(v);        // quad_perm [1, 0, 3, 2]: lane ^ 1
  v += gd_dpp_mov<0x4E>(v);        // quad_perm [2, 3, 0, 1]: lane ^ 2
  v += gd_dpp_mov<0x141>(v);       // row_half_mirror: the other quad of the 8-lane half row
  v += gd_dpp_mov<0x140>(v);       // row_mirror: the other half of the 16-lane row
  if (LPR >= 32) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);       // rows 0 + 1, 2 + 3
  }
  if (LPR >= 64) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);       // both halves of the wavefront
  }
  return v;
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
#ifndef GD_SCAN_STRIPED
#define GD_SCAN_STRIPED 1      // 0: experiment switch (blocked evaluation order of the look-back scans' functors)
#endif

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


// ------------------------------------------------------------------------------------------
// Single-launch device-wide exclusive scan (decoupled look-back), same load / store functors as gd_device_scan:
//   store(i, exclusive_prefix, value) for every i < n;  on_total(grand_total) once, by the last tile (optional work that
//   used to need a 1-thread "finalize" launch);  total (optional device pointer) receives the grand total.
// One workgroup = one tile of GD_SCAN_TILE elements; tiles take a ticket (atomic counter), publish their aggregate, look
// back over their predecessors' aggregates / inclusive prefixes (one wavefront, 64 predecessors per step) and publish their
// inclusive prefix.  All cross-workgroup words are accessed with RELAXED agent-scope atomics (loads / stores that go past the
// per-XCD L2) and ordered by hand: a value is stored, the store is waited for (s_waitcnt vmcnt(0)), then its flag
// is stored; a reader loads the flag and only then the value.  Agent-scope release / acquire fences instead would write back /
// invalidate the whole L2 of the XCD once per tile (measured: 66-104 us per 1.5 M-element scan instead of ~15).
// `state`: gd_scan_lb_state_bytes<T>(n) bytes that are ZERO at entry (the callers of the geometry plan clear the states of
// all their scans with ONE memset); a state must not be reused by a second scan without clearing it.
// ------------------------------------------------------------------------------------------
template <typename T>
struct GdScanWords {
  static constexpr int N = (sizeof(T) + 7) / 8;
};
template <typename T>
static inline size_t gd_scan_lb_state_bytes(long long n) {
  const size_t nb = (size_t)gd_div_up(n > 0 ? n : 1, GD_SCAN_TILE);
  return gd_align(8 + nb * 4) + 2 * gd_align(nb * GdScanWords<T>::N * 8);
}
__device__ inline void gd_scan_put(unsigned long long* w, int v) { __hip_atomic_store(w, (unsigned long long)(unsigned)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void gd_scan_put(unsigned long long* w, unsigned long long v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void gd_scan_put(unsigned long long* w, U128 v) {
  __hip_atomic_store(w, v.a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(w + 1, v.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline void gd_scan_get(const unsigned long long* w, int& v) { v = (int)(unsigned)__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void gd_scan_get(const unsigned long long* w, unsigned long long& v) { v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void gd_scan_get(const unsigned long long* w, U128& v) {
  v.a = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.b = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the value stores above (sc1: past the per-XCD L2) have been acknowledged before the flag store below is issued.  A
// workgroup-scope release fence emits NO wait on gfx950 (verified in the ISA: the two stores went out back to back and a
// reader on another XCD saw the flag before the value); an agent-scope one adds an L2 write-back the protocol does not need.
// The vmcnt wait orders stores only on targets whose stores are counted by vmcnt (gfx9 family: gfx942, gfx950).  gfx10+ counts
// them in a separate counter (vscnt): there the flag could overtake the value without any diagnostic - refuse to build.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "gd_scan_store_done relies on stores being counted by vmcnt (gfx942 / gfx950); use an agent-scope release on other targets"
#endif
__device__ inline void gd_scan_store_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
struct GdNoTotal {
  template <typename T>
  __device__ void operator()(T) const {}
};

// one tile of a look-back scan (the whole workgroup): `nb` tiles take part, every one of them must run this exactly once
// STRIPED (4-byte T): load() and store() are evaluated in striped order (element k * BLOCK + thread of the tile: the 64 lanes of a load
// instruction touch adjacent elements) and the values / prefixes change places through a padded LDS tile; the blocked order (a
// thread's 16 consecutive elements) makes every instruction of a functor with indexed reads touch 64 cache lines - the strided-conv
// output scan (nine map lookups per cell, 107 workgroups) took 166 us that way.  Same values, same prefixes.
__device__ __forceinline__ int gd_scan_pad(int x) { return x + (x >> 5); }
template <typename T, typename LoadF, typename StoreF, typename TotalF, bool STRIPED = false>
__device__ __forceinline__ void gd_scan_lb_tile(long long n, const LoadF& load, const StoreF& store, const TotalF& on_total, T* total,
                                                unsigned* ticket, unsigned* flags, unsigned long long* agg, unsigned long long* incl,
                                                const unsigned nb) {
  static_assert(!STRIPED || sizeof(T) == 4, "striped evaluation: 4-byte values");
  constexpr int W = GdScanWords<T>::N;
  __shared__ T s_xch[STRIPED ? GD_SCAN_TILE + GD_SCAN_TILE / 32 : 1];
  __shared__ T smem[GD_SCAN_BLOCK / GD_WAVE + 1];
  __shared__ unsigned s_tile;
  __shared__ T s_excl;
  if (threadIdx.x == 0) s_tile = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const unsigned tile = s_tile;
  const long long base = (long long)tile * GD_SCAN_TILE;
  T vals[GD_SCAN_ITEMS];
  T acc = gd_zero<T>();
  if constexpr (STRIPED) {
    if (base + GD_SCAN_TILE <= n) {
#pragma unroll
      for (int k = 0; k < GD_SCAN_ITEMS; ++k) vals[k] = load(base + k * GD_SCAN_BLOCK + (int)threadIdx.x);
    } else {
#pragma unroll
      for (int k = 0; k < GD_SCAN_ITEMS; ++k) {
        const long long i = base + k * GD_SCAN_BLOCK + (int)threadIdx.x;
        vals[k] = (i < n) ? load(i) : gd_zero<T>();
      }
    }
#pragma unroll
    for (int k = 0; k < GD_SCAN_ITEMS; ++k) s_xch[gd_scan_pad(k * GD_SCAN_BLOCK + (int)threadIdx.x)] = vals[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GD_SCAN_ITEMS; ++k) vals[k] = s_xch[gd_scan_pad((int)threadIdx.x * GD_SCAN_ITEMS + k)];
  } else if (base + GD_SCAN_TILE <= n) {      // full tile: no per-element bounds branch, the loads of a thread go out together
#pragma unroll
    for (int k = 0; k < GD_SCAN_ITEMS; ++k) vals[k] = load(base + (long long)threadIdx.x * GD_SCAN_ITEMS + k);
  } else {
#pragma unroll
    for (int k = 0; k < GD_SCAN_ITEMS; ++k) {
      const long long i = base + (long long)threadIdx.x * GD_SCAN_ITEMS + k;
      vals[k] = (i < n) ? load(i) : gd_zero<T>();
    }
  }
#pragma unroll
  for (int k = 0; k < GD_SCAN_ITEMS; ++k) acc = acc + vals[k];
  T tot;
  const T ex = gd_block_exclusive_scan<T, GD_SCAN_BLOCK>(acc, tot, smem);
  // ---- publish the aggregate, look back, publish the inclusive prefix (wavefront 0)
  if (threadIdx.x < GD_WAVE) {
    const int lane = threadIdx.x;
    T excl = gd_zero<T>();
    if (tile == 0) {
      if (lane == 0) {
        gd_scan_put(incl, tot);
        gd_scan_store_done();
        __hip_atomic_store(flags, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      if (lane == 0) {
        gd_scan_put(agg + (size_t)tile * W, tot);
        gd_scan_store_done();
        __hip_atomic_store(flags + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      long long hi = (long long)tile - 1;      // newest predecessor not yet accounted for
      while (true) {
        const long long j = hi - lane;         // lane 0 looks at the nearest predecessor
        unsigned f = 2u;
        if (j >= 0) {
          while ((f = __hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // nearest lane whose predecessor already has an inclusive prefix: everything beyond it is covered by that prefix
        const unsigned long long done = __ballot(f == 2u);
        const int stop = __ffsll((long long)done) - 1;          // -1: none among these 64
        T part = gd_zero<T>();
        if (j >= 0 && (stop < 0 || lane <= stop)) {
          if (f == 2u) gd_scan_get(incl + (size_t)j * W, part);
          else gd_scan_get(agg + (size_t)j * W, part);
        }
        // ordered sum over the lanes (lane 0 = nearest tile; the order only matters for the association of T = U128 / int sums,
        // which are exact anyway)
#pragma unroll
        for (int d = GD_WAVE / 2; d > 0; d >>= 1) {
          T o = gd_shfl(part, (lane + d) & (GD_WAVE - 1));
          if (lane + d < GD_WAVE) part = part + o;
        }
        part = gd_shfl(part, 0);
        excl = excl + part;
        if (stop >= 0) break;
        hi -= GD_WAVE;
      }
      if (lane == 0) {
        gd_scan_put(incl + (size_t)tile * W, excl + tot);
        gd_scan_store_done();
        __hip_atomic_store(flags + tile, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (lane == 0) {
      s_excl = excl;
      if (tile == nb - 1) {
        if (total) *total = excl + tot;
        on_total(excl + tot);
      }
    }
  }
  __syncthreads();
  T run = s_excl + ex;
  if constexpr (STRIPED) {
    // blocked prefixes -> LDS (the values are still there: every thread rewrites only the slots it read) -> striped stores
    T pre[GD_SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < GD_SCAN_ITEMS; ++k) {
      pre[k] = run;
      run = run + vals[k];
    }
#pragma unroll
    for (int k = 0; k < GD_SCAN_ITEMS; ++k) vals[k] = s_xch[gd_scan_pad(k * GD_SCAN_BLOCK + (int)threadIdx.x)];      // this thread's striped values
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GD_SCAN_ITEMS; ++k) s_xch[gd_scan_pad((int)threadIdx.x * GD_SCAN_ITEMS + k)] = pre[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GD_SCAN_ITEMS; ++k) {
      const long long i = base + k * GD_SCAN_BLOCK + (int)threadIdx.x;
      if (i < n) store(i, s_xch[gd_scan_pad(k * GD_SCAN_BLOCK + (int)threadIdx.x)], vals[k]);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < GD_SCAN_ITEMS; ++k) {
    const long long i = base + (long long)threadIdx.x * GD_SCAN_ITEMS + k;
    if (i < n) store(i, run, vals[k]);
    run = run + vals[k];
  }
}

template <typename T, typename LoadF, typename StoreF, typename TotalF>
__global__ __launch_bounds__(GD_SCAN_BLOCK) void gd_scan_lb_kernel(long long n, LoadF load, StoreF store, TotalF on_total, T* total,
                                                                  unsigned* ticket, unsigned* flags, unsigned long long* agg,
                                                                  unsigned long long* incl) {
  gd_scan_lb_tile<T, LoadF, StoreF, TotalF, GD_SCAN_STRIPED && sizeof(T) == 4>(n, load, store, on_total, total, ticket, flags, agg, incl, gridDim.x);
}

// the cross-workgroup words of a scan inside its (zeroed) state
struct GdScanState {
  unsigned* ticket;
  unsigned* flags;
  unsigned long long* agg;
  unsigned long long* incl;
  int nb;
};
template <typename T>
static inline GdScanState gd_scan_lb_state(long long n, void* state) {
  GdScanState S;
  S.nb = gd_div_up(n > 0 ? n : 1, GD_SCAN_TILE);
  char* p = (char*)state;
  S.ticket = (unsigned*)p;
  S.flags = (unsigned*)(p + 8);
  S.agg = (unsigned long long*)(p + gd_align(8 + (size_t)S.nb * 4));
  S.incl = (unsigned long long*)((char*)S.agg + gd_align((size_t)S.nb * GdScanWords<T>::N * 8));
  return S;
}
template <typename T, typename LoadF, typename StoreF, typename TotalF>
static inline int gd_device_scan_lb(long long n, LoadF load, StoreF store, TotalF on_total, T* total, void* state, hipStream_t st) {
  const GdScanState S = gd_scan_lb_state<T>(n, state);
  hipLaunchKernelGGL((gd_scan_lb_kernel<T, LoadF, StoreF, TotalF>), dim3(S.nb), dim3(GD_SCAN_BLOCK), 0, st, n, load, store, on_total, total,
                     S.ticket, S.flags, S.agg, S.incl);
  GD_LAUNCH_CHECK();
  return 0;
}

// Several independent scans of the same functor types as ONE launch (blockIdx.y = scan): the geometry plan is a chain of short
// dependent launches on its own stream, and a one- to eight-tile scan costs its launch, not its bytes (the six window scans of the
// three stages: 79 us as six launches).
template <typename T, typename LoadF, typename StoreF, typename TotalF, int NJ>
struct GdScanBatch {
  long long n[NJ];
  LoadF load[NJ];
  StoreF store[NJ];
  TotalF on_total[NJ];
  GdScanState S[NJ];
};
template <typename T, typename LoadF, typename StoreF, typename TotalF, int NJ>
__global__ __launch_bounds__(GD_SCAN_BLOCK) void gd_scan_lb_batch_kernel(GdScanBatch<T, LoadF, StoreF, TotalF, NJ> Bt) {
  const int j = blockIdx.y;
  if ((int)blockIdx.x >= Bt.S[j].nb) return;          // (before the ticket: exactly nb tiles of a scan run)
  gd_scan_lb_tile<T>(Bt.n[j], Bt.load[j], Bt.store[j], Bt.on_total[j], (T*)nullptr, Bt.S[j].ticket, Bt.S[j].flags, Bt.S[j].agg, Bt.S[j].incl,
                     (unsigned)Bt.S[j].nb);
}
template <typename T, typename LoadF, typename StoreF, typename TotalF, int NJ>
static inline int gd_device_scan_lb_batch(const GdScanBatch<T, LoadF, StoreF, TotalF, NJ>& Bt, int count, hipStream_t st) {
  if (count <= 0) return 0;
  int nb = 1;
  for (int j = 0; j < count; ++j) nb = Bt.S[j].nb > nb ? Bt.S[j].nb : nb;
  hipLaunchKernelGGL((gd_scan_lb_batch_kernel<T, LoadF, StoreF, TotalF, NJ>), dim3(nb, count), dim3(GD_SCAN_BLOCK), 0, st, Bt);
  GD_LAUNCH_CHECK();
  return 0;
}
