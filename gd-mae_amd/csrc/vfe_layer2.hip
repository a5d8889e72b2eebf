// DynVFE second point layer + per-pillar maximum without its (N, 128) intermediates (reference
// pcdet/models/backbones_3d/vfe/dyn_vfe.py:107-112, network_utils.py:7-21:  h = y1 W^T (64 -> 128),
// v = relu(BatchNorm1d_train(h)), pillar feature = max of v over the pillar's points [torch_scatter.scatter_max]).
//
// 16-bit throughput mode only (y1 is the bf16 - or, round 6, fp16 - output of gdmae_vfe_point_layer_fwd; the fp32 parity mode keeps the
// op-by-op path).  F16 instantiations (gdmae_vfe_max_layer_*_f16): y1 holds fp16 values and W arrives as the fp32 master matrix, rounded
// to fp16 for the pre-activation h = y1 W^T (v_mfma_f32_32x32x16_f16) and to bf16 for the gradient product dy1 = dx W; y1^T is converted
// to bf16 when k_v2_dw stages it.  Why: the pillar maximum passes ONE point's value on - the rounding of y1 and of W does not average
// over a pillar's points - and DynVFE in bf16 was the largest term of config E's small-case loss scatter.  h (N x 128, 366 MB in bf16 for an 8-frame batch) is never stored: a 32-point tile of it is
// 16 v_mfma_f32_32x32x16_bf16 from a 4 KB tile of y1, so every kernel below recomputes it in accumulators:
//   forward   k_v2_stats  : column sums of h, h^2 (fp32 accumulators)  -> gd_bn_fold_from_partials
//             k_v2_max    : tiles in pillar (CSR) order; the tile goes through LDS and each lane walks one column
//   (see k_v2_max below)    a pillar is closed whenever the pillar id changes -> out, arg
//   backward  k_v2_gstats : gm = g [out > 0], column sums of gm, gm*h(arg) with h(arg) = (out - b) / a
//             k_v2_dy     : dx = a dh + c0 + c1 h  (dh = gm at the arg-max point) -> LDS -> dy1 = dx W  (MFMA)
//             k_v2_dw     : dW^T += y1^T dx, the dx accumulators are the B operand, y1^T comes from a transposed
//                           LDS tile; per-workgroup partials, fixed-order reduction
// Accumulator layout of v_mfma_f32_32x32x16: lane (n = lane % 32, half = lane / 32) holds column n, register r holds
// row (r & 3) + 8 (r >> 2) + 4 half.
#include "common.h"
#include "gemm.h"

int gd_bn_fold_from_partials(hipStream_t st, const float* part, int nblk, int C, double count, const float* gamma,
                             const float* beta, double eps, double momentum, float* running_mean, float* running_var,
                             long long* num_batches, double* stats, float* ab, float* mv);
int gd_partials_to_f64(hipStream_t st, const float* part, int nblk, int C2, double* out);
extern "C" int gdmae_bn_bwd_coeffs(const double* st, int n_st, const double* stats, const float* ab, const float* gamma, int C,
                                   double count, const double* tot, float* dgamma, float* dbeta, int accumulate, float* c01,
                                   void* stream);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union V2Frag {
  uint4 u;
  uint2 u2[2];
  bf16x8 v;
  f16x8 hv;
  unsigned short s[8];
};

#ifndef V2_DW_WPE
#define V2_DW_WPE 0
#endif
#if V2_DW_WPE
#define V2_DW_ATTR __attribute__((amdgpu_waves_per_eu(V2_DW_WPE, V2_DW_WPE)))
#else
#define V2_DW_ATTR
#endif
constexpr int V2_CI = 64, V2_CO = 128;
constexpr int V2_WAVES = 4;
constexpr int V2_LDW = V2_CI + 8;     // bf16 elements per row of the W tile (128 x 64) in LDS
constexpr int V2_LDX = V2_CO + 8;     // ... of the dx tile (32 x 128) and of W^T (64 x 128)
constexpr int V2_LDT = 32 + 8;        // ... of the transposed y1 tile (64 x 32)
constexpr int V2_MAX_GRID = 1024;     // rows of the statistics partials
constexpr int V2_DW_GRID = 1024;      // rows of the dW partials (32 KB each)
constexpr int V2_MAXW_GRID = 2048;    // workgroups of k_v2_max (8 workers each; their boundary slots share the dW partial area)
static_assert((size_t)V2_MAXW_GRID * 8 * 2 * 128 * 8 <= (size_t)V2_DW_GRID * 128 * 64 * 4, "boundary slots fit the partial area");

__device__ __forceinline__ int v2_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }
__device__ __forceinline__ unsigned short v2_f2bf(float f) { return gd_to_bf16(f); }
__device__ __forceinline__ f32x16 v2_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ bf16x8 v2_pack(const f32x16& a, int r0) {   // 8 accumulator registers -> bf16 fragment
  f32x8 t;
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = a[r0 + j];
  return __builtin_convertvector(t, bf16x8);
}
__device__ __forceinline__ void v2_wave_sync() {   // LDS written and read by the same wave: ordering only
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// W (128, 64) -> LDS, rows padded to V2_LDW.  F16: W is the fp32 master matrix, rounded to fp16 here; else bf16, copied
template <bool F16>
__device__ __forceinline__ void v2_load_w(const void* __restrict__ Wv, unsigned short* __restrict__ sW) {
  for (int q = threadIdx.x; q < V2_CO * (V2_CI / 8); q += V2_WAVES * 64) {
    const int row = q >> 3, ch = q & 7;
    if constexpr (F16) {
      const float* W = (const float*)Wv + row * V2_CI + 8 * ch;
      const float4 a = *reinterpret_cast<const float4*>(W), b = *reinterpret_cast<const float4*>(W + 4);
      uint4 o;
      o.x = gd_pack_f16(a.x, a.y); o.y = gd_pack_f16(a.z, a.w); o.z = gd_pack_f16(b.x, b.y); o.w = gd_pack_f16(b.z, b.w);
      *reinterpret_cast<uint4*>(sW + row * V2_LDW + 8 * ch) = o;
    } else {
      *reinterpret_cast<uint4*>(sW + row * V2_LDW + 8 * ch) = *reinterpret_cast<const uint4*>((const unsigned short*)Wv + row * V2_CI + 8 * ch);
    }
  }
}

// A fragments of h = y1 W^T for tile row `row` (this lane: columns 16 s + 8 half .. + 8 of that row); zero when !live
__device__ __forceinline__ void v2_load_y(const unsigned short* __restrict__ y1, long long row, bool live, int half,
                                          V2Frag (&ya)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    ya[s].u = *reinterpret_cast<const uint4*>(y1 + row * V2_CI + 16 * s + 8 * half);
    if (!live) ya[s].u = make_uint4(0, 0, 0, 0);
  }
}
// the same in two steps for software-pipelined loops: the loads now, the clearing of a dead row where the fragments are first needed
// (a select right behind the load waits for it there, and a prefetch "in flight behind the arithmetic" is drained before it starts)
__device__ __forceinline__ void v2_load_y_raw(const unsigned short* __restrict__ y1, long long row, int half, V2Frag (&ya)[4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) ya[s].u = *reinterpret_cast<const uint4*>(y1 + row * V2_CI + 16 * s + 8 * half);
}
__device__ __forceinline__ void v2_keep_y(V2Frag (&ya)[4], bool live) {
  const unsigned m = live ? 0xFFFFFFFFu : 0u;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    ya[s].u.x &= m; ya[s].u.y &= m; ya[s].u.z &= m; ya[s].u.w &= m;
    asm volatile("" : "+v"(ya[s].u.x), "+v"(ya[s].u.y), "+v"(ya[s].u.z), "+v"(ya[s].u.w));      // here, not where the compiler would sink it
  }
}

// pillar id and A fragments of this lane's row (n) of tile t
__device__ __forceinline__ void v2_load_tile(const unsigned short* __restrict__ y1, const int* __restrict__ rowpil, long long t,
                                             long long N, int n, int half, V2Frag (&ya)[4], int& pil) {
  const bool live = t * 32 + n < N;
  const long long row = live ? t * 32 + n : N - 1;
  pil = rowpil[row];
  v2_load_y(y1, row, live, half, ya);
}

// h[:, 32 b + n] for the tile: D[i = row][j = column] = sum_k y1[row][k] W[column][k]
template <bool F16>
__device__ __forceinline__ f32x16 v2_h(const V2Frag (&ya)[4], const unsigned short* __restrict__ sW, int n, int half, int b) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    V2Frag w;
    w.u = *reinterpret_cast<const uint4*>(sW + (32 * b + n) * V2_LDW + 16 * s + 8 * half);
    if constexpr (F16) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ya[s].hv, w.hv, acc, 0, 0, 0);
    else acc = v2_mfma(ya[s].v, w.v, acc);
  }
  return acc;
}

// ---- forward statistics --------------------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(V2_WAVES * 64) void k_v2_stats(const unsigned short* __restrict__ y1, long long N,
                                                            const void* __restrict__ W, float* __restrict__ part) {
  __shared__ unsigned short sW[V2_CO * V2_LDW];
  __shared__ float sR[V2_WAVES * 2 * V2_CO];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
  v2_load_w<F16>(W, sW);
  __syncthreads();
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const long long ntiles = (N + 31) / 32;
  for (long long t = (long long)blockIdx.x * V2_WAVES + wave; t < ntiles; t += (long long)gridDim.x * V2_WAVES) {
    const long long row = t * 32 + n;
    V2Frag ya[4];
    v2_load_y(y1, row < N ? row : N - 1, row < N, half, ya);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const f32x16 h = v2_h<F16>(ya, sW, n, half, b);   // rows past N are exactly zero
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s1[b] += h[r];
        s2[b] = fmaf(h[r], h[r], s2[b]);
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    s1[b] += __shfl_xor(s1[b], 32, 64);
    s2[b] += __shfl_xor(s2[b], 32, 64);
    if (half == 0) {
      sR[wave * 2 * V2_CO + 32 * b + n] = s1[b];
      sR[wave * 2 * V2_CO + V2_CO + 32 * b + n] = s2[b];
    }
  }
  __syncthreads();
  {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < V2_WAVES; ++w) a += sR[w * 2 * V2_CO + threadIdx.x];
    part[(long long)blockIdx.x * 2 * V2_CO + threadIdx.x] = a;
  }
}

// ---- forward maximum -----------------------------------------------------------------------------------------
// Every half-wave is a worker with its own contiguous range of `per` rows - the SAME number of rows for every worker, wherever the
// pillar boundaries fall (ranges snapped to pillar starts gave a crowded pillar - 1 000 points next to the sensor against 11 on
// average - to one half-wave as a serial chain of 68 tiles: 118 of the launch's 174 us did not scale with the batch).  A 32-row MFMA
// tile holds 16 rows of each worker, permuted so that a lane's 16 accumulator registers are 16 CONSECUTIVE rows of its worker:
// MFMA row i = (r & 3) + 8 (r >> 2) + 4 h  <->  worker h, local row r.  Each lane then walks its registers in row order, closing a
// pillar (three coalesced 128-byte stores per column block) whenever the pillar id changes - no LDS, no atomics.  A pillar that lies
// inside the range goes straight to out / arg; the piece of a pillar that began before the range goes to the worker's boundary slot
// 0, the piece of one that continues behind it to slot 1 (a range inside one pillar: slot 0), and k_v2_max_fix joins the pieces of
// such a pillar in row order.  arg = row of the maximum (strict >: the first row, i.e. the lowest point id, wins ties - (max value,
// min row) is associative, so joining pieces in order gives the sequential walk's answer bit for bit).
template <bool F16>
__global__ __launch_bounds__(V2_WAVES * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_v2_max(const unsigned short* __restrict__ y1, long long N,
                                                          const void* __restrict__ W, const int* __restrict__ pt_off,
                                                          const int* __restrict__ rowpil, int M, const float* __restrict__ ab,
                                                          float* __restrict__ out, int* __restrict__ arg, long long per,
                                                          float* __restrict__ pbest, int* __restrict__ parg) {
  __shared__ unsigned short sW[V2_CO * V2_LDW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
  v2_load_w<F16>(W, sW);
  __syncthreads();
  // this worker's share of the rows; head: its first pillar began before the range, tail: its last pillar continues behind it
  const int wk = (blockIdx.x * V2_WAVES + wave) * 2 + half;      // (int, and the slot offsets recomputed where used: 162 registers = 3 waves per SIMD)
  const long long qa = (long long)wk * per < N ? (long long)wk * per : N, qb = (long long)(wk + 1) * per < N ? (long long)(wk + 1) * per : N;
  const int q0 = (int)qa, q1 = (int)qb;
  bool hp, tail;
  {
    const long long ia = qa > 0 ? qa - 1 : 0, ib = qa < N ? qa : N - 1, ic = qb > 0 ? qb - 1 : 0, id = qb < N ? qb : N - 1;
    const int a0 = rowpil[ia], a1 = rowpil[ib], b0 = rowpil[ic], b1 = rowpil[id];     // four independent loads
    hp = qa > 0 && qa < qb && a0 == a1;
    tail = qb < N && qa < qb && b0 == b1;
  }
  // the tile row this lane loads as MFMA row n: worker (n >> 2) & 1, local row (n & 3) + 4 (n >> 3)
  const int ld_wk = (n >> 2) & 1, ld_r = (n & 3) + 4 * (n >> 3);
  const int q0o = __shfl(q0, lane ^ 32, 64), q1o = __shfl(q1, lane ^ 32, 64);
  const int lq0 = ld_wk == half ? q0 : q0o, lq1 = ld_wk == half ? q1 : q1o;   // range of the worker whose row this lane loads
  const int len = q1 - q0, leno = q1o - q0o;
  const int iters = ((len > leno ? len : leno) + 15) / 16;                      // wave-uniform
  float ca[4], cb[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    ca[b] = ab[32 * b + n];
    cb[b] = ab[V2_CO + 32 * b + n];
  }
  float best[4];
  int bi[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) { best[b] = -1.f; bi[b] = 0; }
  int cur = -1;

  V2Frag ya[4], yn[4];
  int pl = 0, pln = 0;                            // pillar of the row this lane loads
  {
    const int row = lq0 + ld_r;
    const bool live = row < lq1;
    const int rc = live ? row : (lq1 > 0 ? lq1 - 1 : 0);
    pl = rowpil[rc];
    v2_load_y(y1, rc, live, half, ya);
  }
  for (int it = 0; it < iters; ++it) {
    const int qt = q0 + 16 * it;                 // first row of this worker's 16-row slice
    const int nrows = q1 - qt < 16 ? (q1 - qt > 0 ? q1 - qt : 0) : 16;
    bool live_n;
    {                                            // next tile: in flight during this tile's products (unconditional: past the
      const int row = lq0 + 16 * (it + 1) + ld_r;      // worker's last slice the clamped row is re-read and cleared)
      live_n = row < lq1;
      const int rc = live_n ? row : (lq1 > 0 ? lq1 - 1 : 0);
      pln = rowpil[rc];
      v2_load_y_raw(y1, rc, half, yn);
    }
    __builtin_amdgcn_sched_barrier(0);           // the loads go out first
    int prow[16];                                // pillar of each of this worker's rows (uniform within the half-wave)
#pragma unroll
    for (int r = 0; r < 16; ++r) prow[r] = __shfl(pl, v2_row(r, half), 64);   // the lane that loaded this worker's row r
    // st[r]: row r opens a new pillar.  The walk itself is branch-free (selects); only the stores that close a pillar
    // sit behind a (rarely taken) branch.
    bool st[16];
    st[0] = nrows > 0 && prow[0] != cur;
#pragma unroll
    for (int r = 1; r < 16; ++r) st[r] = r < nrows && prow[r] != prow[r - 1];
    const int cur_in = cur;
#pragma unroll
    for (int r = 0; r < 16; ++r) cur = r < nrows ? prow[r] : cur;
    f32x16 h[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) h[b] = v2_h<F16>(ya, sW, n, half, b);
    // the next tile is waited for HERE, before this tile's stores go out: loads and stores retire out of order with respect to each
    // other, so once stores are pending any wait for a load is a full drain of both
    v2_keep_y(yn, live_n);
    asm volatile("" : "+v"(pln));
    if (st[0] && cur_in >= 0) {                  // the pillar carried over from the previous tile ends here
      float* const ob = hp ? pbest : out;
      int* const oa = hp ? parg : arg;
      const long long o0 = (long long)(hp ? wk * 2 : cur_in) * V2_CO;
      hp = false;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        ob[o0 + 32 * b + n] = best[b];
        oa[o0 + 32 * b + n] = bi[b];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float bin = st[r] ? -1.f : best[b];
        const float v = fmaxf(fmaf(ca[b], h[b][r], cb[b]), 0.f);
        const bool up = r < nrows && v > bin;    // strict: the first row of a pillar (lowest point id) wins ties
        best[b] = up ? v : bin;
        bi[b] = up ? qt + r : bi[b];
      }
      if (r < 15 && st[r < 15 ? r + 1 : 15]) {   // row r closes its pillar: one branch for the four column blocks
        float* const ob = hp ? pbest : out;
        int* const oa = hp ? parg : arg;
        const long long o0 = (long long)(hp ? wk * 2 : prow[r]) * V2_CO;
        hp = false;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          ob[o0 + 32 * b + n] = best[b];
          oa[o0 + 32 * b + n] = bi[b];
        }
      }
    }
    pl = pln;
#pragma unroll
    for (int s = 0; s < 4; ++s) ya[s] = yn[s];
  }
  if (cur >= 0) {                                // the range's last pillar: whole (out), first piece (slot 1) or a later piece (slot 0)
    float* const ob = hp || tail ? pbest : out;
    int* const oa = hp || tail ? parg : arg;
    const long long o0 = (long long)(hp ? wk * 2 : (tail ? wk * 2 + 1 : cur)) * V2_CO;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      ob[o0 + 32 * b + n] = best[b];
      oa[o0 + 32 * b + n] = bi[b];
    }
  }
}

// joins the pieces of every pillar that crosses a range boundary of k_v2_max: one workgroup per worker; the worker in whose range
// the pillar BEGINS (its slot 1) walks the following workers' slot 0 pieces in row order until the pillar ends.  Chains are one
// piece long except for crowded pillars (points / per pieces).
__global__ __launch_bounds__(V2_CO) void k_v2_max_fix(const int* __restrict__ rowpil, long long N, long long per,
                                                      const float* __restrict__ pbest, const int* __restrict__ parg,
                                                      float* __restrict__ out, int* __restrict__ arg) {
  const long long w = blockIdx.x;
  const int c = threadIdx.x;
  const long long qa = w * per < N ? w * per : N, qb = (w + 1) * per < N ? (w + 1) * per : N;
  if (qa >= qb || qb >= N) return;
  const int P = rowpil[qb - 1];
  if (rowpil[qb] != P) return;                        // the last pillar ends with the range
  if (qa > 0 && rowpil[qa - 1] == P) return;          // it began before this range: the chain belongs to an earlier worker
  float bv = pbest[(w * 2 + 1) * V2_CO + c];
  int bi = parg[(w * 2 + 1) * V2_CO + c];
  for (long long v = w + 1;; ++v) {
    const float pv = pbest[(v * 2) * V2_CO + c];
    const int pi = parg[(v * 2) * V2_CO + c];
    const bool up = pv > bv;                           // strict: the earlier piece keeps ties
    bv = up ? pv : bv;
    bi = up ? pi : bi;
    const long long vb = (v + 1) * per < N ? (v + 1) * per : N;
    if (vb >= N || rowpil[vb] != P) break;
  }
  out[(long long)P * V2_CO + c] = bv;
  arg[(long long)P * V2_CO + c] = bi;
}

// ---- backward: masked gradient and its column sums ---------------------------------------------------------------
// the pre-activation at the arg-max row is recovered from the stored maximum, h = (out - b) / a (out > 0 there, so the
// ReLU was the identity; a = 0 only for gamma = 0, where the sum it feeds is multiplied by gamma anyway)
__global__ __launch_bounds__(256) void k_v2_gstats(const float* __restrict__ out, const float* __restrict__ ab,
                                                   const float* __restrict__ g, long long M, float* __restrict__ gm,
                                                   float* __restrict__ part) {
  // round 5: four channels (16 bytes) per thread, eight pillar rows per workgroup pass, four rows of each array in flight per thread
  // (4-byte accesses left this 200 MB stream at SQ_WAIT_ANY 0.94 / 3.4 TB/s)
  __shared__ float sR[8 * 2 * V2_CO];
  const int tr = threadIdx.x >> 5, c4 = (threadIdx.x & 31) * 4;
  const long long chunk = (M + gridDim.x - 1) / gridDim.x;
  const long long r0 = blockIdx.x * chunk, r1 = r0 + chunk < M ? r0 + chunk : M;
  float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  const float4 a4 = *reinterpret_cast<const float4*>(ab + c4), b4 = *reinterpret_cast<const float4*>(ab + V2_CO + c4);
  const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
  float ia[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) ia[e] = av[e] != 0.f ? 1.f / av[e] : 0.f;
  for (long long p0 = r0 + tr; p0 < r1; p0 += 32) {
    float4 ov[4], gv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                    // unconditional (clamped row), masked below
      const long long p = p0 + 8 * u < r1 ? p0 + 8 * u : r1 - 1;
      ov[u] = *reinterpret_cast<const float4*>(out + p * V2_CO + c4);
      gv[u] = *reinterpret_cast<const float4*>(g + p * V2_CO + c4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (p0 + 8 * u < r1) {
        const float o[4] = {ov[u].x, ov[u].y, ov[u].z, ov[u].w}, gg[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
        float m[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          m[e] = o[e] > 0.f ? gg[e] : 0.f;
          s0[e] += m[e];
          s1[e] = fmaf(m[e], (o[e] - bv[e]) * ia[e], s1[e]);
        }
        *reinterpret_cast<float4*>(gm + (p0 + 8 * u) * V2_CO + c4) = make_float4(m[0], m[1], m[2], m[3]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    sR[(tr * 2 + 0) * V2_CO + c4 + e] = s0[e];
    sR[(tr * 2 + 1) * V2_CO + c4 + e] = s1[e];
  }
  __syncthreads();
  {
    const int q = threadIdx.x;   // 256 = 2 * V2_CO: (statistic, channel)
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) a += sR[t * 2 * V2_CO + q];
    part[(long long)blockIdx.x * 2 * V2_CO + q] = a;
  }
}

// dx of one 32-column block of the tile: dx[row][c] = a dh + c0 + c1 h,  dh = gm[pillar][c] at the arg-max row
struct V2Coef {
  float a, c0, c1;
};
__device__ __forceinline__ f32x16 v2_dx(const f32x16& h, const V2Coef& k, const int* __restrict__ arg,
                                        const float* __restrict__ gm, int c, const int (&prow)[16], int q0, int half) {
  f32x16 dx;
  float gv[16];
  int am[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {          // 32 independent loads; the rows of a pillar are adjacent and share cache lines
    const long long o = (long long)prow[r] * V2_CO + c;
    am[r] = arg[o];
    gv[r] = gm[o];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float dh = am[r] == q0 + v2_row(r, half) ? gv[r] : 0.f;
    dx[r] = fmaf(k.c1, h[r], fmaf(k.a, dh, k.c0));
  }
  return dx;
}

// ---- backward: dy1 = dx W ------------------------------------------------------------------------------------------
template <bool F16>
__global__ __launch_bounds__(V2_WAVES * 64) void k_v2_dy(const unsigned short* __restrict__ y1, long long N,
                                                         const void* __restrict__ W, const int* __restrict__ rowpil,
                                                         const float* __restrict__ ab, const float* __restrict__ c01,
                                                         const int* __restrict__ arg, const float* __restrict__ gm,
                                                         unsigned* __restrict__ dy1) {
  __shared__ unsigned short sW[V2_CO * V2_LDW];
  __shared__ unsigned short sWT[V2_CI * V2_LDX];            // W^T: row j (input channel), column c
  __shared__ unsigned short sX[V2_WAVES * 32 * V2_LDX];     // per wave: the bf16 dx tile, row-major
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
  v2_load_w<F16>(W, sW);
  for (int q = threadIdx.x; q < V2_CO * V2_CI; q += V2_WAVES * 64) {      // W^T in bf16: operand of the gradient product
    const int c = q >> 6, j = q & 63;
    sWT[j * V2_LDX + c] = F16 ? v2_f2bf(((const float*)W)[q]) : ((const unsigned short*)W)[q];
  }
  __syncthreads();
  unsigned short* sx = sX + wave * 32 * V2_LDX;
  V2Coef k[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int c = 32 * b + n;
    k[b].a = ab[c]; k[b].c0 = c01[c]; k[b].c1 = c01[V2_CO + c];
  }
  const long long ntiles = (N + 31) / 32;
  const long long stride = (long long)gridDim.x * V2_WAVES;
  long long t = (long long)blockIdx.x * V2_WAVES + wave;
  V2Frag ya[4], yn[4];
  int pil = 0, piln = 0;
  if (t < ntiles) v2_load_tile(y1, rowpil, t, N, n, half, ya, pil);
  for (; t < ntiles; t += stride) {
    const long long q = t * 32;
    const int nrows = (int)(N - q < 32 ? N - q : 32);
    if (t + stride < ntiles) v2_load_tile(y1, rowpil, t + stride, N, n, half, yn, piln);   // in flight during this tile
    int prow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) prow[r] = __shfl(pil, v2_row(r, half), 64);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const f32x16 h = v2_h<F16>(ya, sW, n, half, b);
      const f32x16 dx = v2_dx(h, k[b], arg, gm, 32 * b + n, prow, (int)q, half);
#pragma unroll
      for (int r = 0; r < 16; ++r) sx[v2_row(r, half) * V2_LDX + 32 * b + n] = v2_f2bf(dx[r]);
      __builtin_amdgcn_sched_barrier(0);      // one block at a time: keeps the register count at 3 waves per SIMD
    }
    v2_wave_sync();
    // D[i = row][j = input channel 2 n + blk] = sum_c dx[row][c] W[c][j]
    f32x16 d[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) d[0][r] = d[1][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      V2Frag a, b0, b1;
      a.u = *reinterpret_cast<const uint4*>(sx + n * V2_LDX + 16 * s + 8 * half);
      b0.u = *reinterpret_cast<const uint4*>(sWT + (2 * n) * V2_LDX + 16 * s + 8 * half);
      b1.u = *reinterpret_cast<const uint4*>(sWT + (2 * n + 1) * V2_LDX + 16 * s + 8 * half);
      d[0] = v2_mfma(a.v, b0.v, d[0]);
      d[1] = v2_mfma(a.v, b1.v, d[1]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (v2_row(r, half) < nrows)
        dy1[(q + v2_row(r, half)) * (V2_CI / 2) + n] = (unsigned)v2_f2bf(d[0][r]) | ((unsigned)v2_f2bf(d[1][r]) << 16);
    }
    v2_wave_sync();
    pil = piln;
#pragma unroll
    for (int s = 0; s < 4; ++s) ya[s] = yn[s];
  }
}

// ---- backward: dW = dx^T y1 ----------------------------------------------------------------------------------------
// the four waves of a workgroup share each tile, wave w owns the 32 columns [32 w, 32 w + 32) of dx / rows of dW
template <bool F16>
__global__ __launch_bounds__(V2_WAVES * 64) V2_DW_ATTR void k_v2_dw(const unsigned short* __restrict__ y1, long long N,
                                                         const void* __restrict__ W, const int* __restrict__ rowpil,
                                                         const float* __restrict__ ab, const float* __restrict__ c01,
                                                         const int* __restrict__ arg, const float* __restrict__ gm,
                                                         float* __restrict__ part) {
  __shared__ unsigned short sW[V2_CO * V2_LDW];
  __shared__ unsigned short sT[V2_WAVES * V2_CI * V2_LDT];   // per wave: y1 tile transposed, row j, column = tile row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 31, half = lane >> 5;
  v2_load_w<F16>(W, sW);
  __syncthreads();
  unsigned short* st = sT + wave * V2_CI * V2_LDT;
  const int b = wave, c = 32 * b + n;
  V2Coef k;
  k.a = ab[c]; k.c0 = c01[c]; k.c1 = c01[V2_CO + c];
  f32x16 acc[2];      // [input-channel block i2]: lane = column c, register = channel 32 i2 + row(r)
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
  const long long ntiles = (N + 31) / 32;
  long long t = blockIdx.x;
  V2Frag ya[4], yn[4];
  int pil = 0, piln = 0;
  if (t < ntiles) v2_load_tile(y1, rowpil, t, N, n, half, ya, pil);   // rows past the end are zero: nothing added to dW
  for (; t < ntiles; t += gridDim.x) {
    const long long q = t * 32;
    if (t + gridDim.x < ntiles) v2_load_tile(y1, rowpil, t + gridDim.x, N, n, half, yn, piln);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j)      // (F16: the transposed tile is the bf16 operand of the gradient product with dx)
        st[(16 * s + 8 * half + j) * V2_LDT + n] = F16 ? v2_f2bf((float)ya[s].hv[j]) : ya[s].s[j];
    int prow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) prow[r] = __shfl(pil, v2_row(r, half), 64);
    const f32x16 h = v2_h<F16>(ya, sW, n, half, b);
    const f32x16 dx = v2_dx(h, k, arg, gm, c, prow, (int)q, half);
    v2_wave_sync();
    // A fragments of D[i = channel][j = column] = sum_rows y1[row][channel] dx[row][column]: channel 32 i2 + n, the 8 rows
    // base + 4 half + {0..3}, base + 8 + 4 half + {0..3} - the rows of accumulator registers [base / 2, base / 2 + 8)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 bx = v2_pack(dx, 8 * kk);
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        V2Frag at;
        at.u2[0] = *reinterpret_cast<const uint2*>(st + (32 * i2 + n) * V2_LDT + 16 * kk + 4 * half);
        at.u2[1] = *reinterpret_cast<const uint2*>(st + (32 * i2 + n) * V2_LDT + 16 * kk + 8 + 4 * half);
        acc[i2] = v2_mfma(at.v, bx, acc[i2]);
      }
    }
    v2_wave_sync();
    pil = piln;
#pragma unroll
    for (int s = 0; s < 4; ++s) ya[s] = yn[s];
  }
#pragma unroll
  for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      part[(long long)blockIdx.x * V2_CO * V2_CI + c * V2_CI + 32 * i2 + v2_row(r, half)] = acc[i2][r];
}

template <typename K>
int v2_resident_blocks(K kernel, int slot, int cap) {
  static int cache[8] = {0};
  if (cache[slot] == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, V2_WAVES * 64, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
    const long long b = (long long)per_cu * prop.multiProcessorCount;
    cache[slot] = (int)(b > cap ? cap : b);
  }
  return cache[slot];
}

inline int v2_grid(long long N, int cap) {
  const long long blocks = ((N + 31) / 32 + V2_WAVES - 1) / V2_WAVES;
  return (int)(blocks < 1 ? 1 : (blocks > cap ? cap : blocks));
}

struct V2Ws {
  float* part;
  double* sums;
  float* c01;
};
inline size_t v2_ws_bytes() {
  return gd_align((size_t)V2_DW_GRID * V2_CO * V2_CI * sizeof(float)) + gd_align(2 * V2_CO * sizeof(double)) +
         gd_align(2 * V2_CO * sizeof(float));
}
inline V2Ws v2_ws(void* workspace) {
  char* p = (char*)workspace;
  V2Ws w;
  w.part = (float*)p;
  p += gd_align((size_t)V2_DW_GRID * V2_CO * V2_CI * sizeof(float));
  w.sums = (double*)p;
  p += gd_align(2 * V2_CO * sizeof(double));
  w.c01 = (float*)p;
  return w;
}

}  // namespace

extern "C" size_t gdmae_vfe_max_layer_workspace_bytes(void) { return v2_ws_bytes(); }

// y1 (N, 64) bf16 with its rows in pillar-major order (gdmae_pillar_major_rows + gdmae_vfe_point_layer_fwd),
// row_pillar (N) = pillar of each row, pillar_pt_off (M + 1) = first row of each pillar; W (128, 64) bf16.
// out (M, 128) fp32 = max over the pillar of relu(BatchNorm1d_train(y1 W^T)), arg = row of the maximum (first row on
// ties = lowest point id); stats / ab / mv as gdmae_bn_fold.
// *_f16: y1 holds fp16 values (gdmae_vfe_point_layer_fwd with out_bf16 = 2) and W is the fp32 (128, 64) master matrix.
template <bool F16>
static int v2_fwd(const void* y1, long long N, const void* W, const int* pillar_pt_off, const int* row_pillar, int M, const float* gamma,
                  const float* beta, double eps, double momentum, float* running_mean, float* running_var, long long* num_batches,
                  double* stats, float* ab, float* mv, float* out, int* arg, void* workspace, void* stream) {
  GD_REQUIRE(N > 0 && M > 0, "vfe max layer: no points");
  hipStream_t st = (hipStream_t)stream;
  const V2Ws ws = v2_ws(workspace);
  const int g1 = v2_grid(N, v2_resident_blocks(k_v2_stats<F16>, F16 ? 4 : 0, V2_MAX_GRID));
  hipLaunchKernelGGL(k_v2_stats<F16>, dim3(g1), dim3(V2_WAVES * 64), 0, st, (const unsigned short*)y1, N, W, ws.part);
  GD_LAUNCH_CHECK();
  int rc = gd_bn_fold_from_partials(st, ws.part, g1, V2_CO, (double)N, gamma, beta, eps, momentum, running_mean, running_var,
                                    num_batches, stats, ab, mv);
  if (rc) return rc;
  // boundary pieces of k_v2_max: 2 slots x 128 columns x (value, row) per worker, in the partial area (the statistics partials have
  // been consumed by the fold above): V2_MAXW_GRID workgroups x 8 workers x 2 KB = the area's 32 MB
  const int g2 = v2_grid(N, v2_resident_blocks(k_v2_max<F16>, F16 ? 5 : 1, V2_MAXW_GRID));
  const long long nwk = (long long)g2 * V2_WAVES * 2, per = (N + nwk - 1) / nwk;
  float* const pbest = ws.part;
  int* const parg = (int*)(ws.part + nwk * 2 * V2_CO);
  hipLaunchKernelGGL(k_v2_max<F16>, dim3(g2), dim3(V2_WAVES * 64), 0, st, (const unsigned short*)y1, N, W, pillar_pt_off, row_pillar, M,
                     (const float*)ab, out, arg, per, pbest, parg);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_v2_max_fix, dim3((unsigned)nwk), dim3(V2_CO), 0, st, row_pillar, N, per, (const float*)pbest, (const int*)parg,
                     out, arg);
  GD_LAUNCH_CHECK();
  return 0;
}
extern "C" int gdmae_vfe_max_layer_fwd(const void* y1, long long N, const void* W, const int* pillar_pt_off,
                                       const int* row_pillar, int M, const float* gamma,
                                       const float* beta, double eps, double momentum, float* running_mean,
                                       float* running_var, long long* num_batches, double* stats, float* ab, float* mv,
                                       float* out, int* arg, void* workspace, void* stream) {
  return v2_fwd<false>(y1, N, W, pillar_pt_off, row_pillar, M, gamma, beta, eps, momentum, running_mean, running_var, num_batches, stats, ab,
                       mv, out, arg, workspace, stream);
}
extern "C" int gdmae_vfe_max_layer_fwd_f16(const void* y1, long long N, const float* W, const int* pillar_pt_off,
                                           const int* row_pillar, int M, const float* gamma,
                                           const float* beta, double eps, double momentum, float* running_mean,
                                           float* running_var, long long* num_batches, double* stats, float* ab, float* mv,
                                           float* out, int* arg, void* workspace, void* stream) {
  return v2_fwd<true>(y1, N, W, pillar_pt_off, row_pillar, M, gamma, beta, eps, momentum, running_mean, running_var, num_batches, stats, ab,
                      mv, out, arg, workspace, stream);
}

// g (M, 128) fp32: gradient of out.  gm: scratch, M * 128 floats (the masked gradient).  dy1 (N, 64) bf16 is written; dgamma / dbeta / dW (128, 64)
// fp32 are written, or accumulated into when `accumulate`.
template <bool F16>
static int v2_bwd(const void* y1, long long N, const void* W, const int* row_pillar, int M, const float* gamma, const double* stats,
                  const float* ab, const float* out, const int* arg, const float* g, void* gm, void* dy1, float* dgamma, float* dbeta,
                  float* dW, int accumulate, void* workspace, void* stream) {
  GD_REQUIRE(N > 0 && M > 0, "vfe max layer: no points");
  hipStream_t st = (hipStream_t)stream;
  const V2Ws ws = v2_ws(workspace);
  int g0 = (int)(M / 64 > V2_MAX_GRID ? V2_MAX_GRID : (M / 64 > 0 ? M / 64 : 1));
  hipLaunchKernelGGL(k_v2_gstats, dim3(g0), dim3(256), 0, st, out, ab, g, (long long)M, (float*)gm, ws.part);
  GD_LAUNCH_CHECK();
  int rc = gd_partials_to_f64(st, ws.part, g0, 2 * V2_CO, ws.sums);
  if (rc) return rc;
  rc = gdmae_bn_bwd_coeffs(ws.sums, 2, stats, ab, gamma, V2_CO, (double)N, nullptr, dgamma, dbeta, accumulate, ws.c01, stream);
  if (rc) return rc;
  const int g1 = v2_grid(N, v2_resident_blocks(k_v2_dy<F16>, F16 ? 6 : 2, 4096));
  hipLaunchKernelGGL(k_v2_dy<F16>, dim3(g1), dim3(V2_WAVES * 64), 0, st, (const unsigned short*)y1, N, W, row_pillar, (const float*)ab,
                     (const float*)ws.c01, arg, (const float*)gm, (unsigned*)dy1);
  GD_LAUNCH_CHECK();
  int g2 = v2_resident_blocks(k_v2_dw<F16>, F16 ? 7 : 3, V2_DW_GRID);     // one tile per workgroup and iteration
  if (g2 > (N + 31) / 32) g2 = (int)((N + 31) / 32);
  hipLaunchKernelGGL(k_v2_dw<F16>, dim3(g2), dim3(V2_WAVES * 64), 0, st, (const unsigned short*)y1, N, W, row_pillar, (const float*)ab,
                     (const float*)ws.c01, arg, (const float*)gm, ws.part);
  GD_LAUNCH_CHECK();
  return gd_splitk_acc(st, ws.part, g2, (long long)V2_CO * V2_CI, dW, accumulate);
}
extern "C" int gdmae_vfe_max_layer_bwd(const void* y1, long long N, const void* W, const int* row_pillar, int M,
                                       const float* gamma, const double* stats, const float* ab,
                                       const float* out, const int* arg, const float* g, void* gm,
                                       void* dy1, float* dgamma, float* dbeta, float* dW, int accumulate, void* workspace,
                                       void* stream) {
  return v2_bwd<false>(y1, N, W, row_pillar, M, gamma, stats, ab, out, arg, g, gm, dy1, dgamma, dbeta, dW, accumulate, workspace, stream);
}
extern "C" int gdmae_vfe_max_layer_bwd_f16(const void* y1, long long N, const float* W, const int* row_pillar, int M,
                                           const float* gamma, const double* stats, const float* ab,
                                           const float* out, const int* arg, const float* g, void* gm,
                                           void* dy1, float* dgamma, float* dbeta, float* dW, int accumulate, void* workspace,
                                           void* stream) {
  return v2_bwd<true>(y1, N, W, row_pillar, M, gamma, stats, ab, out, arg, g, gm, dy1, dgamma, dbeta, dW, accumulate, workspace, stream);
}
