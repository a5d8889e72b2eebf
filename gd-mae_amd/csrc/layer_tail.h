// The reductions that close an encoder layer's backward (split-K partial tiles of the five weight gradients and three bias column sums,
// LayerNorm partial rows, the attention's d loss / d tau partials): job tables + the device code of one 256-thread block of that work.
// Two hosts: k_layer_tail (encoder_layer.hip: a launch of its own, the launch-per-product layer path) and - the fused stage path - the first
// blocks of k_layer_bwd_in (layer_fused.hip): the reductions of layer i ride along with the in-projection launch of the same layer, which
// follows the grouped weight-gradient launch anyway and touches none of their buffers (12 launches per step less, the partial tiles are
// read back while the memory-side cache still holds them).  Same sums in the same order either way.
#pragma once
#include "common.h"

// dst[j][i] += src[j][i]  (nblk == 0)   or   dst[j][i] += sum_b src[j][b * stride + i]  (nblk partial rows, e.g. the
// per-workgroup dgamma / dbeta / column-sum partials of the LayerNorm backward), fixed association order.
struct AccJobs {
  float* dst[8];
  const float* src[8];
  int len[8], nblk[8], stride[8];
  int count;
};

// the split-K reduces of a layer's five weight gradients as one section each:
// dst[i] += sum_{s < S} part[s * P + i], same slicing and association order as k_splitk_acc (gemm.hip)
struct SplitkJobs {
  const float* part[8];
  float* dst[8];
  int S[8];
  long long P4[8];
  int count;
};

// The reductions that close a layer's backward as ONE grid (y = section):
//   y <  J.count      split-K reduce of weight-gradient / bias-column-sum partials (as k_splitk_acc_jobs)
//   y == J.count      vector accumulations (as k_acc_vectors: LayerNorm dgamma / dbeta / column-sum partial rows)
//   y == J.count + 1  dtau += gate(tau) * sum of the attention partials (fixed order, one workgroup)
struct TailJobs {
  SplitkJobs J;
  AccJobs a;
  const float* tau_part;
  long long n_part;
  const float* tau;
  float tau_min;
  float* dtau;
};
__device__ __forceinline__ void tail_splitk(const SplitkJobs& J, int job, const unsigned bx, const int tid) {
  // round 5: one 16-byte column per thread, ALL slices of it requested before the first is added (eight at a time, unconditional on
  // a clamped slice index), added in slice order - 256 contiguous columns per workgroup instead of 64 columns x 4 slice groups
  // meeting in LDS: a quarter of the workgroups, no barrier, 4 KB per wavefront-row of a slice
  const long long P4 = J.P4[job];
  const int S = J.S[job];
  const long long i = bx * 256ll + tid;
  if (bx * 256ll >= P4) return;                // uniform per workgroup
  if (i >= P4) return;
  const float4* p = (const float4*)J.part[job] + i;
  float4* d = (float4*)J.dst[job] + i;
  float4 acc = *d;
  for (int s0 = 0; s0 < S; s0 += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(long long)(s0 + u < S ? s0 + u : S - 1) * P4];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float m = s0 + u < S ? 1.f : 0.f;
      acc.x = fmaf(m, v[u].x, acc.x); acc.y = fmaf(m, v[u].y, acc.y); acc.z = fmaf(m, v[u].z, acc.z); acc.w = fmaf(m, v[u].w, acc.w);
    }
  }
  *d = acc;
}
__device__ __forceinline__ void tail_vectors(const AccJobs& a, const unsigned bx, const int tid) {
  __shared__ float sh[16][17];
  const int cl = tid & 15, ps = tid >> 4;
  int col = bx * 16 + cl;
  int j = 0;
  while (j < a.count && col >= a.len[j]) col -= a.len[j++];
  float acc = 0.f;
  if (j < a.count) {
    if (a.nblk[j] == 0) {
      if (ps == 0) acc = a.src[j][col];
    } else {
      const float* p = a.src[j] + col;
      const long long st = a.stride[j];
#pragma unroll 8
      for (int b = ps; b < a.nblk[j]; b += 16) acc += p[b * st];
    }
  }
  sh[ps][cl] = acc;
  __syncthreads();
  if (ps == 0 && j < a.count) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sh[k][cl];
    a.dst[j][col] += s;
  }
}
// one 256-thread block (bx, y) of the tail grid (tail_grid_x(T), T.J.count + 2); tid = 0 ... 255.  Barriers inside: every thread of the
// calling workgroup that is still alive must come here with the same (bx, y)
__device__ __forceinline__ void tail_block(const TailJobs& T, const unsigned bx, const int y, const int tid) {
  if (y < T.J.count) {
    tail_splitk(T.J, y, bx, tid);
  } else if (y == T.J.count) {
    int cols = 0;
    for (int j = 0; j < T.a.count; ++j) cols += T.a.len[j];
    if (bx * 16 < cols) tail_vectors(T.a, bx, tid);
  } else if (bx == 0) {
    __shared__ float shw[4];
    // 16-byte loads, four of them in flight per thread: the ~50 k partials of a layer are one latency-bound chain per thread
    // otherwise (this single workgroup was the longest-running part of the launch)
    // (round 5: sixteen in flight, requested unconditionally on a clamped index - with four the workgroup needed twelve dependent
    // round trips for the ~50 k partials of a d = 256 layer and set the duration of the whole launch: 13.7 us)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const long long n4 = T.n_part >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(T.tau_part);
    for (long long i0 = tid; i0 < n4; i0 += 16 * 256) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = p4[i0 + u * 256 < n4 ? i0 + u * 256 : n4 - 1];
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        const float m0 = i0 + u * 256 < n4 ? 1.f : 0.f, m1 = i0 + (u + 1) * 256 < n4 ? 1.f : 0.f;
        const float m2 = i0 + (u + 2) * 256 < n4 ? 1.f : 0.f, m3 = i0 + (u + 3) * 256 < n4 ? 1.f : 0.f;
        a0 = fmaf(m0, (v[u].x + v[u].y) + (v[u].z + v[u].w), a0);
        a1 = fmaf(m1, (v[u + 1].x + v[u + 1].y) + (v[u + 1].z + v[u + 1].w), a1);
        a2 = fmaf(m2, (v[u + 2].x + v[u + 2].y) + (v[u + 2].z + v[u + 2].w), a2);
        a3 = fmaf(m3, (v[u + 3].x + v[u + 3].y) + (v[u + 3].z + v[u + 3].w), a3);
      }
    }
    for (long long j = (n4 << 2) + tid; j < T.n_part; j += 256) a1 += T.tau_part[j];
    const float w = gd_wave_sum((a0 + a1) + (a2 + a3));
    if ((tid & 63) == 0) shw[tid >> 6] = w;
    __syncthreads();
    if (tid == 0) {
      float t = (shw[0] + shw[1]) + (shw[2] + shw[3]);
      if (!(T.tau[0] >= T.tau_min)) t = 0.f;            // d clamp(tau, min) / d tau
      T.dtau[0] += t;
    }
  }
}
// x extent of the tail grid: the widest section
static inline long long tail_grid_x(const TailJobs& T) {
  int cols = 0;
  for (int q = 0; q < T.a.count; ++q) cols += T.a.len[q];
  long long gx = (cols + 15) / 16;
  for (int q = 0; q < T.J.count; ++q) gx = (T.J.P4[q] + 255) / 256 > gx ? (T.J.P4[q] + 255) / 256 : gx;
  return gx > 0 ? gx : 1;
}
