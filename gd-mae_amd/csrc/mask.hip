// MAE random masking on the device.
//
// Replaces reference pcdet/utils/common_utils.py:49-63 (random_masking: rand -> argsort -> scatter)
// and the per-sample python loop with a .sum().item() host sync per sample at
// pcdet/models/backbones_3d/spt_backbone_mae.py:96-100.
//
// A full argsort is not needed: a pillar is visible iff its noise is among the len_keep smallest
// of its sample (ties -> lower index first = stable argsort, the canonical order).  One 1024-thread
// workgroup per sample finds the len_keep-th smallest value with a 4 x 8-bit radix select (LDS
// histogram), then one ordered pass assigns the mask, ranking exact ties with a workgroup scan.
// len_keep = int(L * (1 - ratio)) is evaluated in fp64 exactly like the python expression.
#include "common.h"

__device__ inline unsigned gd_fkey(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// digit of the k-th smallest key among the histogram's entries, by ONE wavefront (4 bins per lane + wave scan) instead of a
// 256-step serial walk by one thread (each step a dependent LDS read: ~8 us per radix pass)
__device__ inline void mask_pick_digit(const unsigned* hist, unsigned prefix, int shift, unsigned* s_prefix, int* s_k) {
  const int lane = threadIdx.x;
  if (lane >= GD_WAVE) return;
  const int k = *s_k;
  const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
  const int mine = (int)(h0 + h1 + h2 + h3);
  const int incl = gd_wave_inclusive_scan(mine);
  const int excl = incl - mine;
  const unsigned long long hit = __ballot(incl >= k);           // first lane whose cumulative count reaches k
  const int first = __ffsll((long long)hit) - 1;
  if (lane == first) {
    int cum = excl, d = 4 * lane;
    const unsigned h[4] = {h0, h1, h2, h3};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (cum + (int)h[q] >= k) {
        d = 4 * lane + q;
        break;
      }
      cum += (int)h[q];
    }
    *s_prefix = prefix | ((unsigned)d << shift);
    *s_k = k - cum;
  }
}

constexpr int kMaskKpt = 16;     // keys a thread keeps in registers: samples up to 16 K pillars read the noise once

template <bool CACHED>
__device__ inline void mask_select_body(const float* __restrict__ noise, int off, int L, int len_keep, float* __restrict__ mask,
                                        unsigned* hist, unsigned* s_prefix, int* s_k, int* s_scan) {
  const int tid = threadIdx.x;
  unsigned kreg[CACHED ? kMaskKpt : 1];
  if (CACHED) {
#pragma unroll
    for (int r = 0; r < kMaskKpt; ++r) {
      const int i = r * 1024 + tid;
      kreg[r] = i < L ? gd_fkey(noise[off + i]) : 0u;
    }
  }
  if (tid == 0) {
    *s_prefix = 0u;
    *s_k = len_keep;
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const unsigned hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    const unsigned prefix = *s_prefix;
    if (CACHED) {
#pragma unroll
      for (int r = 0; r < kMaskKpt; ++r) {
        const int i = r * 1024 + tid;
        if (i < L && (kreg[r] & hi_mask) == prefix) atomicAdd(&hist[(kreg[r] >> shift) & 255u], 1u);
      }
    } else {
      for (int i = tid; i < L; i += 1024) {
        const unsigned u = gd_fkey(noise[off + i]);
        if ((u & hi_mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    mask_pick_digit(hist, prefix, shift, s_prefix, s_k);
    __syncthreads();
  }
  const unsigned thr = *s_prefix;
  const int ties_to_keep = *s_k;  // >= 1
  int carry = 0;
  const int rounds = (L + 1023) / 1024;
  for (int r = 0; r < rounds; ++r) {
    const int i = r * 1024 + tid;
    unsigned u = 0xFFFFFFFFu;
    if (i < L) {
      if (CACHED) {
        u = 0u;
#pragma unroll
        for (int q = 0; q < kMaskKpt; ++q) u = q == r ? kreg[q] : u;
      } else {
        u = gd_fkey(noise[off + i]);
      }
    }
    const int eq = (i < L && u == thr) ? 1 : 0;
    // exact ties with the threshold are rare: the ordered scan only runs for the chunks that contain one
    const int n_eq = __syncthreads_count(eq);
    int ord = carry;
    if (n_eq > 0) {
      int tot;
      ord = carry + gd_block_exclusive_scan<int, 1024>(eq, tot, s_scan);
      carry += tot;
    }
    if (i < L) {
      const bool keep = (u < thr) || (eq && ord < ties_to_keep);
      mask[off + i] = keep ? 0.f : 1.f;
    }
  }
}

__global__ __launch_bounds__(1024) void k_mask_select(const float* __restrict__ noise,
                                                      const int* __restrict__ sample_off, double keep_frac,
                                                      float* __restrict__ mask, int* __restrict__ len_keep_out) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix;
  __shared__ int s_k;
  __shared__ int s_scan[1024 / GD_WAVE + 1];
  const int b = blockIdx.x;
  const int off = sample_off[b];
  const int L = sample_off[b + 1] - off;
  const int len_keep = (int)((double)L * keep_frac);
  if (threadIdx.x == 0) len_keep_out[b] = len_keep;
  if (len_keep <= 0 || len_keep >= L) {
    const float v = len_keep <= 0 ? 1.f : 0.f;
    for (int i = threadIdx.x; i < L; i += blockDim.x) mask[off + i] = v;
    return;
  }
  if (L <= kMaskKpt * 1024) mask_select_body<true>(noise, off, L, len_keep, mask, hist, &s_prefix, &s_k, s_scan);
  else mask_select_body<false>(noise, off, L, len_keep, mask, hist, &s_prefix, &s_k, s_scan);
}

// noise: (M,) one value per pillar, pillars grouped by sample (sample_pillar_off from gdmae_voxelize)
extern "C" int gdmae_random_mask(const float* noise, const int* sample_pillar_off, int batch_size, double keep_frac,
                                 float* mask_out, int* len_keep_out, void* stream) {
  GD_REQUIRE(batch_size > 0, "batch_size");
  hipLaunchKernelGGL(k_mask_select, dim3(batch_size), dim3(1024), 0, (hipStream_t)stream, noise, sample_pillar_off,
                     keep_frac, mask_out, len_keep_out);
  GD_LAUNCH_CHECK();
  return 0;
}
