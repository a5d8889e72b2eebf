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
  if (threadIdx.x == 0) {
    s_prefix = 0u;
    s_k = len_keep;
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const unsigned hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
    if (threadIdx.x < 256) hist[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned prefix = s_prefix;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
      unsigned u = gd_fkey(noise[off + i]);
      if ((u & hi_mask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int k = s_k;
      unsigned cum = 0u;
      for (int d = 0; d < 256; ++d) {
        unsigned h = hist[d];
        if ((int)(cum + h) >= k) {
          s_prefix = prefix | ((unsigned)d << shift);
          s_k = k - (int)cum;
          break;
        }
        cum += h;
      }
    }
    __syncthreads();
  }
  const unsigned thr = s_prefix;
  const int ties_to_keep = s_k;  // >= 1
  int carry = 0;
  for (int base = 0; base < L; base += blockDim.x) {
    const int i = base + threadIdx.x;
    unsigned u = i < L ? gd_fkey(noise[off + i]) : 0xFFFFFFFFu;
    const int eq = (i < L && u == thr) ? 1 : 0;
    int tot;
    const int ord = carry + gd_block_exclusive_scan<int, 1024>(eq, tot, s_scan);
    carry += tot;
    if (i < L) {
      const bool keep = (u < thr) || (eq && ord < ties_to_keep);
      mask[off + i] = keep ? 0.f : 1.f;
    }
  }
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
