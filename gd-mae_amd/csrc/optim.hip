// Fused global-norm clip + decoupled weight decay + Adam over ONE flat fp32 parameter buffer.
//
// Replaces the per-parameter python loops of the reference optimizer step:
//   clip_grad_norm_(model.parameters(), 10)                      reference tools/train_utils/train_utils.py:52
//   OptimWrapper.step: p.mul_(1 - wd*lr) for every param, then Adam.step()
//                                                                tools/train_utils/optimization/fastai_optim.py:135-152
// (~190 tensors x several kernels each in the reference; two launches here).  The same flat gradient
// buffer is what the data-parallel all-reduce operates on, so no gather/scatter of gradients is needed.
#include "common.h"

__global__ __launch_bounds__(256) void k_sq_partials(const float* __restrict__ g, long long n, float* __restrict__ part) {
  __shared__ float sh[4];
  // 16-byte loads, four in flight per thread (requested unconditionally on a clamped index): the 32 MB gradient buffer moved at
  // 1.9 TB/s through 4-byte accesses
  float a4[4] = {0.f, 0.f, 0.f, 0.f};
  const long long n4 = (reinterpret_cast<unsigned long long>(g) & 15ull) == 0 ? n / 4 : 0;
  const float4* gv = reinterpret_cast<const float4*>(g);
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x; i0 < n4; i0 += 4 * stride) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = gv[i0 + u * stride < n4 ? i0 + u * stride : n4 - 1];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float m = i0 + u * stride < n4 ? 1.f : 0.f;
      a4[u] = fmaf(m, fmaf(v[u].x, v[u].x, fmaf(v[u].y, v[u].y, fmaf(v[u].z, v[u].z, v[u].w * v[u].w))), a4[u]);
    }
  }
  for (long long i = 4 * n4 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += stride) a4[0] = fmaf(g[i], g[i], a4[0]);
  float acc = gd_wave_sum((a4[0] + a4[1]) + (a4[2] + a4[3]));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x / 64] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

#define GD_ADAM_MAX_SEG 64
struct AdamArgs {
  float lr, beta1, beta2, eps, wd, max_norm;
  float bc1, bc2_sqrt;  // 1 - beta1^t, sqrt(1 - beta2^t)
  float grad_scale;     // 1 / world size: the all-reduce SUMS, the average is folded in here
  int nseg;
  long long seg_begin[GD_ADAM_MAX_SEG], seg_end[GD_ADAM_MAX_SEG];   // optimised element ranges of the flat buffers
};

__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                              float* __restrict__ v, AdamArgs A, const float* __restrict__ sq_norm,
                                              unsigned short* __restrict__ p_bf16) {
  float coef = A.grad_scale;
  if (A.max_norm > 0.f) {
    const float total = sqrtf(*sq_norm) * A.grad_scale;      // norm of the AVERAGED gradient
    coef *= fminf(A.max_norm / (total + 1e-6f), 1.f);
  }
  const float decay = 1.f - A.wd * A.lr;
  const float step = A.lr / A.bc1;
  for (int s = 0; s < A.nseg; ++s) {
    const long long e = A.seg_end[s];
    for (long long i = A.seg_begin[s] + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < e; i += (long long)gridDim.x * blockDim.x) {
      const float gi = g[i] * coef;
      float pi = p[i] * decay;
      const float mi = A.beta1 * m[i] + (1.f - A.beta1) * gi;
      const float vi = A.beta2 * v[i] + (1.f - A.beta2) * gi * gi;
      m[i] = mi;
      v[i] = vi;
      const float denom = sqrtf(vi) / A.bc2_sqrt + A.eps;
      const float pn = pi - step * (mi / denom);
      p[i] = pn;
      if (p_bf16) {                                    // the bf16 shadow the GEMMs read: written here instead of by a cast pass
        p_bf16[i] = gd_to_bf16(pn);
      }
    }
  }
}

// sq_norm_out[0] receives |g|^2 (after the caller's all-reduce of g); partials needs >= 1024 floats.
extern "C" int gdmae_grad_sq_norm(const float* grad, long long n, float* partials, float* sq_norm_out, void* stream);
extern "C" int gdmae_sum_partials(const float* part, long long n, float scale, float* out, int accumulate, void* stream);

extern "C" int gdmae_grad_sq_norm(const float* grad, long long n, float* partials, float* sq_norm_out, void* stream) {
  const int nb = 1024;
  hipLaunchKernelGGL(k_sq_partials, dim3(nb), dim3(256), 0, (hipStream_t)stream, grad, n, partials);
  GD_LAUNCH_CHECK();
  return gdmae_sum_partials(partials, nb, 1.f, sq_norm_out, 0, stream);
}

// segments: HOST array of 2 * n_segments element offsets [begin, end) into the flat buffers (the parameters that are
// optimised; everything else only takes part in the norm).  grad_scale multiplies the gradient (and the norm) first:
// 1 / world size after a SUM all-reduce.
extern "C" int gdmae_adam_step_shadow(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const long long* segments,
                                      int n_segments, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                      float max_norm, float grad_scale, const float* sq_norm, void* param_bf16, void* stream);
extern "C" int gdmae_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const long long* segments,
                               int n_segments, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                               float max_norm, float grad_scale, const float* sq_norm, void* stream) {
  return gdmae_adam_step_shadow(param, grad, exp_avg, exp_avg_sq, segments, n_segments, lr, beta1, beta2, eps, weight_decay, step, max_norm,
                                grad_scale, sq_norm, nullptr, stream);
}
// ... and the bf16 copy of every UPDATED element into param_bf16 (same offsets; may be null)
extern "C" int gdmae_adam_step_shadow(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const long long* segments,
                                      int n_segments, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                      float max_norm, float grad_scale, const float* sq_norm, void* param_bf16, void* stream) {
  GD_REQUIRE(step >= 1, "step counts from 1");
  GD_REQUIRE(n_segments >= 0 && n_segments <= GD_ADAM_MAX_SEG, "adam_step: at most 64 segments");
  AdamArgs A;
  A.lr = lr;
  A.beta1 = beta1;
  A.beta2 = beta2;
  A.eps = eps;
  A.wd = weight_decay;
  A.max_norm = max_norm;
  A.grad_scale = grad_scale;
  A.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  A.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  long long longest = 0;
  A.nseg = 0;
  for (int s = 0; s < n_segments; ++s) {
    const long long b = segments[2 * s], e = segments[2 * s + 1];
    GD_REQUIRE(b >= 0 && e >= b, "adam_step: bad segment");
    if (e == b) continue;
    A.seg_begin[A.nseg] = b;
    A.seg_end[A.nseg] = e;
    ++A.nseg;
    if (e - b > longest) longest = e - b;
  }
  if (A.nseg == 0) return 0;                       // nothing to optimise
  int grid = gd_div_up(longest, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_adam, dim3(grid), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, A, sq_norm,
                     (unsigned short*)param_bf16);
  GD_LAUNCH_CHECK();
  return 0;
}
