// Fused residual add + LayerNorm (forward and backward) for the post-norm encoder layers.
//
// Replaces, per EncoderLayer (reference pcdet/models/model_utils/sst_basic_block.py:77-84):
//   src = src + src2 ; src = self.norm1(src)      and      src = src + src2 ; src = self.norm2(src)
// i.e. an elementwise add followed by nn.LayerNorm(d, eps=1e-5) (torch: RowwiseMoments + LayerNormForward,
// and three kernels in the backward).  Token tensors are only (n, d <= 256): one wavefront per row, the row
// lives in registers (d/64 floats per lane), mean / variance by xor-shuffles, one pass over HBM.
// The residual branch `b` may be bf16 (GEMM output under autocast) or fp32; `a` and the output are fp32.
// Backward: dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)) goes to both inputs; dgamma / dbeta
// are accumulated per lane over the workgroup's rows and reduced in a fixed order by a second kernel.
#include "common.h"

__device__ inline float bf16_to_f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }

template <int VPL /* values per lane = d / 64 */, bool B_BF16>
__device__ inline void ln_load_sum(const float* __restrict__ a, const void* __restrict__ b, long long row, int lane,
                                   float (&s)[VPL]) {
  constexpr int D = VPL * GD_WAVE;
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    const int c = k * GD_WAVE + lane;
    float bv = B_BF16 ? bf16_to_f(((const unsigned short*)b)[row * D + c]) : ((const float*)b)[row * D + c];
    s[k] = a[row * D + c] + bv;
  }
}

template <int VPL, bool B_BF16>
__global__ __launch_bounds__(256) void k_add_ln_fwd(const float* __restrict__ a, const void* __restrict__ b,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    long long n, float eps, float* __restrict__ y,
                                                    float* __restrict__ stats /* (n, 2): mean, rstd */) {
  constexpr int D = VPL * GD_WAVE;
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  float g[VPL], bt[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    g[k] = gamma[k * GD_WAVE + lane];
    bt[k] = beta[k * GD_WAVE + lane];
  }
  for (long long row = blockIdx.x * 4ll + wib; row < n; row += gridDim.x * 4ll) {
    float s[VPL];
    ln_load_sum<VPL, B_BF16>(a, b, row, lane, s);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) sum += s[k];
    const float mean = gd_wave_sum(sum) * (1.f / D);
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const float dlt = s[k] - mean;
      sq = fmaf(dlt, dlt, sq);
    }
    const float var = gd_wave_sum(sq) * (1.f / D);
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int k = 0; k < VPL; ++k) y[row * D + k * GD_WAVE + lane] = (s[k] - mean) * rstd * g[k] + bt[k];
    if (lane == 0) {
      stats[row * 2] = mean;
      stats[row * 2 + 1] = rstd;
    }
  }
}

template <int VPL, bool B_BF16>
__global__ __launch_bounds__(256) void k_add_ln_bwd(const float* __restrict__ a, const void* __restrict__ b,
                                                    const float* __restrict__ gamma, const float* __restrict__ stats,
                                                    const float* __restrict__ dy, long long n, float* __restrict__ dx,
                                                    float* __restrict__ part /* (grid, 2, D) */) {
  constexpr int D = VPL * GD_WAVE;
  __shared__ float sh[4][2][D];
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  float g[VPL], dg[VPL], db[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    g[k] = gamma[k * GD_WAVE + lane];
    dg[k] = 0.f;
    db[k] = 0.f;
  }
  for (long long row = blockIdx.x * 4ll + wib; row < n; row += gridDim.x * 4ll) {
    float s[VPL];
    ln_load_sum<VPL, B_BF16>(a, b, row, lane, s);
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float gy[VPL], xh[VPL];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      const float d = dy[row * D + k * GD_WAVE + lane];
      xh[k] = (s[k] - mean) * rstd;
      gy[k] = d * g[k];
      m1 += gy[k];
      m2 = fmaf(gy[k], xh[k], m2);
      dg[k] = fmaf(d, xh[k], dg[k]);
      db[k] += d;
    }
    m1 = gd_wave_sum(m1) * (1.f / D);
    m2 = gd_wave_sum(m2) * (1.f / D);
#pragma unroll
    for (int k = 0; k < VPL; ++k) dx[row * D + k * GD_WAVE + lane] = rstd * (gy[k] - m1 - xh[k] * m2);
  }
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    sh[wib][0][k * GD_WAVE + lane] = dg[k];
    sh[wib][1][k * GD_WAVE + lane] = db[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * D; c += 256) {
    const int which = c / D, col = c % D;
    part[(long long)blockIdx.x * 2 * D + c] = sh[0][which][col] + sh[1][which][col] + sh[2][which][col] + sh[3][which][col];
  }
}

// out[c] = sum_b part[b, c] (fixed order), c in [0, C2); one workgroup per 16 columns, 16 partial slices per column
__global__ __launch_bounds__(256) void k_reduce_partials_f32(const float* __restrict__ part, int nblk, int C2,
                                                             float* __restrict__ out) {
  __shared__ float sh[16][17];
  const int cl = threadIdx.x & 15, ps = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  float acc = 0.f;
  if (c < C2)
    for (int b = ps; b < nblk; b += 16) acc += part[(long long)b * C2 + c];
  sh[ps][cl] = acc;
  __syncthreads();
  if (ps == 0 && c < C2) {
    float s = 0.f;
    for (int k = 0; k < 16; ++k) s += sh[k][cl];
    out[c] = s;
  }
}

static inline int ln_grid(long long n) {
  long long g = (n + 3) / 4;
  if (g > 256) g = 256;     // few partial blocks: the dgamma/dbeta reduction stays a handful of iterations
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" size_t gdmae_add_layernorm_workspace_bytes(int d) { return (size_t)2048 * 2 * d * sizeof(float); }

// y = LayerNorm(a + b) * gamma + beta over rows of d in {64, 128, 256}; b_is_bf16: dtype of b.  stats (n,2) out.
extern "C" int gdmae_add_layernorm_fwd(const float* a, const void* b, int b_is_bf16, const float* gamma, const float* beta,
                                       long long n, int d, float eps, float* y, float* stats, void* stream) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(ln_grid(n)), block(256);
#define GD_LN_FWD(V, BF) hipLaunchKernelGGL((k_add_ln_fwd<V, BF>), grid, block, 0, st, a, b, gamma, beta, n, eps, y, stats)
  if (d == 64) { if (b_is_bf16) GD_LN_FWD(1, true); else GD_LN_FWD(1, false); }
  else if (d == 128) { if (b_is_bf16) GD_LN_FWD(2, true); else GD_LN_FWD(2, false); }
  else if (d == 256) { if (b_is_bf16) GD_LN_FWD(4, true); else GD_LN_FWD(4, false); }
  else GD_REQUIRE(false, "add_layernorm supports d in {64, 128, 256}");
#undef GD_LN_FWD
  GD_LAUNCH_CHECK();
  return 0;
}

// dx (n,d) = gradient w.r.t. (a + b); dgamma_dbeta (2*d) = {dgamma, dbeta}; workspace from ..._workspace_bytes(d)
extern "C" int gdmae_add_layernorm_bwd(const float* a, const void* b, int b_is_bf16, const float* gamma, const float* stats,
                                       const float* dy, long long n, int d, float* dx, float* dgamma_dbeta, void* workspace,
                                       void* stream) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = ln_grid(n);
  const dim3 grid(nblk), block(256);
  float* part = (float*)workspace;
#define GD_LN_BWD(V, BF) hipLaunchKernelGGL((k_add_ln_bwd<V, BF>), grid, block, 0, st, a, b, gamma, stats, dy, n, dx, part)
  if (d == 64) { if (b_is_bf16) GD_LN_BWD(1, true); else GD_LN_BWD(1, false); }
  else if (d == 128) { if (b_is_bf16) GD_LN_BWD(2, true); else GD_LN_BWD(2, false); }
  else if (d == 256) { if (b_is_bf16) GD_LN_BWD(4, true); else GD_LN_BWD(4, false); }
  else GD_REQUIRE(false, "add_layernorm supports d in {64, 128, 256}");
#undef GD_LN_BWD
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_reduce_partials_f32, dim3(gd_div_up(2 * d, 16)), dim3(256), 0, st, part, nblk, 2 * d, dgamma_dbeta);
  GD_LAUNCH_CHECK();
  return 0;
}
