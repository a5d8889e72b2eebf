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
__device__ inline unsigned short f_to_bf16(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7F800000u) == 0x7F800000u) return (unsigned short)(u >> 16);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

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
                                                    float* __restrict__ stats /* (n, 2): mean, rstd */,
                                                    unsigned short* __restrict__ y_bf16 /* optional copy */,
                                                    const float* __restrict__ pos_table, const int* __restrict__ tok_pos,
                                                    unsigned short* __restrict__ ypos_bf16 /* optional: bf16(y + pos) */) {
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
    for (int k = 0; k < VPL; ++k) {
      const float o = (s[k] - mean) * rstd * g[k] + bt[k];
      y[row * D + k * GD_WAVE + lane] = o;
      if (y_bf16) y_bf16[row * D + k * GD_WAVE + lane] = f_to_bf16(o);
      if (ypos_bf16)   // q/k input of the NEXT layer (its gdmae_prep_tokens folded into this pass)
        ypos_bf16[row * D + k * GD_WAVE + lane] = f_to_bf16(o + pos_table[(long long)tok_pos[row] * D + k * GD_WAVE + lane]);
    }
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
                                                    float* __restrict__ part /* (grid, 3, D) */,
                                                    const void* __restrict__ dy2 /* optional 2nd gradient */, int dy2_bf16,
                                                    unsigned short* __restrict__ dx_bf16 /* optional copy */,
                                                    const void* __restrict__ dy3 /* optional 3rd gradient */, int dy3_bf16) {
  constexpr int D = VPL * GD_WAVE;
  __shared__ float sh[4][3][D];
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  float g[VPL], dg[VPL], db[VPL], dsx[VPL];
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    g[k] = gamma[k * GD_WAVE + lane];
    dg[k] = 0.f;
    db[k] = 0.f;
    dsx[k] = 0.f;
  }
  for (long long row = blockIdx.x * 4ll + wib; row < n; row += gridDim.x * 4ll) {
    float s[VPL];
    ln_load_sum<VPL, B_BF16>(a, b, row, lane, s);
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float gy[VPL], xh[VPL];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < VPL; ++k) {
      float d = dy[row * D + k * GD_WAVE + lane];
      if (dy2) d += dy2_bf16 ? bf16_to_f(((const unsigned short*)dy2)[row * D + k * GD_WAVE + lane])
                             : ((const float*)dy2)[row * D + k * GD_WAVE + lane];
      if (dy3) d += dy3_bf16 ? bf16_to_f(((const unsigned short*)dy3)[row * D + k * GD_WAVE + lane])
                             : ((const float*)dy3)[row * D + k * GD_WAVE + lane];
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
    for (int k = 0; k < VPL; ++k) {
      const float o = rstd * (gy[k] - m1 - xh[k] * m2);
      dx[row * D + k * GD_WAVE + lane] = o;
      if (dx_bf16) dx_bf16[row * D + k * GD_WAVE + lane] = f_to_bf16(o);
      dsx[k] += o;
    }
  }
#pragma unroll
  for (int k = 0; k < VPL; ++k) {
    sh[wib][0][k * GD_WAVE + lane] = dg[k];
    sh[wib][1][k * GD_WAVE + lane] = db[k];
    sh[wib][2][k * GD_WAVE + lane] = dsx[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 3 * D; c += 256) {
    const int which = c / D, col = c % D;
    part[(long long)blockIdx.x * 3 * D + c] = sh[0][which][col] + sh[1][which][col] + sh[2][which][col] + sh[3][which][col];
  }
}

// out[c] = sum_b part[b, c] (fixed order), c in [0, C2); one workgroup per 4 columns, 64 partial slices per column
__global__ __launch_bounds__(256) void k_reduce_partials_f32(const float* __restrict__ part, int nblk, int C2,
                                                             float* __restrict__ out) {
  const int cl = threadIdx.x >> 6, ps = threadIdx.x & 63;   // one wavefront per column
  const int c = blockIdx.x * 4 + cl;
  float acc = 0.f;
  if (c < C2)
    for (int b = ps; b < nblk; b += 64) acc += part[(long long)b * C2 + c];
  acc = gd_wave_sum(acc);
  if (ps == 0 && c < C2) out[c] = acc;
}

static inline int ln_grid(long long n) {
  long long g = (n + 3) / 4;
  if (g > 1024) g = 1024;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" size_t gdmae_add_layernorm_workspace_bytes(int d) { return (size_t)1024 * 3 * d * sizeof(float); }

// y = LayerNorm(a + b) * gamma + beta over rows of d in {64, 128, 256}; b_is_bf16: dtype of b.  stats (n,2) out.
// y_bf16 (optional, may be NULL): bf16 copy of y for the next GEMM; ypos_bf16 (optional): bf16(y + pos_table[tok_pos])
int gd_add_layernorm_fwd_ex(const float* a, const void* b, int b_is_bf16, const float* gamma, const float* beta, long long n, int d,
                            float eps, float* y, float* stats, void* y_bf16, const float* pos_table, const int* tok_pos,
                            void* ypos_bf16, hipStream_t st) {
  if (n <= 0) return 0;
  const dim3 grid(ln_grid(n)), block(256);
#define GD_LN_FWD(V, BF) hipLaunchKernelGGL((k_add_ln_fwd<V, BF>), grid, block, 0, st, a, b, gamma, beta, n, eps, y, stats, (unsigned short*)y_bf16, pos_table, tok_pos, (unsigned short*)ypos_bf16)
  if (d == 64) { if (b_is_bf16) GD_LN_FWD(1, true); else GD_LN_FWD(1, false); }
  else if (d == 128) { if (b_is_bf16) GD_LN_FWD(2, true); else GD_LN_FWD(2, false); }
  else if (d == 256) { if (b_is_bf16) GD_LN_FWD(4, true); else GD_LN_FWD(4, false); }
  else GD_REQUIRE(false, "add_layernorm supports d in {64, 128, 256}");
#undef GD_LN_FWD
  GD_LAUNCH_CHECK();
  return 0;
}
extern "C" int gdmae_add_layernorm_fwd(const float* a, const void* b, int b_is_bf16, const float* gamma, const float* beta,
                                       long long n, int d, float eps, float* y, float* stats, void* y_bf16, void* stream) {
  return gd_add_layernorm_fwd_ex(a, b, b_is_bf16, gamma, beta, n, d, eps, y, stats, y_bf16, nullptr, nullptr, nullptr,
                                 (hipStream_t)stream);
}

// dx (n,d) = gradient w.r.t. (a + b) for upstream gradient dy (+ dy2, + dy3 if not NULL); sums (3*d) = {dgamma, dbeta,
// column sums of dx (= bias gradient of the GEMM that produced b)}; dx_bf16 (optional): bf16 copy of dx;
// workspace from ..._workspace_bytes(d)
int gd_add_layernorm_bwd_ex(const float* a, const void* b, int b_is_bf16, const float* gamma, const float* stats, const float* dy,
                            const void* dy2, int dy2_bf16, const void* dy3, int dy3_bf16, long long n, int d, float* dx,
                            void* dx_bf16, float* sums, void* workspace, hipStream_t st) {
  if (n <= 0) return 0;
  const int nblk = ln_grid(n);
  const dim3 grid(nblk), block(256);
  float* part = (float*)workspace;
#define GD_LN_BWD(V, BF) hipLaunchKernelGGL((k_add_ln_bwd<V, BF>), grid, block, 0, st, a, b, gamma, stats, dy, n, dx, part, dy2, dy2_bf16, (unsigned short*)dx_bf16, dy3, dy3_bf16)
  if (d == 64) { if (b_is_bf16) GD_LN_BWD(1, true); else GD_LN_BWD(1, false); }
  else if (d == 128) { if (b_is_bf16) GD_LN_BWD(2, true); else GD_LN_BWD(2, false); }
  else if (d == 256) { if (b_is_bf16) GD_LN_BWD(4, true); else GD_LN_BWD(4, false); }
  else GD_REQUIRE(false, "add_layernorm supports d in {64, 128, 256}");
#undef GD_LN_BWD
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_reduce_partials_f32, dim3(gd_div_up(3 * d, 4)), dim3(256), 0, st, part, nblk, 3 * d, sums);
  GD_LAUNCH_CHECK();
  return 0;
}
extern "C" int gdmae_add_layernorm_bwd(const float* a, const void* b, int b_is_bf16, const float* gamma, const float* stats,
                                       const float* dy, const void* dy2, int dy2_bf16, long long n, int d, float* dx,
                                       void* dx_bf16, float* sums, void* workspace, void* stream) {
  return gd_add_layernorm_bwd_ex(a, b, b_is_bf16, gamma, stats, dy, dy2, dy2_bf16, nullptr, 0, n, d, dx, dx_bf16, sums, workspace,
                                 (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// small token-wise helpers of the hand-written encoder layer (gdmae_hip/encoder.py)
// ------------------------------------------------------------------------------------------------
// xo = x, xpo = x + pos_table[tok_pos] in the GEMM input dtype (bf16 under autocast, fp32 otherwise): replaces the
// `x + pos` add and two dtype casts in front of the q/k and v projections (sst_basic_block.py:44-49).
template <bool OBF>
__global__ __launch_bounds__(256) void k_prep_tokens(const float* __restrict__ x, const float* __restrict__ pos_table,
                                                     const int* __restrict__ tok_pos, long long n, int d,
                                                     void* __restrict__ xo, void* __restrict__ xpo) {
  const long long total = n * (long long)d;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / d;
    const int c = (int)(e % d);
    const float xv = x[e];
    const float pv = xv + pos_table[(long long)tok_pos[r] * d + c];
    if (OBF) {
      ((unsigned short*)xo)[e] = f_to_bf16(xv);
      ((unsigned short*)xpo)[e] = f_to_bf16(pv);
    } else {
      ((float*)xpo)[e] = pv;
    }
  }
}

extern "C" int gdmae_prep_tokens(const float* x, const float* pos_table, const int* tok_pos, long long n, int d, void* x_out,
                                 void* xpos_out, int out_bf16, void* stream) {
  if (n <= 0) return 0;
  long long g = (n * d + 255) / 256;
  if (g > 8192) g = 8192;
  if (out_bf16)
    hipLaunchKernelGGL((k_prep_tokens<true>), dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, pos_table, tok_pos, n, d, x_out, xpos_out);
  else
    hipLaunchKernelGGL((k_prep_tokens<false>), dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, pos_table, tok_pos, n, d, x_out, xpos_out);
  GD_LAUNCH_CHECK();
  return 0;
}

// out (fp32) = a (fp32) + b + c; b / c optional (NULL), each fp32 or bf16
__global__ __launch_bounds__(256) void k_add3(const float* __restrict__ a, const void* __restrict__ b, int b_bf16,
                                              const void* __restrict__ c, int c_bf16, long long total, float* __restrict__ out) {
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    float v = a[e];
    if (b) v += b_bf16 ? bf16_to_f(((const unsigned short*)b)[e]) : ((const float*)b)[e];
    if (c) v += c_bf16 ? bf16_to_f(((const unsigned short*)c)[e]) : ((const float*)c)[e];
    out[e] = v;
  }
}

extern "C" int gdmae_add3(const float* a, const void* b, int b_bf16, const void* c, int c_bf16, long long total, float* out,
                          void* stream) {
  if (total <= 0) return 0;
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_add3, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a, b, b_bf16, c, c_bf16, total, out);
  GD_LAUNCH_CHECK();
  return 0;
}
