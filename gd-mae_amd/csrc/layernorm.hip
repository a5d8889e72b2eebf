// Fused residual add + LayerNorm (forward and backward) for the post-norm encoder layers.
//
// Replaces, per EncoderLayer (reference pcdet/models/model_utils/sst_basic_block.py:77-84):
//   src = src + src2 ; src = self.norm1(src)      and      src = src + src2 ; src = self.norm2(src)
// i.e. an elementwise add followed by nn.LayerNorm(d, eps=1e-5) (torch: RowwiseMoments + LayerNormForward,
// and three kernels in the backward).  Token tensors are only (n, d <= 256): the row lives in registers (4 consecutive
// columns per lane, 16-byte accesses), mean / variance by xor-shuffles inside the d/4-lane group, one pass over HBM.
// The residual branch `b` may be bf16 (GEMM output under autocast) or fp32; `a` and the output are fp32.
// Backward: dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat)) goes to both inputs; dgamma / dbeta
// are accumulated per lane over the workgroup's rows and reduced in a fixed order by a second kernel.
#include "common.h"

__device__ inline float bf16_to_f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ inline unsigned short f_to_bf16(float f) { return gd_to_bf16(f); }

// Layout: a lane owns 4 consecutive columns (one 16-byte access per fp32 operand, 8 bytes per bf16 operand), LPR = d / 4
// lanes form a row, a wavefront holds 64 / LPR rows (4 / 2 / 1 for d = 64 / 128 / 256); reductions are xor-shuffles
// inside the LPR-lane group.
template <int LPR>
__device__ inline float ln_group_sum(float v) {
  return gd_group_sum<LPR>(v);       // the same reduction as the fused epilogues (tok_tiles.h): rows stay bit-identical to them
}
__device__ inline void ln_ld4(const float* __restrict__ p, long long e, float (&v)[4]) {
  const float4 q = *reinterpret_cast<const float4*>(p + e);
  v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
__device__ inline void ln_ld4_bf(const void* __restrict__ p, long long e, float (&v)[4]) {
  const uint2 q = *reinterpret_cast<const uint2*>((const unsigned short*)p + e);
  v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xFFFF0000u);
  v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xFFFF0000u);
}
__device__ inline void ln_ld4_any(const void* __restrict__ p, int bf, long long e, float (&v)[4]) {
  if (bf) ln_ld4_bf(p, e, v); else ln_ld4((const float*)p, e, v);
}
__device__ inline void ln_st4(float* __restrict__ p, long long e, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p + e) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ inline void ln_st4_bf(unsigned short* __restrict__ p, long long e, const float (&v)[4]) {
  uint2 q;
  q.x = (unsigned)f_to_bf16(v[0]) | ((unsigned)f_to_bf16(v[1]) << 16);
  q.y = (unsigned)f_to_bf16(v[2]) | ((unsigned)f_to_bf16(v[3]) << 16);
  *reinterpret_cast<uint2*>(p + e) = q;
}

template <int D, bool B_BF16>
__global__ __launch_bounds__(256) void k_add_ln_fwd(const float* __restrict__ a, const void* __restrict__ b,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    long long n, float eps, float* __restrict__ y,
                                                    float* __restrict__ stats /* (n, 2): mean, rstd */,
                                                    unsigned short* __restrict__ y_bf16 /* optional copy */,
                                                    const float* __restrict__ pos_table, const int* __restrict__ tok_pos,
                                                    unsigned short* __restrict__ ypos_bf16 /* optional: bf16(y + pos) */) {
  constexpr int LPR = D / 4, RPW = GD_WAVE / LPR, RPB = 4 * RPW;   // lanes per row, rows per wavefront / workgroup
  const int lane = threadIdx.x & (GD_WAVE - 1), wib = threadIdx.x / GD_WAVE;
  const int c0 = 4 * (lane % LPR), rl = wib * RPW + lane / LPR;
  float g[4], bt[4];
  ln_ld4(gamma, c0, g);
  ln_ld4(beta, c0, bt);
  for (long long r0 = (long long)blockIdx.x * RPB; r0 < n; r0 += (long long)gridDim.x * RPB) {
    const long long row = r0 + rl;
    const bool live = row < n;
    const long long e = (live ? row : n - 1) * D + c0;
    float s[4], bv[4];
    ln_ld4(a, e, s);
    if (B_BF16) ln_ld4_bf(b, e, bv); else ln_ld4((const float*)b, e, bv);
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += bv[k];
    const float mean = ln_group_sum<LPR>((s[0] + s[1]) + (s[2] + s[3])) * (1.f / D);
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dlt = s[k] - mean;
      sq = fmaf(dlt, dlt, sq);
    }
    const float rstd = rsqrtf(ln_group_sum<LPR>(sq) * (1.f / D) + eps);
    if (!live) continue;
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (s[k] - mean) * rstd * g[k] + bt[k];
    ln_st4(y, e, o);
    if (y_bf16) ln_st4_bf(y_bf16, e, o);
    if (ypos_bf16) {   // q/k input of the NEXT layer (its gdmae_prep_tokens folded into this pass)
      float pv[4];
      ln_ld4(pos_table, (long long)tok_pos[row] * D + c0, pv);
#pragma unroll
      for (int k = 0; k < 4; ++k) pv[k] += o[k];
      ln_st4_bf(ypos_bf16, e, pv);
    }
    if (c0 == 0) *reinterpret_cast<float2*>(stats + row * 2) = make_float2(mean, rstd);
  }
}

template <int D, bool B_BF16>
__global__ __launch_bounds__(256) void k_add_ln_bwd(const float* __restrict__ a, const void* __restrict__ b,
                                                    const float* __restrict__ gamma, const float* __restrict__ stats,
                                                    const float* __restrict__ dy, long long n, float* __restrict__ dx,
                                                    float* __restrict__ part /* (grid, 3, D) */,
                                                    const void* __restrict__ dy2 /* optional 2nd gradient */, int dy2_bf16,
                                                    unsigned short* __restrict__ dx_bf16 /* optional copy */,
                                                    const void* __restrict__ dy3 /* optional 3rd gradient */, int dy3_bf16) {
  constexpr int LPR = D / 4, RPW = GD_WAVE / LPR, RPB = 4 * RPW;
  __shared__ float sh[RPB][3][D];
  const int lane = threadIdx.x & (GD_WAVE - 1), wib = threadIdx.x / GD_WAVE;
  const int c0 = 4 * (lane % LPR), rl = wib * RPW + lane / LPR;
  float g[4], dg[4], db[4], dsx[4];
  ln_ld4(gamma, c0, g);
#pragma unroll
  for (int k = 0; k < 4; ++k) dg[k] = db[k] = dsx[k] = 0.f;
  for (long long r0 = (long long)blockIdx.x * RPB; r0 < n; r0 += (long long)gridDim.x * RPB) {
    const long long row = r0 + rl;
    const bool live = row < n;
    const long long e = (live ? row : n - 1) * D + c0;
    float s[4], bv[4], d[4], t[4];
    ln_ld4(a, e, s);
    if (B_BF16) ln_ld4_bf(b, e, bv); else ln_ld4((const float*)b, e, bv);
    ln_ld4(dy, e, d);
    if (dy2) {
      ln_ld4_any(dy2, dy2_bf16, e, t);
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] += t[k];
    }
    if (dy3) {
      ln_ld4_any(dy3, dy3_bf16, e, t);
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] += t[k];
    }
    const float2 mr = *reinterpret_cast<const float2*>(stats + (live ? row : n - 1) * 2);
    const float mean = mr.x, rstd = mr.y;
    float gy[4], xh[4];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!live) d[k] = 0.f;
      xh[k] = (s[k] + bv[k] - mean) * rstd;
      gy[k] = d[k] * g[k];
      m1 += gy[k];
      m2 = fmaf(gy[k], xh[k], m2);
      dg[k] = fmaf(d[k], xh[k], dg[k]);
      db[k] += d[k];
    }
    m1 = ln_group_sum<LPR>(m1) * (1.f / D);
    m2 = ln_group_sum<LPR>(m2) * (1.f / D);
    if (!live) continue;
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k] = rstd * (gy[k] - m1 - xh[k] * m2);
      dsx[k] += o[k];
    }
    ln_st4(dx, e, o);
    if (dx_bf16) ln_st4_bf(dx_bf16, e, o);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sh[rl][0][c0 + k] = dg[k];
    sh[rl][1][c0 + k] = db[k];
    sh[rl][2][c0 + k] = dsx[k];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 3 * D; c += 256) {
    const int which = c / D, col = c % D;
    float acc = 0.f;
#pragma unroll
    for (int r = 0; r < RPB; ++r) acc += sh[r][which][col];
    part[(long long)blockIdx.x * 3 * D + c] = acc;
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

static inline int ln_grid(long long n, int d, int cap = 1024) {
  const int rpb = 4 * (64 / (d / 4));            // rows per workgroup pass
  long long g = (n + rpb - 1) / rpb;
  if (g > cap) g = cap;                          // backward: one row of partials per workgroup (workspace = 1024 rows)
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
  const dim3 grid(ln_grid(n, d, 4096)), block(256);   // no partials in the forward; per call: 512 blocks 23.9 us, 1024: 17.6 us, 4096: 17.2 us
#define GD_LN_FWD(V, BF) hipLaunchKernelGGL((k_add_ln_fwd<V, BF>), grid, block, 0, st, a, b, gamma, beta, n, eps, y, stats, (unsigned short*)y_bf16, pos_table, tok_pos, (unsigned short*)ypos_bf16)
  if (d == 64) { if (b_is_bf16) GD_LN_FWD(64, true); else GD_LN_FWD(64, false); }
  else if (d == 128) { if (b_is_bf16) GD_LN_FWD(128, true); else GD_LN_FWD(128, false); }
  else if (d == 256) { if (b_is_bf16) GD_LN_FWD(256, true); else GD_LN_FWD(256, false); }
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
  const int nblk = ln_grid(n, d);
  const dim3 grid(nblk), block(256);
  float* part = (float*)workspace;
#define GD_LN_BWD(V, BF) hipLaunchKernelGGL((k_add_ln_bwd<V, BF>), grid, block, 0, st, a, b, gamma, stats, dy, n, dx, part, dy2, dy2_bf16, (unsigned short*)dx_bf16, dy3, dy3_bf16)
  if (d == 64) { if (b_is_bf16) GD_LN_BWD(64, true); else GD_LN_BWD(64, false); }
  else if (d == 128) { if (b_is_bf16) GD_LN_BWD(128, true); else GD_LN_BWD(128, false); }
  else if (d == 256) { if (b_is_bf16) GD_LN_BWD(256, true); else GD_LN_BWD(256, false); }
  else GD_REQUIRE(false, "add_layernorm supports d in {64, 128, 256}");
#undef GD_LN_BWD
  GD_LAUNCH_CHECK();
  if (sums) {   // sums == NULL: the caller reduces the (gd_ln_partial_rows(n, d), 3, d) partials in `workspace` itself
    hipLaunchKernelGGL(k_reduce_partials_f32, dim3(gd_div_up(3 * d, 4)), dim3(256), 0, st, part, nblk, 3 * d, sums);
    GD_LAUNCH_CHECK();
  }
  return 0;
}
int gd_ln_partial_rows(long long n, int d) { return ln_grid(n, d); }
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

// out (fp32 or bf16) = a (fp32) + b + c; b / c optional (NULL), each fp32 or bf16.  8 elements per thread (16 / 32-byte accesses);
// the tail (total % 8) goes element by element.
__device__ inline void add3_load8(const void* p, int bf, long long e8, float (&v)[8]) {
  if (bf) {
    const uint4 q = reinterpret_cast<const uint4*>(p)[e8];
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] += __uint_as_float(w[j] << 16);
      v[2 * j + 1] += __uint_as_float(w[j] & 0xFFFF0000u);
    }
  } else {
    const float4 x = reinterpret_cast<const float4*>(p)[2 * e8], y = reinterpret_cast<const float4*>(p)[2 * e8 + 1];
    v[0] += x.x; v[1] += x.y; v[2] += x.z; v[3] += x.w; v[4] += y.x; v[5] += y.y; v[6] += y.z; v[7] += y.w;
  }
}
__device__ inline unsigned add3_f2bf(float f) { return gd_to_bf16(f); }
__global__ __launch_bounds__(256) void k_add3(const float* __restrict__ a, const void* __restrict__ b, int b_bf16,
                                              const void* __restrict__ c, int c_bf16, long long total, void* __restrict__ out, int out_bf16) {
  const long long n8 = total >> 3;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < n8; e += (long long)gridDim.x * blockDim.x) {
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    add3_load8(a, 0, e, v);
    if (b) add3_load8(b, b_bf16, e, v);
    if (c) add3_load8(c, c_bf16, e, v);
    if (out_bf16) {
      uint4 q;
      q.x = add3_f2bf(v[0]) | (add3_f2bf(v[1]) << 16); q.y = add3_f2bf(v[2]) | (add3_f2bf(v[3]) << 16);
      q.z = add3_f2bf(v[4]) | (add3_f2bf(v[5]) << 16); q.w = add3_f2bf(v[6]) | (add3_f2bf(v[7]) << 16);
      reinterpret_cast<uint4*>(out)[e] = q;
    } else {
      reinterpret_cast<float4*>(out)[2 * e] = make_float4(v[0], v[1], v[2], v[3]);
      reinterpret_cast<float4*>(out)[2 * e + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
  if (blockIdx.x == 0)
    for (long long e = (n8 << 3) + threadIdx.x; e < total; e += blockDim.x) {
      float v = a[e];
      if (b) v += b_bf16 ? bf16_to_f(((const unsigned short*)b)[e]) : ((const float*)b)[e];
      if (c) v += c_bf16 ? bf16_to_f(((const unsigned short*)c)[e]) : ((const float*)c)[e];
      if (out_bf16) ((unsigned short*)out)[e] = (unsigned short)add3_f2bf(v);
      else ((float*)out)[e] = v;
    }
}

extern "C" int gdmae_add3_to(const float* a, const void* b, int b_bf16, const void* c, int c_bf16, long long total, void* out, int out_bf16,
                             void* stream) {
  if (total <= 0) return 0;
  long long g = ((total >> 3) + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_add3, dim3((int)g), dim3(256), 0, (hipStream_t)stream, a, b, b_bf16, c, c_bf16, total, out, out_bf16);
  GD_LAUNCH_CHECK();
  return 0;
}
extern "C" int gdmae_add3(const float* a, const void* b, int b_bf16, const void* c, int c_bf16, long long total, float* out,
                          void* stream) {
  return gdmae_add3_to(a, b, b_bf16, c, c_bf16, total, out, 0, stream);
}

// ------------------------------------------------------------------------------------------------
// out = sum of k (<= 8) bf16 buffers of `total` elements, accumulated in fp32 and rounded once: the gradient of a map that feeds several
// consumers (the shared feature map under CenterHead's five branches, center_head.py:26-45) as ONE pass instead of the k - 1
// read-read-write additions the autograd engine issues (and one rounding instead of k - 1).
// ------------------------------------------------------------------------------------------------
struct GdSumSrc {
  const uint4* p[8];
};
__global__ __launch_bounds__(256) void k_sum_bf16(GdSumSrc S, int k, long long n16, uint4* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
    uint4 q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = S.p[j < k ? j : 0][i];          // unconditional (slot 0 again past k), masked below
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float m = j < k ? 1.f : 0.f;
      const unsigned w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[2 * e] = fmaf(m, __uint_as_float(w[e] << 16), a[2 * e]);
        a[2 * e + 1] = fmaf(m, __uint_as_float(w[e] & 0xFFFF0000u), a[2 * e + 1]);
      }
    }
    uint4 o;
    o.x = gd_pack_bf16(a[0], a[1]); o.y = gd_pack_bf16(a[2], a[3]); o.z = gd_pack_bf16(a[4], a[5]); o.w = gd_pack_bf16(a[6], a[7]);
    out[i] = o;
  }
}
extern "C" int gdmae_sum_bf16(const void* const* src, int k, long long total, void* out, void* stream) {
  GD_REQUIRE(k >= 1 && k <= 8 && total % 8 == 0, "sum_bf16: 1..8 sources, element count a multiple of 8");
  if (total <= 0) return 0;
  GdSumSrc S;
  for (int j = 0; j < 8; ++j) S.p[j] = (const uint4*)src[j < k ? j : 0];
  const long long n16 = total / 8;
  long long grid = (n16 + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(k_sum_bf16, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, S, k, n16, (uint4*)out);
  GD_LAUNCH_CHECK();
  return 0;
}
