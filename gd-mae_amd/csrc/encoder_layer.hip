// Native executor of one post-norm encoder layer: the whole forward or backward of
// EncoderLayer.forward (reference pcdet/models/model_utils/sst_basic_block.py:77-84, WindowAttention :22-54,
// cosine attention cosine_msa.py) is ONE C-ABI call that enqueues ~17 (forward) / ~40 (backward) kernels on the
// caller's stream: token-wise GEMMs through hipBLASLt (algorithms cached per problem shape), everything else the
// hand-written kernels of this library.  At 20-40 k tokens per stage each kernel runs for 5-40 us, so the layer is
// bound by how fast the host can enqueue: issuing it from an interpreter (one framework op per kernel) costs
// ~1.1 ms of host time per layer and direction against ~0.8 ms of GPU work; from here it costs tens of microseconds.
//
// Rows are padded to a multiple of 2048 (pad rows are zero or finite and never read as results) so that
//  * the GEMM shapes repeat from step to step (token counts differ per batch; hipBLASLt heuristics are cached), and
//  * every weight gradient g^T x is ONE batched split-K GEMM over equal K chunks + one reduce-accumulate kernel.
// Parameter gradients are ACCUMULATED into the caller's fp32 buffers (flat optimizer buffer or zeroed temporaries).
#include "../../include/gdmae_hip.h"
#include "common.h"
#include "dw_grouped.h"
#include "gemm.h"
#include "layer_tail.h"
#include <stdlib.h>

namespace {

constexpr long long kPad = 2048;
constexpr size_t kLtWorkspace = GD_LT_WORKSPACE;

__device__ inline float el_bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ inline unsigned short el_f2bf(float f) { return gd_to_bf16(f); }

// ------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------
struct ZeroJobs {
  void* p[8];
  unsigned long long n16[8];   // bytes / 16
  int count;
};
__global__ __launch_bounds__(256) void k_zero_regions(ZeroJobs z) {
  const uint4 zero = make_uint4(0, 0, 0, 0);
  for (int j = 0; j < z.count; ++j)
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < z.n16[j];
         i += (unsigned long long)gridDim.x * blockDim.x)
      ((uint4*)z.p[j])[i] = zero;
}

__global__ __launch_bounds__(256) void k_acc_vectors(AccJobs a) {
  // 16 consecutive elements x 16 slices of the partial rows per workgroup (64-byte row segments, 8 loads in flight per
  // thread), the slices meet in LDS in a fixed order
  __shared__ float sh[16][17];
  const int cl = threadIdx.x & 15, ps = threadIdx.x >> 4;
  int col = blockIdx.x * 16 + cl;
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

__global__ __launch_bounds__(256) void k_splitk_acc_jobs(SplitkJobs J) {
  __shared__ float4 sh[3][64];
  const int job = blockIdx.y;
  const long long P4 = J.P4[job];
  const int S = J.S[job];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long long i = blockIdx.x * 64ll + tx;
  if (blockIdx.x * 64ll >= P4) return;                 // uniform per workgroup
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < P4) {
    const int s0 = (S * ty) / 4, s1 = (S * (ty + 1)) / 4;
    const float4* p = (const float4*)J.part[job] + i;
#pragma unroll 4
    for (int s = s0; s < s1; ++s) {
      const float4 v = p[(long long)s * P4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (ty > 0) sh[ty - 1][tx] = acc;
  __syncthreads();
  if (ty == 0 && i < P4) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float4 v = sh[k][tx]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    float4* d = (float4*)J.dst[job] + i;
    const float4 o = *d;
    acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    *d = acc;
  }
}

__global__ __launch_bounds__(256) void k_layer_tail(TailJobs T) { tail_block(T, blockIdx.x, (int)blockIdx.y, (int)threadIdx.x); }

// exact (erf) GELU, 8 elements per thread
template <bool BF>
__global__ __launch_bounds__(256) void k_gelu_fwd(const void* __restrict__ h, void* __restrict__ out, long long total8) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    float v[8];
    if (BF) {
      const uint4 q = ((const uint4*)h)[i];
      const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = __uint_as_float(w[j] << 16); v[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u); }
    } else {
      const float4 a = ((const float4*)h)[2 * i], b = ((const float4*)h)[2 * i + 1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.5f * v[j] * (1.f + erff(v[j] * 0.70710678118654752440f));
    if (BF) {
      uint4 q;
      q.x = el_f2bf(v[0]) | ((unsigned)el_f2bf(v[1]) << 16);
      q.y = el_f2bf(v[2]) | ((unsigned)el_f2bf(v[3]) << 16);
      q.z = el_f2bf(v[4]) | ((unsigned)el_f2bf(v[5]) << 16);
      q.w = el_f2bf(v[6]) | ((unsigned)el_f2bf(v[7]) << 16);
      ((uint4*)out)[i] = q;
    } else {
      ((float4*)out)[2 * i] = make_float4(v[0], v[1], v[2], v[3]);
      ((float4*)out)[2 * i + 1] = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

// dh = dg * (Phi(h) + h * phi(h))
template <bool BF>
__global__ __launch_bounds__(256) void k_gelu_bwd(const void* __restrict__ dg, const void* __restrict__ h, void* __restrict__ dh,
                                                  long long total8) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
    float g[8], v[8];
    if (BF) {
      const uint4 q = ((const uint4*)dg)[i], r = ((const uint4*)h)[i];
      const unsigned wq[4] = {q.x, q.y, q.z, q.w}, wr[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        g[2 * j] = __uint_as_float(wq[j] << 16); g[2 * j + 1] = __uint_as_float(wq[j] & 0xFFFF0000u);
        v[2 * j] = __uint_as_float(wr[j] << 16); v[2 * j + 1] = __uint_as_float(wr[j] & 0xFFFF0000u);
      }
    } else {
      const float4 a = ((const float4*)dg)[2 * i], b = ((const float4*)dg)[2 * i + 1];
      const float4 c = ((const float4*)h)[2 * i], d = ((const float4*)h)[2 * i + 1];
      g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = b.x; g[5] = b.y; g[6] = b.z; g[7] = b.w;
      v[0] = c.x; v[1] = c.y; v[2] = c.z; v[3] = c.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float cdf = 0.5f * (1.f + erff(v[j] * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * v[j] * v[j]);
      g[j] = g[j] * (cdf + v[j] * pdf);
    }
    if (BF) {
      uint4 q;
      q.x = el_f2bf(g[0]) | ((unsigned)el_f2bf(g[1]) << 16);
      q.y = el_f2bf(g[2]) | ((unsigned)el_f2bf(g[3]) << 16);
      q.z = el_f2bf(g[4]) | ((unsigned)el_f2bf(g[5]) << 16);
      q.w = el_f2bf(g[6]) | ((unsigned)el_f2bf(g[7]) << 16);
      ((uint4*)dh)[i] = q;
    } else {
      ((float4*)dh)[2 * i] = make_float4(g[0], g[1], g[2], g[3]);
      ((float4*)dh)[2 * i + 1] = make_float4(g[4], g[5], g[6], g[7]);
    }
  }
}

// Bias gradients of the three GEMM groups of the layer in ONE launch pair: column sums of the first n rows of up to
// 4 matrices (job = blockIdx.y), fp32 partials per row chunk, then one wavefront per column accumulates into dst.
struct ColsumJobs {
  const void* x[4];
  float* dst[4];
  int C[4];
  int count;
};
template <bool BF>
__global__ __launch_bounds__(256) void k_colsum_jobs_partial(ColsumJobs J, long long n, float* __restrict__ part, int cmax) {
  extern __shared__ float sh[];
  const int job = blockIdx.y;
  const int C = J.C[job];
  const int cv = C >> 3;                       // C in {64, ..., 2048}, cv divides 256
  const int rpi = 256 / cv;
  const int tr = threadIdx.x / cv, c = (threadIdx.x % cv) << 3;
  const long long chunk = (n + gridDim.x - 1) / gridDim.x;
  const long long r0 = blockIdx.x * chunk, r1 = r0 + chunk < n ? r0 + chunk : n;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  for (long long r = r0 + tr; r < r1; r += rpi) {
    if (BF) {
      const uint4 q = *reinterpret_cast<const uint4*>((const unsigned short*)J.x[job] + r * C + c);
      const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) { acc[2 * j] += __uint_as_float(w[j] << 16); acc[2 * j + 1] += __uint_as_float(w[j] & 0xFFFF0000u); }
    } else {
      const float4 a = *reinterpret_cast<const float4*>((const float*)J.x[job] + r * C + c);
      const float4 b = *reinterpret_cast<const float4*>((const float*)J.x[job] + r * C + c + 4);
      acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w; acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sh[tr * C + c + j] = acc[j];
  __syncthreads();
  float* dstp = part + ((long long)job * gridDim.x + blockIdx.x) * cmax;
  for (int q = threadIdx.x; q < C; q += 256) {
    float s = 0.f;
    for (int rr = 0; rr < rpi; ++rr) s += sh[rr * C + q];
    dstp[q] = s;
  }
}
__global__ __launch_bounds__(256) void k_colsum_jobs_final(ColsumJobs J, const float* __restrict__ part, int nblk, int cmax) {
  const int lane = threadIdx.x & 63;
  int col = blockIdx.x * 4 + (threadIdx.x >> 6);     // global column over the concatenated jobs
  int job = 0;
  while (job < J.count && col >= J.C[job]) { col -= J.C[job]; ++job; }
  if (job >= J.count) return;
  double acc = 0.0;
  for (int b = lane; b < nblk; b += 64) acc += (double)part[((long long)job * nblk + b) * cmax + col];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
  if (lane == 0) J.dst[job][col] += (float)acc;
}

#define GD_TRY(x)          \
  do {                     \
    int rc_ = (x);         \
    if (rc_ != 0) return rc_; \
  } while (0)

struct Ctx {
  hipStream_t st;
  hipDataType ty;
  int es;
  void* lt_ws;
};

// row-major Y (n, m) = X (n, k) W^T (m, k) + b
int linear_fwd(const Ctx& c, const void* X, const void* W, const void* b, void* Y, long long n, int m, int k) {
  return gd_gemm(c.st, true, false, m, (int)n, k, W, k, X, k, Y, m, c.ty, c.ty, b, 1, 0, 0, 0, c.lt_ws, kLtWorkspace);
}
// row-major dX (n, k) = dY (n, m) W (m, k)
int linear_dx(const Ctx& c, const void* dY, const void* W, void* dX, long long n, int m, int k) {
  return gd_gemm(c.st, false, false, k, (int)n, m, W, k, dY, m, dX, k, c.ty, c.ty, nullptr, 1, 0, 0, 0, c.lt_ws, kLtWorkspace);
}
int splitk_for(long long n_pad, int m, int k) {
  const int tiles = ((m + 127) / 128) * ((k + 127) / 128);
  long long lim = n_pad / 256;
  // slices x output tiles ~ one workgroup per CU: more slices only add partial-sum traffic (S x m x n x 8 bytes per
  // weight gradient; 1024 -> 256 measured 477 -> 498 frames/s); GDMAE_SPLITK_TARGET overrides
  static const int target = getenv("GDMAE_SPLITK_TARGET") ? atoi(getenv("GDMAE_SPLITK_TARGET")) : 256;
  if (lim > target / tiles) lim = target / tiles;
  if (lim > 256) lim = 256;
  if (lim < 1) lim = 1;
  int S = 1;
  while ((long long)S * 2 <= lim && n_pad % (S * 2) == 0) S *= 2;
  return S;
}
// dW (m, k) fp32 += G^T (m, n_pad) X (n_pad, k): S batched partial products + reduce-accumulate
int linear_dw(const Ctx& c, const void* G, const void* X, float* dW, long long n_pad, int m, int k, float* part) {
  const int S = splitk_for(n_pad, m, k);
  const long long kc = n_pad / S;
  GD_TRY(gd_gemm(c.st, false, true, k, m, (int)kc, X, k, G, m, part, k, c.ty, HIP_R_32F, nullptr, S, kc * k, kc * m, (long long)m * k,
                 c.lt_ws, kLtWorkspace));
  GD_TRY(gd_splitk_acc(c.st, part, S, (long long)m * k, dW, 1));
  return 0;
}
// the same with the reduce deferred: the partial products go to their own region and the job is appended to J
int linear_dw_deferred(const Ctx& c, const void* G, const void* X, float* dW, long long n_pad, int m, int k, float* part,
                       SplitkJobs& J) {
  const int S = splitk_for(n_pad, m, k);
  const long long kc = n_pad / S;
  GD_TRY(gd_gemm(c.st, false, true, k, m, (int)kc, X, k, G, m, part, k, c.ty, HIP_R_32F, nullptr, S, kc * k, kc * m, (long long)m * k,
                 c.lt_ws, kLtWorkspace));
  GD_REQUIRE(J.count < 8 && ((long long)m * k) % 4 == 0, "splitk jobs");
  J.part[J.count] = part; J.dst[J.count] = dW; J.S[J.count] = S; J.P4[J.count] = (long long)m * k / 4;
  ++J.count;
  return 0;
}
int splitk_acc_jobs(const Ctx& c, const SplitkJobs& J) {
  long long mx = 0;
  for (int j = 0; j < J.count; ++j) mx = J.P4[j] > mx ? J.P4[j] : mx;
  if (J.count == 0 || mx == 0) return 0;
  hipLaunchKernelGGL(k_splitk_acc_jobs, dim3((unsigned)((mx + 63) / 64), J.count), dim3(256), 0, c.st, J);
  GD_LAUNCH_CHECK();
  return 0;
}
constexpr int kColsumBlocks = 512;
int colsum_jobs(const Ctx& c, ColsumJobs& J, long long n, int cmax, float* part) {
  size_t lds = 0;
  int ctot = 0;
  for (int j = 0; j < J.count; ++j) {
    GD_REQUIRE(J.C[j] % 8 == 0 && 256 % (J.C[j] / 8) == 0 && J.C[j] <= cmax, "colsum_jobs: unsupported width");
    const size_t l = (size_t)(256 / (J.C[j] / 8)) * J.C[j] * sizeof(float);
    lds = l > lds ? l : lds;
    ctot += J.C[j];
  }
  const dim3 grid(kColsumBlocks, J.count);
  if (c.es == 2) hipLaunchKernelGGL((k_colsum_jobs_partial<true>), grid, dim3(256), lds, c.st, J, n, part, cmax);
  else hipLaunchKernelGGL((k_colsum_jobs_partial<false>), grid, dim3(256), lds, c.st, J, n, part, cmax);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_colsum_jobs_final, dim3((ctot + 3) / 4), dim3(256), 0, c.st, J, (const float*)part, kColsumBlocks, cmax);
  GD_LAUNCH_CHECK();
  return 0;
}
int gelu(const Ctx& c, bool fwd, const void* a, const void* h, void* out, long long total) {
  const long long t8 = total / 8;
  long long g = (t8 + 255) / 256;
  if (g > 16384) g = 16384;
  if (fwd) {
    if (c.es == 2) hipLaunchKernelGGL((k_gelu_fwd<true>), dim3((int)g), dim3(256), 0, c.st, h, out, t8);
    else hipLaunchKernelGGL((k_gelu_fwd<false>), dim3((int)g), dim3(256), 0, c.st, h, out, t8);
  } else {
    if (c.es == 2) hipLaunchKernelGGL((k_gelu_bwd<true>), dim3((int)g), dim3(256), 0, c.st, a, h, out, t8);
    else hipLaunchKernelGGL((k_gelu_bwd<false>), dim3((int)g), dim3(256), 0, c.st, a, h, out, t8);
  }
  GD_LAUNCH_CHECK();
  return 0;
}
int zero_regions(const Ctx& c, ZeroJobs& z) {
  if (z.count == 0) return 0;
  unsigned long long mx = 0;
  for (int j = 0; j < z.count; ++j) mx = z.n16[j] > mx ? z.n16[j] : mx;
  if (mx == 0) return 0;
  long long g = (long long)((mx + 255) / 256);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_zero_regions, dim3((int)g), dim3(256), 0, c.st, z);
  GD_LAUNCH_CHECK();
  return 0;
}
void add_zero(ZeroJobs& z, void* base, long long n, long long n_pad, long long row_bytes) {
  z.p[z.count] = (char*)base + n * row_bytes;
  z.n16[z.count] = (unsigned long long)((n_pad - n) * row_bytes / 16);
  ++z.count;
}

struct Saved {   // layout of the activation block kept between forward and backward
  char *xb, *xpb, *qk, *v, *o, *a, *x1, *st1, *x1b, *h, *gact, *f, *st2, *lse;
  size_t bytes;
};
Saved saved_layout(void* base, long long n_pad, int d, int ff, int es) {
  Saved s;
  size_t off = 0;
  auto take = [&](size_t b) { char* p = (char*)base + off; off += gd_align(b); return p; };
  const size_t rd = (size_t)n_pad * d, rf = (size_t)n_pad * ff;
  s.xb = take(rd * es); s.xpb = take(rd * es); s.qk = take(2 * rd * es); s.v = take(rd * es); s.o = take(rd * es);
  s.a = take(rd * es); s.x1 = take(rd * 4); s.st1 = take((size_t)n_pad * 8);
  s.x1b = es == 2 ? take(rd * es) : s.x1;
  s.h = take(rf * es); s.gact = take(rf * es); s.f = take(rd * es); s.st2 = take((size_t)n_pad * 8);
  s.lse = take((size_t)n_pad * (d / 16) * 4);      // log-sum-exp of the attention rows: (n, H) fp32, H <= d / 16
  s.bytes = off;
  return s;
}
struct Scratch {
  char *lt_ws, *dx1_res, *dfb, *s2, *dg, *dh, *dx1_b, *dx_res, *dab, *s1, *d_o, *dqk, *dv, *apart, *dtau, *dx_qk, *dx_v, *part, *ln_ws,
      *cs_part, *part_w[5], *ln_ws2, *ln_ws2b;
  size_t bytes;
};
Scratch scratch_layout(void* base, long long n_pad, int d, int ff, int es, int nhead) {
  const long long n_attn_items = n_pad * nhead;   // upper bound: every window holds at least one token
  Scratch s;
  size_t off = 0;
  auto take = [&](size_t b) { char* p = (char*)base + off; off += gd_align(b); return p; };
  const size_t rd = (size_t)n_pad * d, rf = (size_t)n_pad * ff;
  s.lt_ws = take(kLtWorkspace);
  s.dx1_res = take(rd * 4); s.dfb = es == 2 ? take(rd * es) : s.dx1_res; s.s2 = take((size_t)3 * d * 4);
  s.dg = take(rf * es); s.dh = take(rf * es); s.dx1_b = take(rd * es);
  s.dx_res = take(rd * 4); s.dab = es == 2 ? take(rd * es) : s.dx_res; s.s1 = take((size_t)3 * d * 4);
  s.d_o = take(rd * es); s.dqk = take(2 * rd * es); s.dv = take(rd * es);
  s.apart = take((size_t)(n_attn_items > 0 ? n_attn_items : 1) * 4); s.dtau = take(256);
  s.dx_qk = take(rd * es); s.dx_v = take(rd * es);
  size_t mk = (size_t)ff * d;
  if ((size_t)2 * d * d > mk) mk = (size_t)2 * d * d;
  s.part = take((size_t)256 * mk * 4);      // S * m * k <= 1024 tiles * 128 * 128 ... bounded by 256 * m * k
  {   // LayerNorm-backward partial rows: the row kernel writes <= 1024 of them, the fused GEMM epilogue one per 32-row workgroup
    size_t lnb = gdmae_add_layernorm_workspace_bytes(d);
    const size_t fused_rows = (size_t)(n_pad / 32) * 3 * d * sizeof(float);
    if (fused_rows > lnb) lnb = fused_rows;
    s.ln_ws = take(lnb);
    s.ln_ws2 = take(lnb);
    s.ln_ws2b = take(lnb);
  }
  {   // one split-K partial region per weight gradient of the backward (reduced together at the end of the layer)
    const int mk5[5][2] = {{d, ff}, {ff, d}, {d, d}, {2 * d, d}, {d, d}};
    int Sg = 0;
    if (gd_dw_group_supported(n_pad, d, ff)) {
      int tiles = 0;
      for (int i = 0; i < 5; ++i) tiles += (mk5[i][0] / 128) * (mk5[i][1] / 128);
      Sg = gd_dw_group_slices(n_pad, tiles);
    }
    for (int i = 0; i < 5; ++i) {
      int S = splitk_for(n_pad, mk5[i][0], mk5[i][1]);
      if (Sg > S) S = Sg;
      s.part_w[i] = take((size_t)S * mk5[i][0] * mk5[i][1] * sizeof(float));
    }
  }
  {
    const int cm = ff > 2 * d ? ff : 2 * d;
    s.cs_part = take((size_t)3 * kColsumBlocks * cm * sizeof(float));
  }
  s.bytes = off;
  return s;
}
long long pad_rows(long long n) { return (n + kPad - 1) / kPad * kPad; }

}  // namespace

int gd_add_layernorm_fwd_ex(const float* a, const void* b, int b_is_bf16, const float* gamma, const float* beta, long long n, int d,
                            float eps, float* y, float* stats, void* y_bf16, const float* pos_table, const int* tok_pos,
                            void* ypos_bf16, hipStream_t st);
int gd_add_layernorm_bwd_ex(const float* a, const void* b, int b_is_bf16, const float* gamma, const float* stats, const float* dy,
                            const void* dy2, int dy2_bf16, const void* dy3, int dy3_bf16, long long n, int d, float* dx,
                            void* dx_bf16, float* sums, void* workspace, hipStream_t st);
int gd_ln_partial_rows(long long n, int d);

// fused token GEMMs (tok_gemm.hip)
void gd_attn_timing_tokens(long long n);   // attention.hip: token count for the byte figure of the next attention entry
bool gd_tok_gemm_supported(int K, int N);
int gd_tok_gemm_plain(hipStream_t st, const void* X, const void* Wp, const void* bias, long long n_pad, int K, int N, void* out);
int gd_tok_gemm_qkv(hipStream_t st, const void* Xpos, const void* X, const void* Wp_qk, const void* Wp_v, const void* bias3,
                    long long n_pad, int d, void* qk, void* v);
int gd_tok_gemm_gelu(hipStream_t st, const void* X, const void* Wp, const void* bias, long long n_pad, int K, int N, void* h, void* gact);
int gd_tok_gemm_gelu_bwd(hipStream_t st, const void* dY, const void* Wp, const void* h, long long n_pad, int K, int N, void* dh, void* gact);
bool gd_tok_gemm_ffn_supported(int d, int ff);
int gd_tok_gemm_ffn(hipStream_t st, const void* X, const void* W1p, const void* b1, const void* W2p, const void* b2, long long n,
                    long long n_pad, int d, void* h, const float* res, const float* gamma, const float* beta, float eps, float* y,
                    float* stats, void* y_bf, const float* pos_table, const int* tok_pos, void* ypos_bf, void* f_out);
int gd_tok_gemm_rows(int N);
int gd_tok_gemm_ln_bwd(hipStream_t st, const void* X, const void* Wp, long long n, long long n_pad, int K, int N, const float* dy,
                       const void* dy2_bf, const float* ln_a, const void* ln_b_bf, const float* stats, const float* gamma, float* dx,
                       void* dx_bf, float* part);
int gd_tok_gemm_res_ln(hipStream_t st, const void* X, const void* Wp, const void* bias, long long n, long long n_pad, int K, int N,
                       const float* res, const float* gamma, const float* beta, float eps, float* y, float* stats, void* y_bf,
                       const float* pos_table, const int* tok_pos, void* ypos_bf, void* f_out, int y_cached);

// layer around its bytes (layer_fused.hip)
// layer in registers (layer_v3.hip)
int gd_layer_v3_rows();
size_t gd_layer_v3_fwd_stream_elems(int d, int ff);
int gd_layer_v3_fwd_pack_jobs(const float* Wo, const float* W1, const float* W2, int d, int ff, void* stream_img, long long* jobs);
int gd_layer_v3_fwd(hipStream_t st, int d, const void* o, const void* x, const void* Wstream, const void* bo, const void* b1, const void* b2,
                    const float* g1, const float* be1, const float* g2, const float* be2, float eps, long long n, long long n_pad, void* a,
                    void* x1, void* h, void* f, float* st1, float* st2, float* y, void* y_bf, void* ypos_bf, const float* pos_table,
                    const int* tok_pos, const void* res0, void* res_out);
bool gd_layer_fused_supported(int d, int ff);
int gd_layer_fused_rows(int d);
int gd_layer_fused_fwd(hipStream_t st, int d, const void* o, const void* x, const void* Wo, const void* W1, const void* W2, const void* bo,
                       const void* b1, const void* b2, const float* g1, const float* be1, const float* g2, const float* be2, float eps,
                       long long n, long long n_pad, void* a, void* x1, void* h, void* f, float* st1, float* st2, float* y, void* y_bf,
                       void* ypos_bf, const float* pos_table, const int* tok_pos, const void* res0, void* res_out, const void* Wqk_n = nullptr,
                       const void* Wv_n = nullptr, const void* bin_n = nullptr, void* qk_n = nullptr, void* v_n = nullptr);
int gd_layer_fused_bwd_ffn(hipStream_t st, int d, const void* df, const void* h, const void* x, const void* a, const float* st1, const float* g1,
                           const void* W2t, const void* W1t, const void* Wot, long long n, long long n_pad, void* dh, void* gact, void* da,
                           void* d_o, float* part);
int gd_layer_fused_bwd_in(hipStream_t st, int d, const void* dqk, const void* dv, const void* Wqkt, const void* Wvt, const void* dres, long long n,
                          long long n_pad, const void* ln_a, const void* ln_b, const float* stats, const float* gamma, void* dout, float* part,
                          float* dx, const void* dtop, void* dx_bf, const TailJobs* tail = nullptr);
int gd_layer_fused_ln2_top(hipStream_t st, int d, const float* dy, const void* dy_bf, const void* ln_a, const void* ln_b, const float* stats,
                           const float* gamma, long long n, long long n_pad, void* dout, float* part);
int gd_layer_fused_prep_bf(hipStream_t st, const void* x, const float* pos_table, const int* tok_pos, long long n, int d, void* xb, void* xpb);

namespace {
// packed weight image of a layer: element offsets (in bf16 elements) of the ten operands
struct Packed {
  const char *qk, *v, *o, *w1, *w2, *w2t, *w1t, *ot, *qkt, *vt;
  const char* f3;          // layer_v3.hip: the forward launch's weight stream (Wo | per hidden chunk W1[chunk] | W2[:, chunk])
  size_t bytes;
};
// GDMAE_LAYER_V3=1: the layer's forward launch on the in-register kernel of layer_v3.hip (experiment: measured slower); its
// weight stream is packed only then
static int layer_v3() {
  static int v = -1;
  if (v < 0) v = getenv("GDMAE_LAYER_V3") ? atoi(getenv("GDMAE_LAYER_V3")) : 0;
  return v;
}
static bool v3_shape(int d, int ff) { return layer_v3() && (d == 128 || d == 256) && ff == 2 * d; }
Packed packed_layout(const void* base, int d, int ff) {
  Packed p;
  size_t off = 0;
  auto take = [&](size_t elems) { const char* q = (const char*)base + off; off += gd_align(elems * 2); return q; };
  const size_t dd = (size_t)d * d, df = (size_t)d * ff;
  p.qk = take(2 * dd); p.v = take(dd); p.o = take(dd); p.w1 = take(df); p.w2 = take(df);
  p.w2t = take(df); p.w1t = take(df); p.ot = take(dd); p.qkt = take(2 * dd); p.vt = take(dd);
  p.f3 = v3_shape(d, ff) ? take(gd_layer_v3_fwd_stream_elems(d, ff)) : nullptr;
  p.bytes = off;
  return p;
}
bool use_fused(const gdmae_layer_args* a) {
  static const int off = getenv("GDMAE_TOKGEMM") ? atoi(getenv("GDMAE_TOKGEMM")) == 0 : 0;
  return !off && a->bf16 && a->packed != nullptr && a->d <= 256 && gd_tok_gemm_supported(a->d, a->ff) &&
         gd_tok_gemm_supported(a->ff, a->d) && gd_tok_gemm_supported(a->d, 2 * a->d) && gd_tok_gemm_supported(a->d, a->d) &&
         gd_tok_gemm_supported(2 * a->d, a->d);
}
bool use_grouped_dw(const gdmae_layer_args* a, long long n_pad) {
  static const int off = getenv("GDMAE_DWGROUP") ? atoi(getenv("GDMAE_DWGROUP")) == 0 : 0;
  return !off && a->bf16 && gd_dw_group_supported(n_pad, a->d, a->ff);
}
// feed-forward block as one forward launch (gelu(h) is then produced by the BACKWARD's GELU kernel, which runs before the grouped
// weight-gradient launch - not before the per-matrix one)
bool use_ffn(const gdmae_layer_args* a, long long n_pad) {
  static const int off = getenv("GDMAE_FFN") ? atoi(getenv("GDMAE_FFN")) == 0 : 0;
  return !off && use_fused(a) && use_grouped_dw(a, n_pad) && gd_tok_gemm_ffn_supported(a->d, a->ff);
}
// the stage as three fused launches per layer and direction around the attention, bf16 residual stream (layer_fused.hip);
// GDMAE_LAYER_V2=0: the launch-per-product sequence with the fp32 stream (A/B reference)
int g_layer_path = -1;      // -1: GDMAE_LAYER_V2 (default on), 0: launch-per-product, 1: fused (gdmae_encoder_set_layer_path)
bool stage_v2(const gdmae_layer_args* layers, int n_layers) {
  static const int env_off = getenv("GDMAE_LAYER_V2") ? atoi(getenv("GDMAE_LAYER_V2")) == 0 : 0;
  if (g_layer_path == 0 || (g_layer_path < 0 && env_off)) return false;
  for (int i = 0; i < n_layers; ++i)
    if (!(use_ffn(&layers[i], pad_rows(layers[i].n)) && gd_layer_fused_supported(layers[i].d, layers[i].ff))) return false;
  return true;
}
}  // namespace

extern "C" int gdmae_encoder_set_layer_path(int path) {
  GD_REQUIRE(path >= -1 && path <= 1, "encoder_set_layer_path: -1 (environment default), 0 or 1");
  g_layer_path = path;
  return 0;
}

extern "C" int gdmae_encoder_stage_fused(const gdmae_layer_args* layers, int n_layers) {
  return (layers != nullptr && n_layers >= 1 && stage_v2(layers, n_layers)) ? 1 : 0;
}

extern "C" size_t gdmae_layer_packed_bytes(int d, int ff) { return packed_layout(nullptr, d, ff).bytes; }

extern "C" int gdmae_layer_pack_jobs(const float* Win, const float* Wo, const float* W1, const float* W2, int d, int ff, void* packed,
                                     long long* jobs) {
  GD_REQUIRE(d % 32 == 0 && ff % 32 == 0, "layer_pack_jobs: d and ff must be multiples of 32");
  const Packed p = packed_layout(packed, d, ff);
  // {src, dst, M, K, ld, transpose}: A (M, K) = src (M, ld)  or, transposed, A[r][c] = src[c * ld + r]
  const long long J[10][6] = {
      {(long long)Win, (long long)p.qk, 2 * d, d, d, 0},
      {(long long)(Win + (size_t)2 * d * d), (long long)p.v, d, d, d, 0},
      {(long long)Wo, (long long)p.o, d, d, d, 0},
      {(long long)W1, (long long)p.w1, ff, d, d, 0},
      {(long long)W2, (long long)p.w2, d, ff, ff, 0},
      {(long long)W2, (long long)p.w2t, ff, d, ff, 1},                      // dg  = df  W2   : A = W2^T (ff, d)
      {(long long)W1, (long long)p.w1t, d, ff, d, 1},                       // dx1 = dh  W1   : A = W1^T (d, ff)
      {(long long)Wo, (long long)p.ot, d, d, d, 1},                         // do  = da  Wo   : A = Wo^T
      {(long long)Win, (long long)p.qkt, d, 2 * d, d, 1},                   // dxqk = dqk Win[:2d] : A = Win[:2d]^T (d, 2d)
      {(long long)(Win + (size_t)2 * d * d), (long long)p.vt, d, d, d, 1},  // dxv = dv Win[2d:]
  };
  for (int i = 0; i < 10; ++i)
    for (int k = 0; k < 6; ++k) jobs[i * 6 + k] = J[i][k];
  if (p.f3) gd_layer_v3_fwd_pack_jobs(Wo, W1, W2, d, ff, (void*)p.f3, jobs + 60);
  return 0;
}

extern "C" int gdmae_layer_pack_job_count(int d, int ff) { return 10 + (v3_shape(d, ff) ? 1 + 2 * (ff / 128) : 0); }

extern "C" int gdmae_encoder_layer_bytes(long long n, int d, int ff, int nhead, int bf16, size_t* saved_bytes,
                                         size_t* fwd_scratch_bytes, size_t* bwd_scratch_bytes) {
  const long long n_pad = pad_rows(n);
  *saved_bytes = saved_layout(nullptr, n_pad, d, ff, bf16 ? 2 : 4).bytes;
  *fwd_scratch_bytes = gd_align(kLtWorkspace);
  *bwd_scratch_bytes = scratch_layout(nullptr, n_pad, d, ff, bf16 ? 2 : 4, nhead).bytes;
  return 0;
}

// One layer forward.  `next` (bf16 mode only): the layer that consumes this one's output - its q/k and v inputs
// (x, x + pos in bf16) are then written by this layer's second LayerNorm pass; `prepped`: this layer's inputs were
// written that way by the previous layer (its gdmae_prep_tokens pass is skipped).
static int layer_fwd(const gdmae_layer_args* a, const gdmae_layer_args* next, bool prepped, void* stream) {
  GD_REQUIRE(a->n > 0 && a->d % 8 == 0 && a->ff % 8 == 0, "encoder layer: bad sizes");
  GD_REQUIRE(a->n_levels >= 1 && a->n_levels <= 4, "encoder layer: 1..4 window levels");
  const long long n = a->n, n_pad = pad_rows(n);
  const int d = a->d, ff = a->ff, es = a->bf16 ? 2 : 4;
  Ctx c{(hipStream_t)stream, a->bf16 ? HIP_R_16BF : HIP_R_32F, es, a->scratch};
  Saved s = saved_layout(a->saved, n_pad, d, ff, es);
  // pad rows of every GEMM operand: zero
  ZeroJobs z;
  z.count = 0;
  add_zero(z, s.xb, n, n_pad, (long long)d * es);
  add_zero(z, s.xpb, n, n_pad, (long long)d * es);
  add_zero(z, s.o, n, n_pad, (long long)d * es);
  add_zero(z, s.x1b, n, n_pad, (long long)d * es);   // fp32 mode: x1 itself
  // (the grouped weight-gradient kernel does not read rows >= n, and every other consumer is row-wise)
  if (!use_grouped_dw(a, n_pad)) GD_TRY(zero_regions(c, z));
  if (a->bf16) {
    if (!prepped) GD_TRY(gdmae_prep_tokens(a->x, a->pos_table, a->tok_pos, n, d, s.xb, s.xpb, 1, stream));
  } else {
    GD_CHECK(hipMemcpyAsync(s.xb, a->x, (size_t)n * d * 4, hipMemcpyDeviceToDevice, c.st));
    GD_TRY(gdmae_prep_tokens(a->x, a->pos_table, a->tok_pos, n, d, nullptr, s.xpb, 0, stream));
  }
  const char* Win = (const char*)a->Win;
  const char* bin = (const char*)a->bin;
  const bool fused = use_fused(a);
  const Packed pk = packed_layout(a->packed, d, ff);
  if (fused) {
    GD_TRY(gd_tok_gemm_qkv(c.st, s.xpb, s.xb, pk.qk, pk.v, bin, n_pad, d, s.qk, s.v));
  } else {
    GD_TRY(linear_fwd(c, s.xpb, Win, bin, s.qk, n_pad, 2 * d, d));
    GD_TRY(linear_fwd(c, s.xb, Win + (size_t)2 * d * d * es, bin + (size_t)2 * d * es, s.v, n_pad, d, d));
  }
  gd_attn_timing_tokens(n);
  GD_TRY(gdmae_window_attention_levels_fwd(s.qk, s.v, s.o, a->bf16, a->csr_tok, a->win_start, a->win_len, a->n_levels, a->n_win,
                                           a->max_tokens, d, a->nhead, a->tau, a->tau_min, (float*)s.lse, stream));
  if (fused) {
    // out-projection + residual + LayerNorm 1; linear1 + GELU; linear2 + residual + LayerNorm 2 (+ the next layer's
    // q/k/v operands): three launches for what is eight in the unfused sequence
    GD_TRY(gd_tok_gemm_res_ln(c.st, s.o, pk.o, a->bo, n, n_pad, d, d, a->x, a->g1, a->be1, a->eps, (float*)s.x1, (float*)s.st1, s.x1b,
                              nullptr, nullptr, nullptr, s.a, 1));      // x1 is the residual operand of the very next launch
    // linear1 + GELU + linear2 + residual + LayerNorm 2 in one launch when the widths allow it: gelu(h) never leaves the CU (the
    // backward's GELU kernel writes it into s.gact for the weight gradient of linear2)
    const bool ffn1 = use_ffn(a, n_pad);
    if (!ffn1) GD_TRY(gd_tok_gemm_gelu(c.st, s.x1b, pk.w1, a->b1, n_pad, d, ff, s.h, s.gact));
    void *y_bf = nullptr, *ypos_bf = nullptr;
    const float* ptab = nullptr;
    const int* tpos = nullptr;
    if (next) {
      Saved sn = saved_layout(next->saved, n_pad, d, ff, es);
      y_bf = sn.xb; ypos_bf = sn.xpb; ptab = next->pos_table; tpos = next->tok_pos;
    }
    if (ffn1)
      GD_TRY(gd_tok_gemm_ffn(c.st, s.x1b, pk.w1, a->b1, pk.w2, a->b2, n, n_pad, d, s.h, (const float*)s.x1, a->g2, a->be2, a->eps, a->y,
                             (float*)s.st2, y_bf, ptab, tpos, ypos_bf, s.f));
    else
      GD_TRY(gd_tok_gemm_res_ln(c.st, s.gact, pk.w2, a->b2, n, n_pad, ff, d, (const float*)s.x1, a->g2, a->be2, a->eps, a->y,
                                (float*)s.st2, y_bf, ptab, tpos, ypos_bf, s.f, 0));
    return 0;
  }
  GD_TRY(linear_fwd(c, s.o, a->Wo, a->bo, s.a, n_pad, d, d));
  GD_TRY(gdmae_add_layernorm_fwd(a->x, s.a, a->bf16, a->g1, a->be1, n, d, a->eps, (float*)s.x1, (float*)s.st1,
                                 a->bf16 ? s.x1b : nullptr, stream));
  GD_TRY(linear_fwd(c, s.x1b, a->W1, a->b1, s.h, n_pad, ff, d));
  GD_TRY(gelu(c, true, nullptr, s.h, s.gact, n_pad * ff));
  GD_TRY(linear_fwd(c, s.gact, a->W2, a->b2, s.f, n_pad, d, ff));
  if (next && a->bf16) {
    Saved sn = saved_layout(next->saved, n_pad, d, ff, es);
    GD_TRY(gd_add_layernorm_fwd_ex((const float*)s.x1, s.f, 1, a->g2, a->be2, n, d, a->eps, a->y, (float*)s.st2, sn.xb, next->pos_table,
                                   next->tok_pos, sn.xpb, c.st));
  } else {
    GD_TRY(gdmae_add_layernorm_fwd((const float*)s.x1, s.f, a->bf16, a->g2, a->be2, n, d, a->eps, a->y, (float*)s.st2, nullptr, stream));
  }
  return 0;
}

extern "C" int gdmae_encoder_layer_fwd(const gdmae_layer_args* a, void* stream) { return layer_fwd(a, nullptr, false, stream); }

// L consecutive layers of one stage (same n, d, ff; layer i + 1 reads layer i's y): one call, and in bf16 mode the
// prep_tokens pass of layers 1.. is folded into the previous layer's second LayerNorm.
// The stage with the fused layer launches (layer_fused.hip): per layer q/k/v projections, attention, ONE launch for everything
// behind it.  The residual stream between the layers is the bf16 row the next layer's v projection reads anyway (xb of its saved
// block); only the last layer writes the fp32 rows the caller sees.
// GDMAE_QKV_RIDES=1: the in-projection of layers 1 ... of a stage as two more products of the launch that closes the layer below
// (k_layer_fwd<D, true>: the rows are still in LDS; bit-identical q / k / v).  Measured, not the default: 9 launches per step less, but the
// step is 7.69 - 7.70 ms against 7.64 - 7.67 with the launches (4 frames: 4.83 against 4.79 - 4.82) - the phases of a row-tile workgroup
// run one after the other at two workgroups per CU, so the products cost there what they cost in a launch of their own.
static bool qkv_rides(const gdmae_layer_args* a) {
  static int v = -1;
  if (v < 0) v = getenv("GDMAE_QKV_RIDES") ? atoi(getenv("GDMAE_QKV_RIDES")) : 0;
  return v != 0 && a->bin != nullptr;
}
static int stage_fwd_v2(const gdmae_layer_args* layers, int n_layers, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  for (int i = 0; i < n_layers; ++i) {
    const gdmae_layer_args* a = &layers[i];
    GD_REQUIRE(a->n > 0 && a->n_levels >= 1 && a->n_levels <= 4, "encoder layer: bad sizes");
    const long long n = a->n, n_pad = pad_rows(n);
    const int d = a->d, ff = a->ff;
    Saved s = saved_layout(a->saved, n_pad, d, ff, 2);
    const Packed pk = packed_layout(a->packed, d, ff);
    if (i == 0) {
      if (a->x_bf16) GD_TRY(gd_layer_fused_prep_bf(st, a->x, a->pos_table, a->tok_pos, n, d, s.xb, s.xpb));
      else GD_TRY(gdmae_prep_tokens(a->x, a->pos_table, a->tok_pos, n, d, s.xb, s.xpb, 1, stream));
    }
    // layers 1 ...: q | k and v were produced by the launch that closed the layer below (its rows were still in LDS)
    if (i == 0 || !qkv_rides(&layers[i])) GD_TRY(gd_tok_gemm_qkv(st, s.xpb, s.xb, pk.qk, pk.v, a->bin, n_pad, d, s.qk, s.v));
    gd_attn_timing_tokens(n);
    GD_TRY(gdmae_window_attention_levels_fwd(s.qk, s.v, s.o, 1, a->csr_tok, a->win_start, a->win_len, a->n_levels, a->n_win, a->max_tokens, d,
                                             a->nhead, a->tau, a->tau_min, (float*)s.lse, stream));
    const gdmae_layer_args* next = i + 1 < n_layers ? &layers[i + 1] : nullptr;
    void *y_bf = nullptr, *ypos_bf = nullptr, *qk_n = nullptr, *v_n = nullptr;
    const char *wqk_n = nullptr, *wv_n = nullptr;
    if (next) {
      Saved sn = saved_layout(next->saved, n_pad, d, ff, 2);
      y_bf = sn.xb; ypos_bf = sn.xpb;
      if (qkv_rides(next)) {
        const Packed pn = packed_layout(next->packed, d, ff);
        wqk_n = pn.qk; wv_n = pn.v; qk_n = sn.qk; v_n = sn.v;
      }
    }
    if (layer_v3() && pk.f3 && !wqk_n) {
      GD_TRY(gd_layer_v3_fwd(st, d, s.o, s.xb, pk.f3, a->bo, a->b1, a->b2, a->g1, a->be1, a->g2, a->be2, a->eps, n, n_pad, s.a, s.x1b, s.h, s.f,
                             (float*)s.st1, (float*)s.st2, (next || a->res_out) ? nullptr : a->y, y_bf, ypos_bf,
                             next ? next->pos_table : nullptr, next ? next->tok_pos : nullptr,
                             (!next && a->res_out) ? saved_layout(layers[0].saved, n_pad, d, ff, 2).xb : nullptr, next ? nullptr : a->res_out));
      continue;
    }
    GD_TRY(gd_layer_fused_fwd(st, d, s.o, s.xb, pk.o, pk.w1, pk.w2, a->bo, a->b1, a->b2, a->g1, a->be1, a->g2, a->be2, a->eps, n, n_pad, s.a,
                              s.x1b, s.h, s.f, (float*)s.st1, (float*)s.st2, (next || a->res_out) ? nullptr : a->y, y_bf, ypos_bf,
                              next ? next->pos_table : nullptr, next ? next->tok_pos : nullptr,
                              (!next && a->res_out) ? saved_layout(layers[0].saved, n_pad, d, ff, 2).xb : nullptr, next ? nullptr : a->res_out,
                              wqk_n, wv_n, wqk_n ? next->bin : nullptr, qk_n, v_n));
  }
  return 0;
}

extern "C" int gdmae_encoder_stage_fwd(const gdmae_layer_args* layers, int n_layers, void* stream) {
  GD_REQUIRE(n_layers >= 1, "encoder stage: no layers");
  for (int i = 0; i < n_layers; ++i)
    GD_REQUIRE(i == 0 || (layers[i].x == layers[i - 1].y && layers[i].n == layers[0].n && layers[i].d == layers[0].d &&
                          layers[i].ff == layers[0].ff && layers[i].bf16 == layers[0].bf16),
               "encoder stage: layers must chain (x[i] = y[i-1]) and share n, d, ff, dtype");
  if (stage_v2(layers, n_layers)) return stage_fwd_v2(layers, n_layers, stream);
  GD_REQUIRE(n_layers == 1 || (uintptr_t)layers[1].x >= 4096, "encoder stage: chained by tokens (the fused path's convention) but the fused path is off");
  GD_REQUIRE(!layers[0].x_bf16 && !layers[n_layers - 1].res_out, "encoder stage: bf16 input rows / the folded block residual need the fused path");
  for (int i = 0; i < n_layers; ++i) {
    GD_REQUIRE(i == 0 || (layers[i].x == layers[i - 1].y && layers[i].n == layers[0].n && layers[i].d == layers[0].d &&
                          layers[i].ff == layers[0].ff && layers[i].bf16 == layers[0].bf16),
               "encoder stage: layers must chain (x[i] = y[i-1]) and share n, d, ff, dtype");
    GD_TRY(layer_fwd(&layers[i], i + 1 < n_layers ? &layers[i + 1] : nullptr, i > 0 && layers[i].bf16, stream));
  }
  return 0;
}


namespace {
// The five weight gradients + three bias column sums of a layer as ONE grouped launch, then the tail launch that reduces their
// split-K partials together with the LayerNorm partial rows (nb1 / nb2 rows of LayerNorm 1 / 2) and the attention's dtau partials.
// ride != null: the tail is not launched here - its job table is handed back and rides along with the layer's in-projection launch
// (gd_layer_fused_bwd_in; ln2: the LayerNorm-2 partial rows of THIS layer, which that launch must not be writing to)
int grouped_dw_and_tail(const gdmae_layer_args* a, const Saved& s, const Scratch& w, const Ctx& c, long long n, long long n_pad, int nb1,
                        int nb2, long long pbase, TailJobs* ride = nullptr, const float* ln2 = nullptr) {
  const int d = a->d, ff = a->ff;
  SplitkJobs SJ;
  SJ.count = 0;
  GdDwGroup Gp;
  Gp.n_jobs = 5;
  float* cp = (float*)w.cs_part;                       // (3, S, cmax) column-sum partials
  const int cmax = ff > 2 * d ? ff : 2 * d;
  int tiles = 0;
  const int mk5[5][2] = {{d, ff}, {ff, d}, {d, d}, {2 * d, d}, {d, d}};
  for (int i = 0; i < 5; ++i) tiles += (mk5[i][0] / 128) * (mk5[i][1] / 128);
  const int S = gd_dw_group_slices(n_pad, tiles);
  Gp.job[0] = GdDwJob{w.dfb, s.gact, d, ff, (float*)w.part_w[0], nullptr, 0};
  Gp.job[1] = GdDwJob{w.dh, s.x1b, ff, d, (float*)w.part_w[1], cp, 0};
  Gp.job[2] = GdDwJob{w.dab, s.o, d, d, (float*)w.part_w[2], nullptr, 0};
  Gp.job[3] = GdDwJob{w.dqk, s.xpb, 2 * d, d, (float*)w.part_w[3], cp + (size_t)S * cmax, 0};
  Gp.job[4] = GdDwJob{w.dv, s.xb, d, d, (float*)w.part_w[4], cp + (size_t)2 * S * cmax, 0};
  GD_TRY(gd_dw_grouped(c.st, Gp, n_pad, n));
  GD_REQUIRE(Gp.S == S, "dw_grouped: slice count");
  float* dW[5] = {a->dW2, a->dW1, a->dWo, a->dWin, a->dWin + (size_t)2 * d * d};
  for (int i = 0; i < 5; ++i) {
    SJ.part[i] = Gp.job[i].part; SJ.dst[i] = dW[i]; SJ.S[i] = S; SJ.P4[i] = (long long)mk5[i][0] * mk5[i][1] / 4;
  }
  // bias column sums: same reduce (partials (S, M) -> dst (M))
  const int cj[3] = {1, 3, 4};
  float* db[3] = {a->db1, a->dbin, a->dbin + 2 * d};
  for (int i = 0; i < 3; ++i) {
    SJ.part[5 + i] = Gp.job[cj[i]].colpart; SJ.dst[5 + i] = db[i]; SJ.S[5 + i] = S; SJ.P4[5 + i] = mk5[cj[i]][0] / 4;
  }
  SJ.count = 8;
  // LayerNorm / bias gradients from the per-workgroup partial rows {dgamma, dbeta, column sums of dx}; dtau from the attention partials
  AccJobs j;
  const float* p1 = (const float*)w.ln_ws;    // LayerNorm 1: dg1, dbe1, bias gradient of the out-projection
  const float* p2 = ln2 ? ln2 : (const float*)w.ln_ws2;   // LayerNorm 2: dg2, dbe2, bias gradient of the second FFN linear
  float* dst[6] = {a->dg1, a->dbe1, a->dbo, a->dg2, a->dbe2, a->db2};
  const float* src[6] = {p1, p1 + d, p1 + 2 * d, p2, p2 + d, p2 + 2 * d};
  int cols = 0;
  for (int i = 0; i < 6; ++i) {
    j.dst[i] = dst[i]; j.src[i] = src[i]; j.len[i] = d; j.nblk[i] = i < 3 ? nb1 : nb2; j.stride[i] = 3 * d;
    cols += d;
  }
  j.count = 6;
  TailJobs T;
  T.J = SJ;
  T.a = j;
  T.tau_part = (const float*)w.apart; T.n_part = pbase; T.tau = a->tau; T.tau_min = a->tau_min; T.dtau = a->dtau;
  if (ride) {
    *ride = T;
    return 0;
  }
  const long long gx = tail_grid_x(T);
  double tail_bytes = 4.0 * pbase;
  for (int q = 0; q < SJ.count; ++q) tail_bytes += 16.0 * SJ.P4[q] * (SJ.S[q] + 2);
  for (int q = 0; q < j.count; ++q) tail_bytes += 4.0 * j.len[q] * (j.nblk[q] + 2);
  GdTimed timed(GD_T_LAYER_TAIL, c.st, tail_bytes);
  hipLaunchKernelGGL(k_layer_tail, dim3((unsigned)gx, SJ.count + 2), dim3(256), 0, c.st, T);
  GD_LAUNCH_CHECK();
  return 0;
}
}  // namespace

// One layer backward.  `upstream3`: the upstream gradient is the sum of three tensors left in `scratch` by the
// backward of the NEXT layer (its residual-stream gradient, dx_qk, dx_v - that layer skipped its add3 pass) instead of
// a->dy; `defer_add3`: leave this layer's own three pieces in scratch for the previous layer the same way.
// prev: the layer below (processed next) whose LayerNorm-2 backward is fused behind this layer's last input-gradient GEMM, or
// null; ln2_done: this layer's own LayerNorm-2 backward was already produced that way by the layer above
static int layer_bwd(const gdmae_layer_args* a, bool upstream3, bool defer_add3, const gdmae_layer_args* prev, bool ln2_done,
                     void* stream) {
  GD_REQUIRE(a->n > 0 && a->d % 8 == 0 && a->ff % 8 == 0, "encoder layer: bad sizes");
  const long long n = a->n, n_pad = pad_rows(n);
  const int d = a->d, ff = a->ff, es = a->bf16 ? 2 : 4;
  long long items = 0;
  for (int l = 0; l < a->n_levels; ++l) items += (long long)a->n_win[l] * a->nhead;
  Saved s = saved_layout(a->saved, n_pad, d, ff, es);
  Scratch w = scratch_layout(a->scratch, n_pad, d, ff, es, a->nhead);
  Ctx c{(hipStream_t)stream, a->bf16 ? HIP_R_16BF : HIP_R_32F, es, w.lt_ws};
  ZeroJobs z;
  z.count = 0;
  add_zero(z, w.dfb, n, n_pad, (long long)d * es);    // fp32 mode: dx1_res / dx_res themselves
  add_zero(z, w.dab, n, n_pad, (long long)d * es);
  add_zero(z, w.dqk, n, n_pad, (long long)2 * d * es);
  add_zero(z, w.dv, n, n_pad, (long long)d * es);
  z.p[z.count] = w.apart;
  z.n16[z.count] = (unsigned long long)(gd_align((size_t)(items > 0 ? items : 1) * 4) / 16);
  ++z.count;
  // bf16 rows: the five weight gradients (and the three bias column sums that are not LayerNorm by-products) of the layer
  // are ONE launch of the hand-written TN kernel at the end (dw_grouped.hip), which ignores rows >= n, and the attention
  // kernels fill every partial slot of their level: nothing to clear.  fp32 rows / odd sizes: library split-K GEMMs over
  // zero-padded operands.
  const bool grouped = use_grouped_dw(a, n_pad);
  if (!grouped) GD_TRY(zero_regions(c, z));
  // ---- LN2 and FFN
  if (ln2_done) {
    // dx1_res / dfb and the partial rows in ln_ws2 were written by the epilogue of the layer above's v-projection gradient
  } else if (upstream3)   // dx_res / dx_qk / dx_v of the next layer are consumed here, before anything overwrites them
    GD_TRY(gd_add_layernorm_bwd_ex((const float*)s.x1, s.f, a->bf16, a->g2, (const float*)s.st2, (const float*)w.dx_res, w.dx_qk, a->bf16,
                                   w.dx_v, a->bf16, n, d, (float*)w.dx1_res, a->bf16 ? w.dfb : nullptr, nullptr, w.ln_ws2, c.st));
  else
    GD_TRY(gd_add_layernorm_bwd_ex((const float*)s.x1, s.f, a->bf16, a->g2, (const float*)s.st2, a->dy, nullptr, 0, nullptr, 0, n, d,
                                   (float*)w.dx1_res, a->bf16 ? w.dfb : nullptr, nullptr, w.ln_ws2, c.st));
  SplitkJobs SJ;
  SJ.count = 0;
  if (!grouped) GD_TRY(linear_dw_deferred(c, w.dfb, s.gact, a->dW2, n_pad, d, ff, (float*)w.part_w[0], SJ));
  const bool fused = use_fused(a);
  const Packed pk = packed_layout(a->packed, d, ff);
  if (fused) {
    // dh = (dfb W2) * gelu'(h); after the one-launch forward also gelu(h), the operand of linear2's weight gradient
    GD_TRY(gd_tok_gemm_gelu_bwd(c.st, w.dfb, pk.w2t, s.h, n_pad, d, ff, w.dh, use_ffn(a, n_pad) ? s.gact : nullptr));
  } else {
    GD_TRY(linear_dx(c, w.dfb, a->W2, w.dg, n_pad, d, ff));
    GD_TRY(gelu(c, false, w.dg, s.h, w.dh, n_pad * ff));
  }
  if (!grouped) GD_TRY(linear_dw_deferred(c, w.dh, s.x1b, a->dW1, n_pad, ff, d, (float*)w.part_w[1], SJ));
  // ---- LN1 (gradient = residual branch + FFN branch) and out-projection
  const bool fuse_ln = fused && grouped;       // LayerNorm backward as the epilogue of the GEMM that produces its last gradient piece
  if (fuse_ln) {
    GD_TRY(gd_tok_gemm_ln_bwd(c.st, w.dh, pk.w1t, n, n_pad, ff, d, (const float*)w.dx1_res, nullptr, a->x, s.a, (const float*)s.st1, a->g1,
                              (float*)w.dx_res, w.dab, (float*)w.ln_ws));
  } else {
    if (fused) GD_TRY(gd_tok_gemm_plain(c.st, w.dh, pk.w1t, nullptr, n_pad, ff, d, w.dx1_b));
    else GD_TRY(linear_dx(c, w.dh, a->W1, w.dx1_b, n_pad, ff, d));
    GD_TRY(gd_add_layernorm_bwd_ex(a->x, s.a, a->bf16, a->g1, (const float*)s.st1, (const float*)w.dx1_res, w.dx1_b, a->bf16, nullptr, 0,
                                   n, d, (float*)w.dx_res, a->bf16 ? w.dab : nullptr, nullptr, w.ln_ws, c.st));
  }
  if (!grouped) GD_TRY(linear_dw_deferred(c, w.dab, s.o, a->dWo, n_pad, d, d, (float*)w.part_w[2], SJ));
  if (fused) GD_TRY(gd_tok_gemm_plain(c.st, w.dab, pk.ot, nullptr, n_pad, d, d, w.d_o));
  else GD_TRY(linear_dx(c, w.dab, a->Wo, w.d_o, n_pad, d, d));
  // ---- attention
  long long pbase = 0;
  for (int l = 0; l < a->n_levels; ++l) pbase += (long long)a->n_win[l] * a->nhead;
  gd_attn_timing_tokens(n);
  GD_TRY(gdmae_window_attention_levels_bwd(s.qk, s.v, w.d_o, w.dqk, w.dv, a->bf16, (float*)w.apart, a->csr_tok, a->win_start, a->win_len,
                                           a->n_levels, a->n_win, a->max_tokens, d, a->nhead, a->tau, a->tau_min, s.o, (const float*)s.lse, stream));
  if (!grouped) GD_TRY(gdmae_sum_partials_gated((const float*)w.apart, pbase, 1.f, (float*)w.dtau, a->tau, a->tau_min, stream));
  const char* Win = (const char*)a->Win;
  const int nb_rows = gd_ln_partial_rows(n, d), nb_fused = (int)(n_pad / gd_tok_gemm_rows(d));
  const int nb1 = fuse_ln ? nb_fused : nb_rows, nb2 = ln2_done ? nb_fused : nb_rows;
  if (grouped) {
    GD_TRY(grouped_dw_and_tail(a, s, w, c, n, n_pad, nb1, nb2, pbase));
  } else {
    GD_TRY(linear_dw_deferred(c, w.dqk, s.xpb, a->dWin, n_pad, 2 * d, d, (float*)w.part_w[3], SJ));
    GD_TRY(linear_dw_deferred(c, w.dv, s.xb, a->dWin + (size_t)2 * d * d, n_pad, d, d, (float*)w.part_w[4], SJ));
    GD_TRY(splitk_acc_jobs(c, SJ));                      // the five split-K reduces of the layer as one launch
    ColsumJobs J;
    J.count = 3;
    J.x[0] = w.dh;  J.dst[0] = a->db1;           J.C[0] = ff;
    J.x[1] = w.dqk; J.dst[1] = a->dbin;          J.C[1] = 2 * d;
    J.x[2] = w.dv;  J.dst[2] = a->dbin + 2 * d;  J.C[2] = d;
    GD_TRY(colsum_jobs(c, J, n, ff > 2 * d ? ff : 2 * d, (float*)w.cs_part));
    // ---- LayerNorm / bias / temperature gradients from the LayerNorm backward partial rows
    AccJobs j;
    const float* p1 = (const float*)w.ln_ws;
    const float* p2 = (const float*)w.ln_ws2;
    float* dst[7] = {a->dg1, a->dbe1, a->dbo, a->dg2, a->dbe2, a->db2, a->dtau};
    const float* src[7] = {p1, p1 + d, p1 + 2 * d, p2, p2 + d, p2 + 2 * d, (const float*)w.dtau};
    int cols = 0;
    for (int i = 0; i < 7; ++i) {
      j.dst[i] = dst[i]; j.src[i] = src[i]; j.len[i] = i < 6 ? d : 1; j.nblk[i] = i < 3 ? nb1 : (i < 6 ? nb2 : 0); j.stride[i] = 3 * d;
      cols += j.len[i];
    }
    j.count = 7;
    hipLaunchKernelGGL(k_acc_vectors, dim3((cols + 15) / 16), dim3(256), 0, c.st, j);
    GD_LAUNCH_CHECK();
  }

  // ---- input gradient of the q/k and v projections (after the tail launch: the fused epilogue below re-uses ln_ws2)
  if (fused) {
    GD_TRY(gd_tok_gemm_plain(c.st, w.dqk, pk.qkt, nullptr, n_pad, 2 * d, d, w.dx_qk));
    if (fuse_ln && prev) {
      // ... and the LayerNorm-2 backward of the layer below: g = dx_res + dx_qk + bf16(dv Wv) is the gradient of its output
      const Saved sp = saved_layout(prev->saved, n_pad, d, ff, es);
      GD_TRY(gd_tok_gemm_ln_bwd(c.st, w.dv, pk.vt, n, n_pad, d, d, (const float*)w.dx_res, w.dx_qk, (const float*)sp.x1, sp.f,
                                (const float*)sp.st2, prev->g2, (float*)w.dx1_res, w.dfb, (float*)w.ln_ws2));
      return 0;
    }
    GD_TRY(gd_tok_gemm_plain(c.st, w.dv, pk.vt, nullptr, n_pad, d, d, w.dx_v));
  } else {
    GD_TRY(linear_dx(c, w.dqk, Win, w.dx_qk, n_pad, 2 * d, d));
    GD_TRY(linear_dx(c, w.dv, Win + (size_t)2 * d * d * es, w.dx_v, n_pad, d, d));
  }
  if (!defer_add3) GD_TRY(gdmae_add3((const float*)w.dx_res, w.dx_qk, a->bf16, w.dx_v, a->bf16, n * d, a->dx, stream));
  return 0;
}

// GDMAE_LAYER_TAIL_RIDES=0: the layer tails as launches of their own (A/B switch)
static bool tail_rides() {
  static int v = -1;
  if (v < 0) v = getenv("GDMAE_LAYER_TAIL_RIDES") ? atoi(getenv("GDMAE_LAYER_TAIL_RIDES")) : 1;
  return v != 0;
}

// Backward of stage_fwd_v2: per layer the feed-forward / LayerNorm-1 / out-projection launch, the attention backward, the grouped
// weight gradients + tail, and the in-projection launch that also runs the LayerNorm-2 backward of the layer below (df of that
// layer and its partial rows land in the shared scratch: dfb, ln_ws2).
static int stage_bwd_v2(const gdmae_layer_args* layers, int n_layers, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  for (int i = n_layers - 1; i >= 0; --i) {
    const gdmae_layer_args* a = &layers[i];
    const long long n = a->n, n_pad = pad_rows(n);
    const int d = a->d, ff = a->ff;
    Saved s = saved_layout(a->saved, n_pad, d, ff, 2);
    Scratch w = scratch_layout(a->scratch, n_pad, d, ff, 2, a->nhead);
    Ctx c{st, HIP_R_16BF, 2, w.lt_ws};
    const Packed pk = packed_layout(a->packed, d, ff);
    const gdmae_layer_args* top = &layers[n_layers - 1];
    const bool res_mode = top->dres != nullptr && layers[0].dx_bf16 != nullptr;      // block residual folded into the stage
    // the LayerNorm-2 partial rows of layer i live in ln2[i & 1]: this layer's tail reads them inside the launch that writes those of
    // the layer below
    float* const ln2[2] = {(float*)w.ln_ws2, (float*)w.ln_ws2b};
    if (i == n_layers - 1)
      GD_TRY(gd_layer_fused_ln2_top(st, d, res_mode ? nullptr : a->dy, res_mode ? a->dres : nullptr, s.x1b, s.f, (const float*)s.st2, a->g2, n,
                                    n_pad, w.dfb, ln2[i & 1]));
    GD_TRY(gd_layer_fused_bwd_ffn(st, d, w.dfb, s.h, s.xb, s.a, (const float*)s.st1, a->g1, pk.w2t, pk.w1t, pk.ot, n, n_pad, w.dh, s.gact,
                                  w.dab, w.d_o, (float*)w.ln_ws));
    long long pbase = 0;
    for (int l = 0; l < a->n_levels; ++l) pbase += (long long)a->n_win[l] * a->nhead;
    gd_attn_timing_tokens(n);
    GD_TRY(gdmae_window_attention_levels_bwd(s.qk, s.v, w.d_o, w.dqk, w.dv, 1, (float*)w.apart, a->csr_tok, a->win_start, a->win_len,
                                             a->n_levels, a->n_win, a->max_tokens, d, a->nhead, a->tau, a->tau_min, s.o, (const float*)s.lse, stream));
    const int nb = (int)(n_pad / gd_layer_fused_rows(d));
    TailJobs tail;
    const bool ride = tail_rides();
    GD_TRY(grouped_dw_and_tail(a, s, w, c, n, n_pad, nb, nb, pbase, ride ? &tail : nullptr, ln2[i & 1]));
    if (i > 0) {
      const Saved sp = saved_layout(layers[i - 1].saved, n_pad, d, ff, 2);
      GD_TRY(gd_layer_fused_bwd_in(st, d, w.dqk, w.dv, pk.qkt, pk.vt, w.dab, n, n_pad, sp.x1b, sp.f, (const float*)sp.st2, layers[i - 1].g2,
                                   w.dfb, ln2[(i - 1) & 1], nullptr, nullptr, nullptr, ride ? &tail : nullptr));
    } else {
      GD_TRY(gd_layer_fused_bwd_in(st, d, w.dqk, w.dv, pk.qkt, pk.vt, w.dab, n, n_pad, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                                   res_mode ? nullptr : a->dx, res_mode ? top->dres : nullptr, res_mode ? a->dx_bf16 : nullptr,
                                   ride ? &tail : nullptr));
    }
  }
  return 0;
}

extern "C" int gdmae_encoder_layer_bwd(const gdmae_layer_args* a, void* stream) {
  return layer_bwd(a, false, false, nullptr, false, stream);
}

// Backward of gdmae_encoder_stage_fwd (layers in reverse); all layers MUST share one scratch buffer: the three pieces
// of a layer's input gradient stay there and are summed on load by the previous layer's LayerNorm backward.
// layers[n_layers - 1].dy = upstream gradient of the stage, layers[0].dx = gradient of the stage input.
extern "C" int gdmae_encoder_stage_bwd(const gdmae_layer_args* layers, int n_layers, void* stream) {
  GD_REQUIRE(n_layers >= 1, "encoder stage: no layers");
  for (int i = 0; i < n_layers; ++i)
    GD_REQUIRE(layers[i].scratch == layers[0].scratch && layers[i].n == layers[0].n && layers[i].d == layers[0].d &&
                   layers[i].ff == layers[0].ff && layers[i].bf16 == layers[0].bf16,
               "encoder stage: layers must share scratch, n, d, ff, dtype");
  // The forward chose its path (gdmae_encoder_stage_fused) and laid the saved blocks out for it; on the fused path the caller chains the
  // layers with small tokens instead of row pointers (x[i] = 1 + i).  A gdmae_encoder_set_layer_path call between a forward and its
  // backward would send token "pointers" into the per-layer kernels, or fused kernels onto a launch-per-product saved block: refuse.
  const bool tokens = n_layers > 1 && (uintptr_t)layers[1].x < 4096;
  if (stage_v2(layers, n_layers)) {
    GD_REQUIRE(n_layers == 1 || tokens, "encoder stage backward: the forward of these layers ran the launch-per-product path (layer path changed in between)");
    return stage_bwd_v2(layers, n_layers, stream);
  }
  GD_REQUIRE(!tokens, "encoder stage backward: the forward of these layers ran the fused path (layer path changed in between)");
  GD_REQUIRE(!layers[n_layers - 1].dres && !layers[0].dx_bf16, "encoder stage: the folded block residual needs the fused path");
  // bf16 rows with packed weights: the LayerNorm-2 backward of layer i - 1 rides on layer i's last input-gradient GEMM
  bool chain = true;
  for (int i = 0; i < n_layers; ++i)
    chain = chain && use_fused(&layers[i]) && use_grouped_dw(&layers[i], pad_rows(layers[i].n)) && layers[i].packed != nullptr;
  for (int i = n_layers - 1; i >= 0; --i)
    GD_TRY(layer_bwd(&layers[i], i + 1 < n_layers, i > 0, (chain && i > 0) ? &layers[i - 1] : nullptr, chain && i + 1 < n_layers, stream));
  return 0;
}
