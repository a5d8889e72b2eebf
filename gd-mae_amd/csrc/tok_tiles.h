// Shared pieces of the token-GEMM kernels (tok_gemm.hip, layer_fused.hip): bf16 pack / unpack, the MFMA fragment union, the
// cache-policy stores / loads and the erf-GELU of the bf16 epilogues.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// streaming (nontemporal) stores for outputs whose next reader is far away - the fp32 rows of the residual stream, the branch
// output and the pre-activation kept for the backward: 2 - 6 % on the LayerNorm-epilogue launches (35.6 vs 37.8 us for 128 -> 256,
// 39.0 vs 40.8 for 256 -> 256 at 43 k rows), -0.085 ms per step in a same-box A/B (TG_NT_STORES=0 builds the plain stores)
#ifndef TG_NT_STORES
#define TG_NT_STORES 1
#endif
#if TG_NT_STORES
#define TG_ST_U2(ptr, val) __builtin_nontemporal_store(*(const unsigned long long*)&(val), (unsigned long long*)(ptr))
#define TG_ST_U4(ptr, val)                                                              \
  do {                                                                                  \
    typedef unsigned int tg_u4v __attribute__((ext_vector_type(4)));                    \
    const tg_u4v v_ = {(val).x, (val).y, (val).z, (val).w};                             \
    __builtin_nontemporal_store(v_, (tg_u4v*)(ptr));                                    \
  } while (0)
#define TG_ST_F4(ptr, a, b, c, d)                                                       \
  do {                                                                                  \
    typedef float tg_f4v __attribute__((ext_vector_type(4)));                           \
    const tg_f4v v_ = {a, b, c, d};                                                     \
    __builtin_nontemporal_store(v_, (tg_f4v*)(ptr));                                    \
  } while (0)
#else
#define TG_ST_U2(ptr, val) (*(uint2*)(ptr) = (val))
#define TG_ST_U4(ptr, val) (*(uint4*)(ptr) = (val))
#define TG_ST_F4(ptr, a, b, c, d) (*(float4*)(ptr) = make_float4(a, b, c, d))
#endif
// streaming (nontemporal) loads for operands that were written a whole forward pass ago and are read exactly once (the LayerNorm
// addends and the GELU pre-activation saved for the backward): they do not displace the gradients the neighbouring launches pass
// to each other through L2 / MALL
#ifndef TG_NT_LOADS
#define TG_NT_LOADS 1
#endif
__device__ inline float4 tg_ld_f4_once(const float* p) {
#if TG_NT_LOADS
  typedef float v4 __attribute__((ext_vector_type(4)));
  const v4 v = __builtin_nontemporal_load((const v4*)p);
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *(const float4*)p;
#endif
}
__device__ inline uint2 tg_ld_u2_once(const void* p) {
#if TG_NT_LOADS
  const unsigned long long v = __builtin_nontemporal_load((const unsigned long long*)p);
  return make_uint2((unsigned)v, (unsigned)(v >> 32));
#else
  return *(const uint2*)p;
#endif
}
__device__ inline uint4 tg_ld_u4_once(const void* p) {
#if TG_NT_LOADS
  typedef unsigned int v4 __attribute__((ext_vector_type(4)));
  const v4 v = __builtin_nontemporal_load((const v4*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
#else
  return *(const uint4*)p;
#endif
}
union TgFrag {
  uint4 q;
  bf16x8 v;
};

#define TG_ROWS 64     // row padding granule of the callers; the kernel's own tile is TG_R<ND> rows
#define TG_WAVES 8
#define TG_PF 4       // weight prefetch distance in k-steps

enum { TG_PLAIN = 0, TG_GELU = 1, TG_GELU_BWD = 2, TG_RES_LN = 3, TG_LN_BWD = 4 };

__device__ inline float tg_bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ inline unsigned short tg_f2bf(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7F800000u) == 0x7F800000u) return (unsigned short)(u >> 16);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ inline unsigned tg_pack2(float lo, float hi) { return tg_f2bf(lo) | ((unsigned)tg_f2bf(hi) << 16); }
__device__ inline void tg_unpack8(const uint4& u, float (&f)[8]) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(w[k] << 16);
    f[2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u);
  }
}
__device__ inline uint4 tg_pack8(const float (&f)[8]) {
  uint4 q;
  q.x = tg_pack2(f[0], f[1]); q.y = tg_pack2(f[2], f[3]); q.z = tg_pack2(f[4], f[5]); q.w = tg_pack2(f[6], f[7]);
  return q;
}

// erf-GELU for rows that are rounded to bf16 right after: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. 2^-13
// of a bf16 ulp at unit scale) on v_rcp_f32 / v_exp_f32 - about a third of the VALU work of erff(), which at one output
// row per lane and no MFMA left to hide behind was what the fused epilogues were bound by (exact erff: +12 us per 10 M
// hidden elements).  Phi(h) = (1 + erf(h / sqrt 2)) / 2 and phi(h) share the exponential exp(-h^2 / 2).
__device__ inline void tg_phi(float h, float& cdf, float& pdf_e) {
  const float x = fabsf(h) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.f));
  const float e = __expf(-x * x);                                   // = exp(-h^2 / 2)
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float q = 0.5f * p * t * e;                                 // (1 - erf(x)) / 2
  cdf = h >= 0.f ? 1.f - q : q;
  pdf_e = e;
}
__device__ inline float tg_gelu(float h) {
  float cdf, e;
  tg_phi(h, cdf, e);
  return h * cdf;
}
__device__ inline float tg_gelu_grad(float h) {
  float cdf, e;
  tg_phi(h, cdf, e);
  return fmaf(h * 0.39894228040143267794f, e, cdf);
}


// sum over the LPR (power of two, <= 64) consecutive lanes that share a row
template <int LPR>
__device__ inline float tg_group_sum(float v) {
#pragma unroll
  for (int d = LPR / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
