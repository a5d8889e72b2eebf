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
#ifdef TL_EXP_NOSTORE      // experiment: the streaming stores never execute (the values stay live)
#define TG_ST_U2(ptr, val) do { if (blockIdx.x == 0x7fffffffu) *(uint2*)(ptr) = (val); } while (0)
#define TG_ST_U4(ptr, val) do { if (blockIdx.x == 0x7fffffffu) *(uint4*)(ptr) = (val); } while (0)
#define TG_ST_F4(ptr, a, b, c, d) do { if (blockIdx.x == 0x7fffffffu) *(float4*)(ptr) = make_float4(a, b, c, d); } while (0)
#elif TG_NT_STORES
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
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union TgFrag {
  uint4 q;
  bf16x8 v;
  f16x8 h;      // the same 16 bytes as eight fp16 values (forward products of the decoder: common.h gd_pack_f16)
};

#define TG_ROWS 64     // row padding granule of the callers; the kernel's own tile is TG_R<ND> rows
#define TG_WAVES 8
#define TG_PF 4       // weight prefetch distance in k-steps

enum { TG_PLAIN = 0, TG_GELU = 1, TG_GELU_BWD = 2, TG_RES_LN = 3, TG_LN_BWD = 4 };

__device__ inline float tg_bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ inline unsigned short tg_f2bf(float f) { return gd_to_bf16(f); }
__device__ inline unsigned tg_pack2(float lo, float hi) { return gd_pack_bf16(lo, hi); }
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
__device__ inline uint4 tg_pack8_f16(const float (&f)[8]) {
  uint4 q;
  q.x = gd_pack_f16(f[0], f[1]); q.y = gd_pack_f16(f[2], f[3]); q.z = gd_pack_f16(f[4], f[5]); q.w = gd_pack_f16(f[6], f[7]);
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
#ifdef TL_EXP_NOGELU     // experiment: what the GELU arithmetic costs the fused launches (wrong values: timing only)
__device__ inline float tg_gelu(float h) { return 0.5f * h; }
__device__ inline float tg_gelu_grad(float h) { return 0.5f + 0.f * h; }
#else
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
#endif


// sum over the LPR (power of two, <= 64) consecutive lanes that share a row
template <int LPR>
__device__ inline float tg_group_sum(float v) {
  return gd_group_sum<LPR>(v);       // DPP / permlane swaps instead of six ds_bpermute round trips (common.h)
}

// ------------------------------------------------------------------------------------------------
// Row-tile building blocks shared by the fused layer launches (layer_fused.hip) and the row GEMMs of the decoder (rows_gemm.hip):
// TL_THREADS threads = TL_WAVES wavefronts (512 = 8 by default), a tile of ROWS bf16 rows in LDS, weights streamed as MFMA A operands from a packed image.
// ------------------------------------------------------------------------------------------------
#ifndef TL_PF
#define TL_PF TG_PF      // weight prefetch distance in k-steps (experiment switch, tools/build_variant.sh)
#endif
#ifndef TL_WAVES
#define TL_WAVES TG_WAVES   // wavefronts of a row tile's workgroup (experiment switch: 4 = 256 threads, three workgroups per CU)
#endif
#define TL_THREADS (64 * TL_WAVES)
// ---- one product of a row tile:  acc[j][b] (32 channels x 32 rows, fp32) += W (ND, KD) X^T, X = bf16 rows in LDS ---------------
template <int KD, int ND, int ROWS>
struct TlShape {
  static constexpr int KS = KD / 16;                                    // k-steps
  static constexpr int MB = ND / 32;                                    // 32-channel blocks of the output
  static constexpr int MPW = MB >= TL_WAVES ? MB / TL_WAVES : 1;        // channel blocks per wavefront
  static constexpr int NPW = MB >= TL_WAVES ? ROWS / 32 : 1;            // 32-row blocks per wavefront
  static_assert(MB >= TL_WAVES || (MB * (ROWS / 32) == TL_WAVES), "every wavefront needs an output block");
  __device__ static int mb0(int wv) { return MB >= TL_WAVES ? wv : (wv / (ROWS / 32)); }
  __device__ static int nb0(int wv) { return MB >= TL_WAVES ? 0 : (wv % (ROWS / 32)); }
};

// fragment (ks, j) of a wavefront; the image of a second matrix continues the K dimension after KSPLIT k-steps (the q/k and v
// halves of the packed in-projection are two images)
// mbs: 32-channel blocks per k-step of the image the pointers point into (ND / 32 for a whole image, more when the workgroup owns a
// column slice of a wider one)
template <int KD, int ND, int ROWS, int KSPLIT>
__device__ __forceinline__ uint4 tl_wfrag(const uint4* __restrict__ w0, const uint4* __restrict__ w1, int ks, int j, int mbs) {
  return ks < KSPLIT ? w0[((size_t)ks * mbs + j * TL_WAVES) * 64] : w1[((size_t)(ks - KSPLIT) * mbs + j * TL_WAVES) * 64];
}

// prefetch distance of a product: TL_PF k-steps, TL_PF1 for products whose wavefronts own ONE channel block (a ring of TL_PF1 + 1
// fragments = 4 VGPRs each; with two blocks per wavefront the deeper ring would not fit 128 registers)
#ifndef TL_PF1
#define TL_PF1 TL_PF
#endif
// F16: both operands hold fp16 values (the packed image and the LDS tile), the product is v_mfma_f32_32x32x16_f16
template <int KD, int ND, int ROWS, int KSPLIT = KD / 16, bool F16 = false>
struct TlProd {
  static constexpr int PFD = (TlShape<KD, ND, ROWS>::MPW == 1 && KD / 16 > TL_PF1) ? TL_PF1 : TL_PF;
  using S = TlShape<KD, ND, ROWS>;
  TgFrag wr[PFD + 1][S::MPW];
  const uint4* __restrict__ w0;
  const uint4* __restrict__ w1;
  int mbs;
  // first PFD k-steps of the weights: issued early (before a row pass or a tile load) so that their latency is hidden
  __device__ __forceinline__ void prefetch(const uint4* W0, const uint4* W1, int wv, int lane, int image_blocks = S::MB) {
    mbs = image_blocks;
    w0 = W0 + (size_t)S::mb0(wv) * 64 + lane;
    w1 = W1 ? W1 + (size_t)S::mb0(wv) * 64 + lane : w0;
#pragma unroll
    for (int ks = 0; ks < PFD; ++ks)
#pragma unroll
      for (int j = 0; j < S::MPW; ++j) wr[ks][j].q = tl_wfrag<KD, ND, ROWS, KSPLIT>(w0, w1, ks, j, mbs);
  }
  __device__ __forceinline__ void run(const unsigned char* xs, int XP, int wv, int lane, f32x16 (&acc)[S::MPW][S::NPW]) {
#ifdef TL_EXP_NOMFMA
    return;
#endif
    const unsigned char* lb = xs + ((S::nb0(wv) * 32) + (lane & 31)) * XP + (lane >> 5) * 16;
    TgFrag sf[2][S::NPW];
#pragma unroll
    for (int b = 0; b < S::NPW; ++b) sf[0][b].q = *(const uint4*)(lb + b * 32 * XP);
#pragma unroll
    for (int ks = 0; ks < S::KS; ++ks) {
      if (ks + PFD < S::KS) {
#pragma unroll
        for (int j = 0; j < S::MPW; ++j) wr[(ks + PFD) % (PFD + 1)][j].q = tl_wfrag<KD, ND, ROWS, KSPLIT>(w0, w1, ks + PFD, j, mbs);
      }
      if (ks + 1 < S::KS) {
#pragma unroll
        for (int b = 0; b < S::NPW; ++b) sf[(ks + 1) & 1][b].q = *(const uint4*)(lb + b * 32 * XP + (ks + 1) * 32);
      }
#pragma unroll
      for (int j = 0; j < S::MPW; ++j)
#pragma unroll
        for (int b = 0; b < S::NPW; ++b)
          if constexpr (F16) acc[j][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wr[ks % (PFD + 1)][j].h, sf[ks & 1][b].h, acc[j][b], 0, 0, 0);
          else acc[j][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[ks % (PFD + 1)][j].v, sf[ks & 1][b].v, acc[j][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

template <int MPW, int NPW>
__device__ __forceinline__ void tl_zero(f32x16 (&acc)[MPW][NPW]) {
#pragma unroll
  for (int j = 0; j < MPW; ++j)
#pragma unroll
    for (int b = 0; b < NPW; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][b][i] = 0.f;
}

// accumulators (+ bias) -> bf16 -> LDS tile [row][channel] with row pitch SP.  The bias fragments of the wavefront are requested
// together and unconditionally (HAS_BIAS is a compile-time flag): a runtime `if (bias)` around each load made the compiler drain the
// load counter after every one of them (a join with a load in flight) - four serial L2 round trips per product.
// bias fragments of a wavefront's output blocks: load() before the product runs (they land behind its MFMAs), used by tl_stage_pre
template <int KD, int ND, int ROWS>
struct TlBias {
  using S = TlShape<KD, ND, ROWS>;
  uint2 bq[S::MPW][4];
  __device__ __forceinline__ void load(const unsigned short* bias, int wv, int lane) {
#pragma unroll
    for (int j = 0; j < S::MPW; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[j][q] = *(const uint2*)(bias + (S::mb0(wv) + j * TL_WAVES) * 32 + 4 * (lane >> 5) + 8 * q);
  }
};
template <int KD, int ND, int ROWS, bool HAS_BIAS>
__device__ __forceinline__ void tl_stage_pre(const f32x16 (&acc)[TlShape<KD, ND, ROWS>::MPW][TlShape<KD, ND, ROWS>::NPW],
                                             const uint2 (&bq)[TlShape<KD, ND, ROWS>::MPW][4], unsigned char* out, int SP, int wv, int lane) {
  using S = TlShape<KD, ND, ROWS>;
#pragma unroll
  for (int j = 0; j < S::MPW; ++j) {
    const int cb = (S::mb0(wv) + j * TL_WAVES) * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int b = 0; b < S::NPW; ++b) {
      const int row = (S::nb0(wv) + b) * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[j][b][4 * q + e];
        if constexpr (HAS_BIAS) {
          v[0] += __uint_as_float(bq[j][q].x << 16); v[1] += __uint_as_float(bq[j][q].x & 0xFFFF0000u);
          v[2] += __uint_as_float(bq[j][q].y << 16); v[3] += __uint_as_float(bq[j][q].y & 0xFFFF0000u);
        }
        uint2 o;
        o.x = tg_pack2(v[0], v[1]);
        o.y = tg_pack2(v[2], v[3]);
        *(uint2*)(out + row * SP + (cb + 8 * q) * 2) = o;
      }
    }
  }
}
template <int KD, int ND, int ROWS>
__device__ __forceinline__ void tl_stage(const f32x16 (&acc)[TlShape<KD, ND, ROWS>::MPW][TlShape<KD, ND, ROWS>::NPW], decltype(nullptr),
                                         unsigned char* out, int SP, int wv, int lane) {
  uint2 none[TlShape<KD, ND, ROWS>::MPW][4];
  tl_stage_pre<KD, ND, ROWS, false>(acc, none, out, SP, wv, lane);
}
template <int KD, int ND, int ROWS>
__device__ __forceinline__ void tl_stage(const f32x16 (&acc)[TlShape<KD, ND, ROWS>::MPW][TlShape<KD, ND, ROWS>::NPW], const unsigned short* bias,
                                         unsigned char* out, int SP, int wv, int lane) {
  TlBias<KD, ND, ROWS> b;
  b.load(bias, wv, lane);                                            // bias must not be null
  tl_stage_pre<KD, ND, ROWS, true>(acc, b.bq, out, SP, wv, lane);
}
template <int KD, int ND, int ROWS>
__device__ __forceinline__ void tl_stage(const f32x16 (&acc)[TlShape<KD, ND, ROWS>::MPW][TlShape<KD, ND, ROWS>::NPW], const TlBias<KD, ND, ROWS>& b,
                                         unsigned char* out, int SP, int wv, int lane) {
  tl_stage_pre<KD, ND, ROWS, true>(acc, b.bq, out, SP, wv, lane);
}

__device__ __forceinline__ void tl_unpack4(const uint2& q, float (&f)[4]) {
  f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xFFFF0000u);
  f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xFFFF0000u);
}
__device__ __forceinline__ uint2 tl_pack4(const float (&f)[4]) {
  uint2 q;
  q.x = tg_pack2(f[0], f[1]); q.y = tg_pack2(f[2], f[3]);
  return q;
}

// rows of a bf16 (n_pad, W) matrix -> LDS tile, 16 bytes per thread and access; c_off: first column of the tile (bytes)
template <int W, int ROWS>
__device__ __forceinline__ void tl_load_tile(const unsigned short* __restrict__ src, long long row0, unsigned char* dst, int P, int c_off, int tid) {
  constexpr int CPR = W / 8, RPP = TL_THREADS / CPR;
  static_assert(ROWS % RPP == 0 || RPP > ROWS, "tile load");
  const int c = tid % CPR, r = tid / CPR;
#pragma unroll
  for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
    const int row = p * RPP + r;
    if (RPP > ROWS && row >= ROWS) break;
    *(uint4*)(dst + row * P + c_off + c * 16) = *(const uint4*)(src + (row0 + row) * W + c * 8);
  }
}

