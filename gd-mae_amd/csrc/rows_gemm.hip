// The decoder's ConvTranspose2d(k = s, stride s, no bias) blocks on token rows (reference
// pcdet/models/backbones_3d/spt_backbone_mae.py:30-45 `decoder_deblocks`, applied to the densified stage maps at :125-131): with kernel
// size = stride the s x s output sites of an input site do not overlap, so on the ACTIVE tokens the block is a row product
//
//     P (n, s*s*cout) = X (n, cin) Wm,   Wm[ci][(q, c)] = w[ci][c][q],  q = dy * s + dx,  w = deconv.weight (cin, cout, s, s)
//
// whose row (token t, q) is the deconvolution output at the full-resolution site of (t, dy, dx).  Forward, input gradient and weight
// gradient ran through hipBLASLt until round 4 (12 library launches + weight permutes / casts per step); here:
//   k_deconv_pack        both MFMA-fragment-ordered images of w (forward (s*s*cout, cin), input gradient (cin, s*s*cout)) in one launch
//   k_rows_gemm          P = X Wm: a workgroup = a row tile x a 128 / 512-column slice, rows guarded (token counts are not padded)
//   k_rows_gemm_kc       dX = dP Wm^T: K = s*s*cout up to 2048 in chunks of <= 512 through one LDS tile, accumulators stay in registers
//   weight gradient      Wm-shaped dWm = X^T dP through the grouped TN kernel (dw_grouped.hip, guarded row loads) and
//   k_deconv_dw_reduce   its fixed-order split-K reduce, accumulated straight into weight.grad's (cin, cout, s, s) layout
// Same MFMA mapping as the token GEMMs (tok_tiles.h), fp32 accumulation, bf16 results; the FORWARD product takes fp16 operands (round 6:
// the bf16 rounding of these weights was the largest single term of the bf16 mode's loss deviation at full size), the gradients bf16.
#include "../../include/gdmae_hip.h"
#include "dw_grouped.h"
#include "tok_tiles.h"

namespace {

// ---- weight images ---------------------------------------------------------------------------------
// image of an (M, K) matrix A: dst[(ks * (M / 32) + mb) * 64 + lane] = 8 bf16 A[mb * 32 + (lane & 31)][ks * 16 + (lane >> 5) * 8 + j]
__global__ __launch_bounds__(256) void k_deconv_pack(const float* __restrict__ w, int cin, int cout, int ss, uint4* __restrict__ fwd,
                                                     uint4* __restrict__ bwd) {
  const int N = ss * cout;
  const int total = (N / 32) * (cin / 16) * 64;          // both images hold N * cin / 8 fragments-of-8
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * total; i += gridDim.x * blockDim.x) {
    const bool is_bwd = i >= total;
    const int e = is_bwd ? i - total : i;
    const int lane = e & 63;
    float f[8];
    if (!is_bwd) {                                        // A = Wm^T: rows r = (q, c), columns ci
      const int MB = N / 32, mb = (e >> 6) % MB, ks = (e >> 6) / MB;
      const int r = mb * 32 + (lane & 31), q = r / cout, c = r - q * cout;
      const int k0 = ks * 16 + (lane >> 5) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = w[((long long)(k0 + j) * cout + c) * ss + q];
    } else {                                              // A = Wm: rows ci, columns k = (q, c)
      const int MB = cin / 32, mb = (e >> 6) % MB, ks = (e >> 6) / MB;
      const int ci = mb * 32 + (lane & 31);
      const int k0 = ks * 16 + (lane >> 5) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j, q = k / cout, c = k - q * cout;
        f[j] = w[((long long)ci * cout + c) * ss + q];
      }
    }
    // the FORWARD image holds fp16 values (k_rows_gemm multiplies fp16 operands: the rounding of these weights is the same error at
    // every site of the map and dominated the bf16 mode's loss deviation at full size, common.h gd_pack_f16); the gradient image bf16
    if (is_bwd) bwd[e] = tg_pack8(f);
    else fwd[e] = tg_pack8_f16(f);
  }
}

// ---- guarded tile load: rows >= n repeat row n - 1 (finite garbage; their results are never stored) -----------------
// TOF16: the bf16 rows are converted to fp16 on their way into LDS (exact: encoder outputs are O(1) LayerNorm / BatchNorm rows)
template <int W, int ROWS, bool TOF16 = false>
__device__ __forceinline__ void rg_load_tile(const unsigned short* __restrict__ src, long long ld, int c0, long long row0, long long n, unsigned char* dst,
                                             int P, int tid) {
  constexpr int CPR = W / 8, RPP = 512 / CPR;
  const int c = tid % CPR, r = tid / CPR;
#pragma unroll
  for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
    const int row = p * RPP + r;
    if (RPP > ROWS && row >= ROWS) break;
    long long g = row0 + row;
    g = g < n ? g : n - 1;
    const uint4 q = *(const uint4*)(src + g * ld + c0 + c * 8);
    if constexpr (TOF16) *(uint4*)(dst + row * P + c * 16) = gd_bf16x8_to_f16(q);
    else *(uint4*)(dst + row * P + c * 16) = q;
  }
}
template <int W, int ROWS>
__device__ __forceinline__ void rg_store_tile(const unsigned char* src, int P, unsigned short* __restrict__ dst, long long ld, int c0, long long row0,
                                              long long n, int tid) {
  constexpr int CPR = W / 8, RPP = 512 / CPR;
  const int c = tid % CPR, r = tid / CPR;
#pragma unroll
  for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
    const int row = p * RPP + r;
    if (RPP > ROWS && row >= ROWS) break;
    if (row0 + row < n) *(uint4*)(dst + (row0 + row) * ld + c0 + c * 8) = *(const uint4*)(src + row * P + c * 16);
  }
}

template <int NSL>
struct RgRows {
#ifndef RG_ROWS_WIDE
#define RG_ROWS_WIDE 64       // rows per workgroup for column slices >= 256 (32 before: deconvolution rows 217 -> 198 us per step)
#endif
  static constexpr int value = NSL >= 256 ? RG_ROWS_WIDE : 64;
};

struct RgArgs {
  const unsigned short* X;      // (n, K) bf16
  const uint4* Wp;              // packed (N, K), fp16 values
  unsigned short* Y;            // (n, N) bf16
  long long n;
  int N;
};

// Y[:, slice] = X Wp[slice]^T; blockIdx.y = column slice of NSL channels
template <int KD, int NSL>
__global__ __launch_bounds__(512, 4) void k_rows_gemm(RgArgs A) {
  constexpr int ROWS = RgRows<NSL>::value;
  constexpr int XP = KD * 2 + 16, SP = NSL * 2 + 16;
  using S = TlShape<KD, NSL, ROWS>;
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long row0 = (long long)blockIdx.x * ROWS;
  const int sl = blockIdx.y;
  TlProd<KD, NSL, ROWS, KD / 16, true> pr;            // fp16 operands: image of k_deconv_pack, rows converted on load
  pr.prefetch(A.Wp + (size_t)sl * (NSL / 32) * 64, nullptr, wv, lane, A.N / 32);
  rg_load_tile<KD, ROWS, true>(A.X, KD, 0, row0, A.n, lds, XP, tid);
  __syncthreads();
  f32x16 acc[S::MPW][S::NPW];
  tl_zero(acc);
  pr.run(lds, XP, wv, lane, acc);
  __syncthreads();
  tl_stage<KD, NSL, ROWS>(acc, nullptr, lds, SP, wv, lane);
  __syncthreads();
  rg_store_tile<NSL, ROWS>(lds, SP, A.Y, A.N, sl * NSL, row0, A.n, tid);
}

struct RkArgs {
  const unsigned short* G;      // (n, KT) bf16
  const uint4* Wp;              // packed (ND, KT)
  unsigned short* Y;            // (n, ND) bf16
  long long n;
  int KT;
};

// Y = G Wp^T with K = KT walked in chunks of KC through one LDS tile
template <int KC, int ND>
__global__ __launch_bounds__(512, 4) void k_rows_gemm_kc(RkArgs A) {
  constexpr int ROWS = RgRows<ND>::value;
  constexpr int XP = KC * 2 + 16, SP = ND * 2 + 16;
  using S = TlShape<KC, ND, ROWS>;
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long row0 = (long long)blockIdx.x * ROWS;
  f32x16 acc[S::MPW][S::NPW];
  tl_zero(acc);
  const int chunks = A.KT / KC;
  for (int ch = 0; ch < chunks; ++ch) {
    TlProd<KC, ND, ROWS> pr;
    pr.prefetch(A.Wp + (size_t)ch * (KC / 16) * (ND / 32) * 64, nullptr, wv, lane);
    if (ch) __syncthreads();                           // every wavefront is done with the previous chunk's tile
    rg_load_tile<KC, ROWS>(A.G, A.KT, ch * KC, row0, A.n, lds, XP, tid);
    __syncthreads();
    pr.run(lds, XP, wv, lane, acc);
  }
  __syncthreads();
  tl_stage<KC, ND, ROWS>(acc, nullptr, lds, SP, wv, lane);
  __syncthreads();
  rg_store_tile<ND, ROWS>(lds, SP, A.Y, ND, 0, row0, A.n, tid);
}

// dW[(ci * cout + c) * ss + q] += sum_s part[s][ci][q * cout + c]   (part: (S, cin, ss * cout) fp32, fixed order)
__global__ __launch_bounds__(256) void k_deconv_dw_reduce(const float* __restrict__ part, int S, int cin, int cout, int ss, float* __restrict__ dW) {
  const long long P = (long long)cin * ss * cout;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < P; e += (long long)gridDim.x * blockDim.x) {
    float a = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {                   // eight slices in flight, added in slice order
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[(long long)(s + j) * P + e];
#pragma unroll
      for (int j = 0; j < 8; ++j) a += v[j];
    }
    for (; s < S; ++s) a += part[(long long)s * P + e];
    const int n = (int)(e % (ss * cout)), ci = (int)(e / (ss * cout));
    const int q = n / cout, c = n - q * cout;
    dW[((long long)ci * cout + c) * ss + q] += a;
  }
}

// ------------------------------------------------------------------------------------------------
// Prediction head: nn.Linear(128 -> n_out <= 64) on the fp32 decoder rows of ALL pillars (reference spt_backbone_mae.py:52,74
// `decoder_pred`, 16 points x 3 = 48 outputs).  Under autocast the framework ran a cast pass, a library GEMM with K / N = 48
// (unaligned: 62 us for the input gradient alone) and another cast; here the fp32 rows are rounded on their way into LDS, the
// weight image is zero-padded to 64 outputs, and the weight / bias gradients are one register-tiled VALU pass over row blocks.
// ------------------------------------------------------------------------------------------------
constexpr int kPredK = 128, kPredN = 64, kPredRows = 128;     // TlShape<128, 64, 128>: 2 channel blocks x 4 row blocks

// images of W (n_out, 128) fp32: forward (64, 128) rows >= n_out zero - TWO images, the bf16 rounding of W and the bf16 rounding of the
// remainder W - bf16(W) (k_pred_fwd's three-term product); input gradient (128, 64) columns >= n_out zero; bias (64) bf16 + fp32
__global__ __launch_bounds__(256) void k_pred_pack(const float* __restrict__ w, const float* __restrict__ b, int n_out, uint4* __restrict__ fwd,
                                                   uint4* __restrict__ fwd_lo, uint4* __restrict__ bwd, unsigned short* __restrict__ bias,
                                                   float* __restrict__ bias_f) {
  const int total = kPredK * kPredN / 8;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * total; i += gridDim.x * blockDim.x) {
    const bool is_bwd = i >= total;
    const int e = is_bwd ? i - total : i, lane = e & 63;
    float f[8];
    if (!is_bwd) {                                      // A (64, 128): row o, column ci
      const int MB = kPredN / 32, mb = (e >> 6) % MB, ks = (e >> 6) / MB;
      const int o = mb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f[j] = o < n_out ? w[o * kPredK + k0 + j] : 0.f;
        r[j] = f[j] - tg_bf2f(tg_f2bf(f[j]));           // exact in fp32
      }
      fwd_lo[e] = tg_pack8(r);
    } else {                                            // A (128, 64): row ci, column o
      const int MB = kPredK / 32, mb = (e >> 6) % MB, ks = (e >> 6) / MB;
      const int ci = mb * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (k0 + j) < n_out ? w[(k0 + j) * kPredK + ci] : 0.f;
    }
    (is_bwd ? bwd : fwd)[e] = tg_pack8(f);
  }
  if (blockIdx.x == 0 && threadIdx.x < kPredN) {
    const float v = b && (int)threadIdx.x < n_out ? b[threadIdx.x] : 0.f;
    bias[threadIdx.x] = tg_f2bf(v);
    bias_f[threadIdx.x] = v;
  }
}

struct PhArgs {
  const float* X;               // (n, 128) fp32
  const uint4* Wp;              // packed (64, 128): bf16(W)
  const uint4* Wlo;             // packed (64, 128): bf16(W - bf16(W))
  const float* bias;            // (64) fp32
  unsigned short* Y;            // optional (n, n_out) bf16: the result rounded
  long long n;
  int n_out;
  unsigned short* Xb;           // optional (n, 128) bf16: the rounded operand rows, kept for the weight gradient
  float* Yf;                    // optional (n, n_out) fp32: the result as accumulated (what the Chamfer kernel reads)
};
// y = x W^T + b to fp32 accuracy on the bf16 matrix cores: x = xh + xl, W = Wh + Wl (each the bf16 rounding of the value and of its
// remainder), y = xh Wh + xl Wh + xh Wl accumulated in fp32 (the dropped xl Wl term is 2^-18 relative).  The Chamfer loss reads these rows
// as point offsets and squares their distance to the ground truth: with the result (or the operands) rounded to bf16 the squared
// rounding error is a BIAS of the loss of 1 - 4e-4 relative (measured on the reference goldens, tools/bench_vs_golden.py) - the one
// place of the bf16 step where rounding does not average out.
__global__ __launch_bounds__(512, 2) void k_pred_fwd(PhArgs A) {
  constexpr int XP = kPredK * 2 + 16, SP = kPredN * 4 + 16;
  constexpr int KS = kPredK / 16;
  using S = TlShape<kPredK, kPredN, kPredRows>;
  static_assert(S::MPW == 1 && S::NPW == 1, "one 32 x 32 output block per wavefront");
  extern __shared__ __align__(16) unsigned char lds[];
  unsigned char* lds_lo = lds + kPredRows * XP;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long row0 = (long long)blockIdx.x * kPredRows;
  TgFrag wh[KS], wl[KS];                                // this wavefront's channel block of both images, whole K: 64 registers
  {
    const uint4* ph = A.Wp + (size_t)S::mb0(wv) * 64 + lane;
    const uint4* pl = A.Wlo + (size_t)S::mb0(wv) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      wh[ks].q = ph[(size_t)ks * S::MB * 64];
      wl[ks].q = pl[(size_t)ks * S::MB * 64];
    }
  }
  {   // fp32 rows -> two bf16 tiles (value, remainder): 16 chunks of 8 channels per row, 32 rows per pass
    const int c = tid & 15, r = tid >> 4;
#pragma unroll
    for (int p = 0; p < kPredRows / 32; ++p) {
      const int row = p * 32 + r;
      long long g = row0 + row;
      g = g < A.n ? g : A.n - 1;
      const float4 a = *(const float4*)(A.X + g * kPredK + c * 8), b = *(const float4*)(A.X + g * kPredK + c * 8 + 4);
      const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
      const uint4 q = tg_pack8(f);
      float h[8], rem[8];
      tg_unpack8(q, h);
#pragma unroll
      for (int e = 0; e < 8; ++e) rem[e] = f[e] - h[e];
      *(uint4*)(lds + row * XP + c * 16) = q;
      *(uint4*)(lds_lo + row * XP + c * 16) = tg_pack8(rem);
      if (A.Xb && row0 + row < A.n) *(uint4*)(A.Xb + (row0 + row) * kPredK + c * 8) = q;
    }
  }
  __syncthreads();
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  {
    const int off = ((S::nb0(wv) * 32) + (lane & 31)) * XP + (lane >> 5) * 16;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      TgFrag xh, xl;
      xh.q = *(const uint4*)(lds + off + ks * 32);
      xl.q = *(const uint4*)(lds_lo + off + ks * 32);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[ks].v, xh.v, acc, 0, 0, 0);      // small terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks].v, xl.v, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[ks].v, xh.v, acc, 0, 0, 0);
    }
  }
  __syncthreads();
  {   // accumulators + bias -> fp32 staging tile [row][channel]
    const int cb = S::mb0(wv) * 32 + 4 * (lane >> 5), row = S::nb0(wv) * 32 + (lane & 31);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bv = *(const float4*)(A.bias + cb + 8 * q);
      *(float4*)(lds + row * SP + (cb + 8 * q) * 4) = make_float4(acc[4 * q] + bv.x, acc[4 * q + 1] + bv.y, acc[4 * q + 2] + bv.z, acc[4 * q + 3] + bv.w);
    }
  }
  __syncthreads();
  const int cpr = A.n_out >> 2;                         // 16-byte chunks per output row (n_out % 4 == 0)
  for (int i = tid; i < kPredRows * cpr; i += 512) {
    const int row = i / cpr, c = i - row * cpr;
    if (row0 + row < A.n) {
      const float4 v = *(const float4*)(lds + row * SP + c * 16);
      if (A.Yf) *(float4*)(A.Yf + (row0 + row) * A.n_out + c * 4) = v;
      if (A.Y) *(uint2*)(A.Y + (row0 + row) * A.n_out + c * 4) = make_uint2(tg_pack2(v.x, v.y), tg_pack2(v.z, v.w));
    }
  }
}

struct PbArgs {
  const void* dY;               // (n, n_out) bf16, or fp32 (dy_f32): then rounded here and written to dYb for the weight gradient
  const uint4* Wp;              // packed (128, 64)
  float* dX;                    // (n, 128) fp32 or null
  long long n;
  int n_out;
  int dy_f32;
  unsigned short* dYb;          // (n, n_out) bf16 out (dy_f32 only)
  const float *scale_a, *scale_b;   // optional device scalars: the fp32 rows are multiplied by scale_a[0] * scale_b[0] first (the loss's
                                    // upstream gradient x 1 / sum of weights: no scaling pass over the Chamfer gradient)
};
// dX (n, 128) fp32 = dY (n, n_out) W: K padded to 64 with zero columns
__global__ __launch_bounds__(512, 2) void k_pred_bwd_input(PbArgs A) {
  constexpr int ROWS = 64, KD = kPredN, ND = kPredK, XP = KD * 2 + 16;
  using S = TlShape<KD, ND, ROWS>;
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long row0 = (long long)blockIdx.x * ROWS;
  TlProd<KD, ND, ROWS> pr;
  pr.prefetch(A.Wp, nullptr, wv, lane);
  const float scale = (A.dy_f32 && A.scale_a) ? A.scale_a[0] * (A.scale_b ? A.scale_b[0] : 1.f) : 1.f;
  {
    const int cpr = A.n_out >> 2;                       // 8-byte chunks per row that hold data; the tile row has 16
    for (int i = tid; i < ROWS * 16; i += 512) {
      const int row = i >> 4, c = i & 15;
      long long g = row0 + row;
      g = g < A.n ? g : A.n - 1;
      uint2 q = make_uint2(0u, 0u);
      if (c < cpr) {
        if (A.dy_f32) {
          float4 v = *(const float4*)((const float*)A.dY + g * A.n_out + c * 4);
          v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
          q.x = tg_pack2(v.x, v.y); q.y = tg_pack2(v.z, v.w);
          if (row0 + row < A.n) *(uint2*)(A.dYb + g * A.n_out + c * 4) = q;
        } else {
          q = *(const uint2*)((const unsigned short*)A.dY + g * A.n_out + c * 4);
        }
      }
      *(uint2*)(lds + row * XP + c * 8) = q;
    }
  }
  __syncthreads();
  f32x16 acc[S::MPW][S::NPW];
  tl_zero(acc);
  pr.run(lds, XP, wv, lane, acc);
  // fp32 rows straight from the accumulators: lane = row, registers = channels cb + 8 q + e
  const int cb = S::mb0(wv) * 32 + 4 * (lane >> 5);
  const long long row = row0 + S::nb0(wv) * 32 + (lane & 31);
  if (row < A.n && A.dX) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
      *(float4*)(A.dX + row * ND + cb + 8 * q) = make_float4(acc[0][0][4 * q], acc[0][0][4 * q + 1], acc[0][0][4 * q + 2], acc[0][0][4 * q + 3]);
  }
}

// dW[o][ci] += sum_s part[s][o][ci] (o < n_out; part (S, 128, 128) from the grouped TN kernel with G = dY zero-padded to 128
// columns), db[o] += sum_s colpart[s][o]: fixed order
// sum of S slice values at stride `st`, eight loads in flight, added in slice order (bit-identical to a one-load-per-iteration loop)
__device__ __forceinline__ float rg_slice_sum(const float* __restrict__ p, int S, long long st) {
  float a = 0.f;
  int s = 0;
  for (; s + 8 <= S; s += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(long long)(s + j) * st];
#pragma unroll
    for (int j = 0; j < 8; ++j) a += v[j];
  }
  for (; s < S; ++s) a += p[(long long)s * st];
  return a;
}
__global__ __launch_bounds__(256) void k_pred_dw_reduce(const float* __restrict__ part, const float* __restrict__ colpart, int S, int n_out,
                                                        float* __restrict__ dW, float* __restrict__ db) {
  const int total = n_out * kPredK + n_out;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const bool is_b = e >= n_out * kPredK;
    float a = 0.f;
    if (is_b) {
      const int o = e - n_out * kPredK;
      a += rg_slice_sum(colpart + o, S, 128);
      if (db) db[o] += a;
    } else {
      a += rg_slice_sum(part + e, S, 128ll * kPredK);
      dW[e] += a;
    }
  }
}

template <typename K>
int rg_set_lds(K kernel, int bytes) {
  GD_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return 0;
}

bool shapes_ok(int cin, int cout, int s) { return (cin == 128 || cin == 256) && cout == 128 && (s == 1 || s == 2 || s == 4); }

}  // namespace

extern "C" size_t gdmae_deconv_rows_packed_bytes(int cin, int cout, int s) { return (size_t)cin * cout * s * s * 2; }

extern "C" int gdmae_deconv_rows_pack(const float* weight, int cin, int cout, int s, void* packed_fwd, void* packed_bwd, void* stream) {
  GD_REQUIRE(shapes_ok(cin, cout, s), "deconv_rows: cin 128 / 256, cout 128, stride 1 / 2 / 4");
  const int total = cin * cout * s * s / 8;
  hipLaunchKernelGGL(k_deconv_pack, dim3(gd_div_up(2ll * total, 256) < 1024 ? gd_div_up(2ll * total, 256) : 1024), dim3(256), 0, (hipStream_t)stream,
                     weight, cin, cout, s * s, (uint4*)packed_fwd, (uint4*)packed_bwd);
  GD_LAUNCH_CHECK();
  return 0;
}

// P (n, s*s*cout) bf16 = X (n, cin) bf16 x the forward image
extern "C" int gdmae_deconv_rows_fwd(const void* X, long long n, int cin, int cout, int s, const void* packed_fwd, void* P, void* stream) {
  GD_REQUIRE(shapes_ok(cin, cout, s), "deconv_rows: cin 128 / 256, cout 128, stride 1 / 2 / 4");
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int N = s * s * cout;
  RgArgs A{(const unsigned short*)X, (const uint4*)packed_fwd, (unsigned short*)P, n, N};
  GdTimed timed(GD_T_ROWS_GEMM, st, 2.0 * n * (cin + N) + 2.0 * cin * N, 2.0 * n * cin * N);
#define RG_CASE(K_, NSL_)                                                                                       \
  {                                                                                                             \
    constexpr int rows = RgRows<NSL_>::value;                                                                   \
    constexpr int lds = rows * ((K_ > NSL_ ? K_ : NSL_) * 2 + 16);                                              \
    static bool once = false;                                                                                   \
    if (!once) { if (int rc = rg_set_lds(k_rows_gemm<K_, NSL_>, lds)) return rc; once = true; }                 \
    hipLaunchKernelGGL((k_rows_gemm<K_, NSL_>), dim3((unsigned)gd_div_up(n, rows), N / NSL_), dim3(512), lds, st, A); \
  }
  if (cin == 128 && N == 128) RG_CASE(128, 128)
  else if (cin == 128) RG_CASE(128, 512)
  else if (N == 128) RG_CASE(256, 128)
  else RG_CASE(256, 512)
#undef RG_CASE
  GD_LAUNCH_CHECK();
  return 0;
}

// dX (n, cin) bf16 = dP (n, s*s*cout) bf16 x the input-gradient image
extern "C" int gdmae_deconv_rows_bwd_input(const void* dP, long long n, int cin, int cout, int s, const void* packed_bwd, void* dX, void* stream) {
  GD_REQUIRE(shapes_ok(cin, cout, s), "deconv_rows: cin 128 / 256, cout 128, stride 1 / 2 / 4");
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int KT = s * s * cout;
  RkArgs A{(const unsigned short*)dP, (const uint4*)packed_bwd, (unsigned short*)dX, n, KT};
  GdTimed timed(GD_T_ROWS_GEMM, st, 2.0 * n * (cin + KT) + 2.0 * cin * KT, 2.0 * n * cin * KT);
#define RK_CASE(KC_, ND_)                                                                                       \
  {                                                                                                             \
    constexpr int rows = RgRows<ND_>::value;                                                                    \
    constexpr int lds = rows * ((KC_ > ND_ ? KC_ : ND_) * 2 + 16);                                              \
    static bool once = false;                                                                                   \
    if (!once) { if (int rc = rg_set_lds(k_rows_gemm_kc<KC_, ND_>, lds)) return rc; once = true; }              \
    hipLaunchKernelGGL((k_rows_gemm_kc<KC_, ND_>), dim3((unsigned)gd_div_up(n, rows)), dim3(512), lds, st, A);  \
  }
  if (KT == 128 && cin == 128) RK_CASE(128, 128)
  else if (KT == 128) RK_CASE(128, 256)
  else if (cin == 128) RK_CASE(512, 128)
  else RK_CASE(512, 256)
#undef RK_CASE
  GD_LAUNCH_CHECK();
  return 0;
}

// weight.grad (cin, cout, s, s) fp32 += X^T dP; workspace: gdmae_deconv_rows_dw_workspace_bytes(n, cin, cout, s)
static int dw_slices(long long n, int cin, int N, long long* n_pad) { return gd_dw_pick(n, (cin / 128) * (N / 128), 512, n_pad); }
extern "C" size_t gdmae_deconv_rows_dw_workspace_bytes(long long n, int cin, int cout, int s) {
  long long n_pad = 0;
  const int S = dw_slices(n, cin, s * s * cout, &n_pad);
  return gd_align((size_t)S * cin * s * s * cout * sizeof(float));
}
extern "C" int gdmae_deconv_rows_bwd_weight(const void* X, const void* dP, long long n, int cin, int cout, int s, float* dW, void* workspace,
                                            void* stream) {
  GD_REQUIRE(shapes_ok(cin, cout, s), "deconv_rows: cin 128 / 256, cout 128, stride 1 / 2 / 4");
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int N = s * s * cout;
  long long n_pad = 0;
  const int S = dw_slices(n, cin, N, &n_pad);
  GdDwGroup Gp;
  Gp.n_jobs = 1;
  Gp.job[0] = GdDwJob{X, dP, cin, N, (float*)workspace, nullptr, 0, nullptr, 0, 0};
  Gp.guard_rows = 1;                                     // neither operand is allocated beyond n rows
  if (int rc = gd_dw_grouped_s(st, Gp, n_pad, n, S)) return rc;
  hipLaunchKernelGGL(k_deconv_dw_reduce, dim3(gd_div_up((long long)cin * N, 256)), dim3(256), 0, st, (const float*)workspace, S, cin, cout, s * s, dW);
  GD_LAUNCH_CHECK();
  return 0;
}

// ---- prediction head (see k_pred_fwd): K = 128 inputs (fp32 rows), n_out <= 64 outputs, n_out % 8 == 0 ---------------------------
// packed: forward image bf16(W) | input-gradient image | bf16 bias (64, padded to 256 B) | forward image of the remainder | fp32 bias (64)
extern "C" size_t gdmae_pred_head_packed_bytes(void) { return 3 * (size_t)kPredK * kPredN * 2 + 256 + 256; }
static inline const uint4* pred_fwd_lo(const void* packed) { return (const uint4*)((const char*)packed + 2 * (size_t)kPredK * kPredN * 2 + 256); }
static inline const float* pred_bias_f(const void* packed) { return (const float*)((const char*)packed + 3 * (size_t)kPredK * kPredN * 2 + 256); }
extern "C" int gdmae_pred_head_pack(const float* weight, const float* bias, int n_in, int n_out, void* packed, void* stream) {
  GD_REQUIRE(n_in == kPredK && n_out >= 8 && n_out <= kPredN && n_out % 8 == 0, "pred_head: 128 inputs, 8..64 outputs (multiple of 8: the weight gradient reads dY in 16-byte chunks)");
  uint4* fwd = (uint4*)packed;
  uint4* bwd = fwd + kPredK * kPredN / 8;
  unsigned short* b16 = (unsigned short*)(bwd + kPredK * kPredN / 8);
  hipLaunchKernelGGL(k_pred_pack, dim3(8), dim3(256), 0, (hipStream_t)stream, weight, bias, n_out, fwd, (uint4*)pred_fwd_lo(packed), bwd, b16,
                     (float*)pred_bias_f(packed));
  GD_LAUNCH_CHECK();
  return 0;
}
extern "C" int gdmae_pred_head_fwd(const float* X, long long n, int n_out, const void* packed, void* Y, void* X_bf16, float* Y_f32, void* stream) {
  if (n <= 0) return 0;
  const uint4* fwd = (const uint4*)packed;
  PhArgs A{X, fwd, pred_fwd_lo(packed), pred_bias_f(packed), (unsigned short*)Y, n, n_out, (unsigned short*)X_bf16, Y_f32};
  constexpr int lds = 2 * kPredRows * (kPredK * 2 + 16);
  static bool once = false;
  if (!once) { if (int rc = rg_set_lds(k_pred_fwd, lds)) return rc; once = true; }
  GdTimed timed(GD_T_ROWS_GEMM, (hipStream_t)stream, (double)n * (4.0 * kPredK + (Y ? 2.0 : 0.0) * n_out + (Y_f32 ? 4.0 : 0.0) * n_out + (X_bf16 ? 2.0 * kPredK : 0.0)), 6.0 * n * kPredK * kPredN);
  hipLaunchKernelGGL(k_pred_fwd, dim3((unsigned)gd_div_up(n, kPredRows)), dim3(512), lds, (hipStream_t)stream, A);
  GD_LAUNCH_CHECK();
  return 0;
}
static int pred_slices(long long n, long long* n_pad) { return gd_dw_pick(n, 1, 512, n_pad); }
extern "C" size_t gdmae_pred_head_bwd_workspace_bytes(long long n) {
  long long n_pad = 0;
  const int S = pred_slices(n, &n_pad);
  return gd_align((size_t)S * 128 * kPredK * sizeof(float)) + gd_align((size_t)S * 128 * sizeof(float));
}
// dY (n, n_out) bf16, or fp32 with dy_f32 = 1 (rounded in the input-gradient launch, the rounded rows go to dY_bf16 (n, n_out));
// X_bf16: the (n, 128) bf16 rows gdmae_pred_head_fwd wrote; dX (n, 128) fp32 (may be null), dW (n_out, 128) / db (n_out) fp32
// ACCUMULATED (db may be null)
extern "C" int gdmae_pred_head_bwd(const void* dY, int dy_f32, void* dY_bf16, const float* scale_a, const float* scale_b, const void* X_bf16,
                                   long long n, int n_out, const void* packed, float* dX, float* dW, float* db, void* workspace, void* stream) {
  if (n <= 0) return 0;
  GD_REQUIRE(n_out >= 8 && n_out <= kPredN && n_out % 8 == 0, "pred_head_bwd: 8..64 outputs (multiple of 8)");
  GD_REQUIRE(!dy_f32 || dY_bf16 != nullptr, "pred_head_bwd: an fp32 dY needs the (n, n_out) bf16 buffer its rounded rows are written to");
  hipStream_t st = (hipStream_t)stream;
  const uint4* bwd = (const uint4*)packed + kPredK * kPredN / 8;
  if (dX || dy_f32) {
    PbArgs A{dY, bwd, dX, n, n_out, dy_f32, (unsigned short*)dY_bf16, scale_a, scale_b};
    constexpr int lds = 64 * (kPredN * 2 + 16);
    GdTimed timed(GD_T_ROWS_GEMM, st, (double)n * (2.0 * n_out + 4.0 * kPredK), 2.0 * n * kPredK * kPredN);
    hipLaunchKernelGGL(k_pred_bwd_input, dim3((unsigned)gd_div_up(n, 64)), dim3(512), lds, st, A);
    GD_LAUNCH_CHECK();
  }
  // dW = dY^T X as one 128 x 128 tile of the grouped TN kernel: G = dY with its 48-element rows read as 128 zero-padded columns
  long long n_pad = 0;
  const int S = pred_slices(n, &n_pad);
  float* part = (float*)workspace;
  float* colpart = (float*)((char*)workspace + gd_align((size_t)S * 128 * kPredK * sizeof(float)));
  GdDwGroup Gp;
  Gp.n_jobs = 1;
  Gp.job[0] = GdDwJob{dy_f32 ? dY_bf16 : dY, X_bf16, 128, kPredK, part, colpart, 0, nullptr, 0, 0, n_out, n_out};
  Gp.guard_rows = 1;
  if (int rc = gd_dw_grouped_s(st, Gp, n_pad, n, S)) return rc;
  hipLaunchKernelGGL(k_pred_dw_reduce, dim3(gd_div_up(n_out * kPredK + n_out, 256)), dim3(256), 0, st, (const float*)part, (const float*)colpart, S,
                     n_out, dW, db);
  GD_LAUNCH_CHECK();
  return 0;
}
