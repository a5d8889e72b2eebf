// Token GEMMs of the SRA encoder layers with fused row epilogues (reference
// pcdet/models/model_utils/sst_basic_block.py:57-84: linear1 -> GELU -> linear2 -> residual + LayerNorm, the attention
// in/out projections and their input-gradient counterparts).
//
//   Y (rows, N) = X (rows, K) W^T (N, K)  [+ bias]  -> epilogue
//
// These products are skinny: 20-40 k token rows against K, N <= 512, i.e. 8-67 MFLOP per 64-row tile and a weight
// matrix of at most 512 KB - they are bound by the activation traffic and by launch count, not by the matrix cores.
// One workgroup (8 wavefronts) therefore owns 64 token rows and the FULL output width, so everything the reference does
// row-wise after the product happens before the rows leave the CU:
//   TG_PLAIN     bias (optional), bf16 store
//   TG_GELU      h = . + bias stored, gelu_erf(h) stored next to it             (linear1 + activation)
//   TG_GELU_BWD  dh = . * gelu'(h)                                              (input gradient of linear2 + activation)
//   TG_RES_LN    y = LayerNorm(residual + bf16(. + bias)) in fp32, row statistics, optional bf16 copies y / y + pos
//                (out-projection or linear2 + the post-norm of the layer, and the next layer's q/k/v operands)
//   TG_LN_BWD    g = dy + [dy2] + bf16(.) is the gradient of a LayerNorm OUTPUT; dx = LayerNorm backward of g through
//                LN(a + b) with the saved row statistics, in fp32 (+ bf16 copy), and one partial row of dgamma / dbeta /
//                column sums of dx per workgroup (the input-gradient GEMM that feeds a post-norm's backward: linear1 of
//                the FFN -> LayerNorm 1, the v projection of the next layer -> LayerNorm 2)
// Numerics are those of the unfused sequence (GEMM output rounded to bf16, then the row kernels of layernorm.hip /
// encoder_layer.hip on the rounded values): the fused path is bit-compatible with it up to the fp32 accumulation order
// of the product.
//
// MFMA mapping (v_mfma_f32_32x32x16_bf16, Y^T = W X^T): A = weights, pre-packed once per optimizer step in fragment
// order (tok_gemm_pack) so a wavefront streams its 1 KB fragments straight from L2 into VGPRs, four k-steps ahead, no
// LDS and no barrier in the K loop; B = the 64-row activation tile, loaded once with 16-byte coalesced accesses into an
// LDS image with a (2K + 16)-byte row pitch (conflict-free ds_read_b128).  Accumulators go through an LDS staging tile
// (channel quads -> row-major bf16) so that the epilogue reads / writes whole rows with 16-byte accesses.
#include "tok_tiles.h"

// ------------------------------------------------------------------------------------------------
// weight packing: dst[(ks * MB + mb) * 64 + lane] = 8 bf16 of row mb * 32 + (lane & 31), columns ks * 16 + (lane >> 5) * 8 ..
// of the (M, K) matrix  A[r][c] = transpose ? src[c * M + r] : src[r * ld + c]   (fp32 master weights, rounded here)
// jobs: device table of n_jobs x 6 int64 {src, dst, M, K, ld, transpose | (es << 2)}; blockIdx.y = job.  es (default 0 = 1): element
// stride of the unit dimension, A[r][c] = transpose ? src[c * ld + r * es] : src[r * ld + c * es]  (a (Cout, Cin, 3, 3) convolution
// weight read per tap has es = 9)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tg_pack(const long long* __restrict__ jobs) {
  const long long* J = jobs + (long long)blockIdx.y * 6;
  const float* src = (const float*)J[0];
  uint4* dst = (uint4*)J[1];
  const int M = (int)J[2], K = (int)J[3], ld = (int)J[4], tr = (int)(J[5] & 1);
  const int esf = (int)((J[5] >> 2) & 0xFFFF), es = esf > 0 ? esf : 1;
  if (J[5] & (1ll << 20)) {
    // layer_v3.hip: fragments of v_mfma_f32_16x16x32_bf16 (16 rows x 32 k; lane = row % 16 + 16 g holds 8 k-slots), k-step major.
    // Natural k order: slot 8 g + j = column 32 ks + 8 g + j.  Chained (bit 21): the B operand is the previous product's accumulator
    // set, slot 8 g + j = column 32 ks + 16 (j / 4) + 4 g + j % 4.
    const bool chained = (J[5] & (1ll << 21)) != 0;
    const int MB16 = M / 16, total3 = (K / 32) * MB16 * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total3; i += gridDim.x * blockDim.x) {
      const int lane = i & 63, mb = (i >> 6) % MB16, ks = (i >> 6) / MB16;
      const int r = mb * 16 + (lane & 15), g = lane >> 4;
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = chained ? ks * 32 + 16 * (j >> 2) + 4 * g + (j & 3) : ks * 32 + 8 * g + j;
        f[j] = tr ? src[(long long)c * ld + (long long)r * es] : src[(long long)r * ld + (long long)c * es];
      }
      dst[i] = tg_pack8(f);
    }
    return;
  }
  const int MB = M / 32, total = (K / 16) * MB * 64;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int lane = i & 63, mb = (i >> 6) % MB, ks = (i >> 6) / MB;
    const int r = mb * 32 + (lane & 31), c = ks * 16 + (lane >> 5) * 8;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = tr ? src[(long long)(c + j) * ld + (long long)r * es] : src[(long long)r * ld + (long long)(c + j) * es];
    dst[i] = tg_pack8(f);
  }
}

extern "C" int gdmae_tok_gemm_pack(const long long* jobs_dev, int n_jobs, void* stream) {
  if (n_jobs <= 0) return 0;
  hipLaunchKernelGGL(k_tg_pack, dim3(32, n_jobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
struct TgArgs {
  const unsigned short* X;      // (n_pad, K) bf16
  const uint4* Wp;              // packed (N, K)
  const unsigned short* bias;   // (N) bf16 or null
  long long n, n_pad;
  unsigned short* out0;         // PLAIN / GELU (h) / GELU_BWD (dh): (n_pad, N) bf16
  unsigned short* out1;         // GELU: gelu(h)
  const unsigned short* aux;    // GELU_BWD: h (n_pad, N)
  // RES_LN
  const float* res;             // (n, N) fp32 residual stream
  const float* gamma;
  const float* beta;
  float eps;
  float* y;                     // (n, N) fp32
  int y_cached;                 // RES_LN: the next launch reads y again (LayerNorm 1 -> feed-forward block): plain store instead of streaming
  float* stats;                 // (n, 2) mean, rstd
  unsigned short* y_bf;         // optional (n_pad, N) bf16 copy of y
  const float* pos_table;       // optional: ypos_bf = bf16(y + pos_table[tok_pos[row]])
  const int* tok_pos;
  unsigned short* ypos_bf;
  unsigned short* f_out;        // optional (n_pad, N): the rounded branch output bf16(. + bias) the LayerNorm backward re-reads
  // LN_BWD: res = dy (n, N) fp32, aux = dy2 (n_pad, N) bf16 or null, stats (n, 2) READ, gamma; y = dx (n, N) fp32, y_bf = bf16 copy
  const float* ln_a;            // (n, N) fp32: first addend of the LayerNorm input
  const unsigned short* ln_b;   // (n_pad, N) bf16: second addend
  float* part;                  // (gridDim.x, 3, N) fp32: dgamma, dbeta, column sums of dx of this workgroup's rows
};


// rows per workgroup: 32 for the wide outputs (one 32-row MFMA block per wavefront: twice the workgroups, half the
// per-workgroup latency - the kernels are bound by how many load / compute / store phases overlap on a CU, not by bytes),
// 64 for ND = 128 where the 8 wavefronts would otherwise not all have an output block
template <int ND>
struct TgRows {
  static constexpr int value = ND >= 256 ? 32 : 64;
};
// ... per epilogue: the plain products (q / k / v projections: no row pass, nothing but the weight image re-streamed per tile
// competes with their rows) take TG_PLAIN_ROWS at ND = 256 (experiment switch)
#ifndef TG_PLAIN_ROWS
#define TG_PLAIN_ROWS 64
#endif
template <int ND, int EPI>
struct TgRowsE {
  static constexpr int value = (EPI == TG_PLAIN && ND >= 256) ? TG_PLAIN_ROWS : TgRows<ND>::value;
};

// One row tile.  tile: index of the tile (rows tile * ROWS ..); mbs: 32-channel blocks per k-step of the packed weight image
// A.Wp points into (MB for a whole image, more when the workgroup owns a channel slice of a wider one); ldo: row pitch of out0
// in elements (plain epilogue only)
template <int KD, int ND, int EPI>
__device__ __forceinline__ void tg_tile(const TgArgs& A, const long long tile, const int mbs, const int ldo) {
  constexpr int ROWS = TgRowsE<ND, EPI>::value;   // 4 waves per SIMD = two workgroups per CU: <= 128 VGPRs
  constexpr int KS = KD / 16;                       // k-steps
  constexpr int MB = ND / 32;                       // 32-channel blocks of the output
  constexpr int MPW = MB >= TG_WAVES ? MB / TG_WAVES : 1;   // channel blocks per wavefront
  constexpr int NPW = MB >= TG_WAVES ? ROWS / 32 : 1;       // 32-row blocks per wavefront
  constexpr int XP = KD * 2 + 16;                   // LDS row pitch of the activation tile (bytes)
  constexpr int SP = ND * 2 + 16;                   // LDS row pitch of the staging tile
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long row0 = tile * ROWS;

  // ---- weight fragments of the first TG_PF k-steps: in flight while the activation tile is loaded
  const int mb0 = MB >= TG_WAVES ? wv : (wv >> 1);          // first channel block of this wavefront (then + TG_WAVES)
  const int nb0 = MB >= TG_WAVES ? 0 : (wv & 1);
  const uint4* __restrict__ wp = A.Wp + (size_t)mb0 * 64 + lane;   // + (ks * MB + j * TG_WAVES) * 64
  TgFrag wr[TG_PF + 1][MPW];
#pragma unroll
  for (int ks = 0; ks < TG_PF; ++ks)
#pragma unroll
    for (int j = 0; j < MPW; ++j) wr[ks][j].q = wp[((size_t)ks * mbs + j * TG_WAVES) * 64];

  // ---- row-epilogue operands that do not depend on the product (residual rows, positional-table rows, GELU-backward
  //      pre-activations) are requested NOW: their latency hides behind the tile load and the K loop instead of being paid
  //      once per epilogue pass (the LayerNorm epilogue ran at half its HBM roof when it fetched them just in time)
  constexpr int LN_LPR = ND / 4, LN_RPP = 512 / (LN_LPR > 0 ? LN_LPR : 1), LN_PASSES = (ROWS / LN_RPP) > 0 ? ROWS / LN_RPP : 1;
  float4 res_pf[EPI == TG_RES_LN ? LN_PASSES : 1];
  int pos_pf[EPI == TG_RES_LN ? LN_PASSES : 1];
  if (EPI == TG_RES_LN) {
    const int c0 = 4 * (tid % LN_LPR), r = tid / LN_LPR;
#pragma unroll
    for (int p = 0; p < LN_PASSES; ++p) {
      const long long row = row0 + p * LN_RPP + r;
      const long long rr = row < A.n ? row : A.n - 1;
      res_pf[p] = *(const float4*)(A.res + rr * ND + c0);
      pos_pf[p] = A.ypos_bf ? A.tok_pos[rr] : 0;
    }
  }

  float4 dy_pf[EPI == TG_LN_BWD ? LN_PASSES : 1], a_pf[EPI == TG_LN_BWD ? LN_PASSES : 1];
  uint2 b_pf[EPI == TG_LN_BWD ? LN_PASSES : 1], d2_pf[EPI == TG_LN_BWD ? LN_PASSES : 1];
  float2 st_pf[EPI == TG_LN_BWD ? LN_PASSES : 1];
  if (EPI == TG_LN_BWD) {
    const int c0 = 4 * (tid % LN_LPR), r = tid / LN_LPR;
#pragma unroll
    for (int p = 0; p < LN_PASSES; ++p) {
      const long long row = row0 + p * LN_RPP + r;
      const long long rr = row < A.n ? row : A.n - 1;
      dy_pf[p] = *(const float4*)(A.res + rr * ND + c0);
      a_pf[p] = tg_ld_f4_once(A.ln_a + rr * ND + c0);
      b_pf[p] = tg_ld_u2_once(A.ln_b + rr * ND + c0);
      d2_pf[p] = A.aux ? *(const uint2*)(A.aux + rr * ND + c0) : make_uint2(0u, 0u);
      st_pf[p] = *(const float2*)(A.stats + rr * 2);
    }
  }

  // ---- activation tile: 64 rows x KD bf16, 16 bytes per thread and access
  {
    constexpr int CPR = KD / 8;                     // 16-byte chunks per row
    constexpr int RPP = 512 / CPR;                  // rows per pass
    const int c = tid % CPR, r = tid / CPR;
#pragma unroll
    for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
      const int row = p * RPP + r;
      const uint4 q = *(const uint4*)(A.X + (row0 + row) * KD + c * 8);
      *(uint4*)(lds + row * XP + c * 16) = q;
    }
  }
  __syncthreads();

  f32x16 acc[MPW][NPW];
#pragma unroll
  for (int j = 0; j < MPW; ++j)
#pragma unroll
    for (int b = 0; b < NPW; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][b][i] = 0.f;
  // bias fragments: requested together, unconditionally (from a readable stand-in when there is no bias, then masked) and before the
  // product - a runtime `if (bias)` around each load is a join with a load in flight: the compiler drained the load counter after
  // every one of the four (serial L2 round trips behind the product)
  uint2 bq[MPW][4];
  const unsigned bmask = (EPI != TG_GELU_BWD && A.bias) ? 0xFFFFFFFFu : 0u;
  {
    const unsigned short* bp = bmask ? A.bias : (const unsigned short*)A.Wp;
#pragma unroll
    for (int j = 0; j < MPW; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) bq[j][q] = *(const uint2*)(bp + (mb0 + j * TG_WAVES) * 32 + 4 * (lane >> 5) + 8 * q);
  }
  {
    const unsigned char* lb = lds + ((nb0 * 32) + (lane & 31)) * XP + (lane >> 5) * 16;
    TgFrag sf[2][NPW];
#pragma unroll
    for (int b = 0; b < NPW; ++b) sf[0][b].q = *(const uint4*)(lb + b * 32 * XP);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + TG_PF < KS) {
#pragma unroll
        for (int j = 0; j < MPW; ++j) wr[(ks + TG_PF) % (TG_PF + 1)][j].q = wp[((size_t)(ks + TG_PF) * mbs + j * TG_WAVES) * 64];
      }
      if (ks + 1 < KS) {
#pragma unroll
        for (int b = 0; b < NPW; ++b) sf[(ks + 1) & 1][b].q = *(const uint4*)(lb + b * 32 * XP + (ks + 1) * 32);
      }
#pragma unroll
      for (int j = 0; j < MPW; ++j)
#pragma unroll
        for (int b = 0; b < NPW; ++b)
          acc[j][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[ks % (TG_PF + 1)][j].v, sf[ks & 1][b].v, acc[j][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();      // every wavefront is done with the activation tile: the staging tile may overwrite it

  // ---- accumulators (+ bias) -> bf16 -> staging tile [row][channel]
#pragma unroll
  for (int j = 0; j < MPW; ++j) {
    const int cb = (mb0 + j * TG_WAVES) * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int b = 0; b < NPW; ++b) {
      const int row = (nb0 + b) * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[j][b][4 * q + e];
        {
          const unsigned bx = bq[j][q].x & bmask, by = bq[j][q].y & bmask;      // + 0 without a bias
          v[0] += __uint_as_float(bx << 16); v[1] += __uint_as_float(bx & 0xFFFF0000u);
          v[2] += __uint_as_float(by << 16); v[3] += __uint_as_float(by & 0xFFFF0000u);
        }
        uint2 o;
        o.x = tg_pack2(v[0], v[1]);
        o.y = tg_pack2(v[2], v[3]);
        *(uint2*)(lds + row * SP + (cb + 8 * q) * 2) = o;
      }
    }
  }
  __syncthreads();

  // ---- row epilogue
  if (EPI == TG_RES_LN) {
    constexpr int LPR = ND / 4;                     // lanes per row (4 consecutive columns per lane), as layernorm.hip
    constexpr int RPP = 512 / LPR;
    const int c0 = 4 * (tid % LPR), r = tid / LPR;
    const float4 g4 = *(const float4*)(A.gamma + c0), b4 = *(const float4*)(A.beta + c0);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
      const int rl = p * RPP + r;
      const long long row = row0 + rl;
      const bool live = row < A.n;
      const long long e = (live ? row : A.n - 1) * ND + c0;
      const float4 a4 = res_pf[p];
      const uint2 fq = *(const uint2*)(lds + rl * SP + c0 * 2);
      if (A.f_out) TG_ST_U2(A.f_out + (row0 + rl) * ND + c0, fq);
      float s[4] = {a4.x, a4.y, a4.z, a4.w};
      s[0] += __uint_as_float(fq.x << 16); s[1] += __uint_as_float(fq.x & 0xFFFF0000u);
      s[2] += __uint_as_float(fq.y << 16); s[3] += __uint_as_float(fq.y & 0xFFFF0000u);
      const float mean = tg_group_sum<LPR>((s[0] + s[1]) + (s[2] + s[3])) * (1.f / ND);
      float sq = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dlt = s[k] - mean;
        sq = fmaf(dlt, dlt, sq);
      }
      const float rstd = rsqrtf(tg_group_sum<LPR>(sq) * (1.f / ND) + A.eps);
      if (!live) continue;
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (s[k] - mean) * rstd * g[k] + bt[k];
      if (A.y_cached) *(float4*)(A.y + e) = make_float4(o[0], o[1], o[2], o[3]);
      else TG_ST_F4(A.y + e, o[0], o[1], o[2], o[3]);
      if (A.y_bf) {
        uint2 q;
        q.x = tg_pack2(o[0], o[1]); q.y = tg_pack2(o[2], o[3]);
        *(uint2*)(A.y_bf + e) = q;
      }
      if (A.ypos_bf) {
        const float4 p4 = *(const float4*)(A.pos_table + (long long)pos_pf[p] * ND + c0);
        uint2 q;
        q.x = tg_pack2(o[0] + p4.x, o[1] + p4.y); q.y = tg_pack2(o[2] + p4.z, o[3] + p4.w);
        *(uint2*)(A.ypos_bf + e) = q;
      }
      if (c0 == 0) *(float2*)(A.stats + row * 2) = make_float2(mean, rstd);
    }
  } else if (EPI == TG_LN_BWD) {
    // arithmetic and association order of k_add_ln_bwd (layernorm.hip) on g = (dy + dy2) + bf16(product)
    constexpr int LPR = ND / 4;
    constexpr int RPP = 512 / LPR;
    const int c0 = 4 * (tid % LPR), r = tid / LPR;
    const float4 g4 = *(const float4*)(A.gamma + c0);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f}, dsx[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
      const int rl = p * RPP + r;
      const long long row = row0 + rl;
      const bool live = row < A.n;
      const long long e = (live ? row : A.n - 1) * ND + c0;
      const uint2 fq = *(const uint2*)(lds + rl * SP + c0 * 2);
      float d[4] = {dy_pf[p].x, dy_pf[p].y, dy_pf[p].z, dy_pf[p].w};
      if (A.aux) {
        d[0] += __uint_as_float(d2_pf[p].x << 16); d[1] += __uint_as_float(d2_pf[p].x & 0xFFFF0000u);
        d[2] += __uint_as_float(d2_pf[p].y << 16); d[3] += __uint_as_float(d2_pf[p].y & 0xFFFF0000u);
      }
      d[0] += __uint_as_float(fq.x << 16); d[1] += __uint_as_float(fq.x & 0xFFFF0000u);
      d[2] += __uint_as_float(fq.y << 16); d[3] += __uint_as_float(fq.y & 0xFFFF0000u);
      const float sa[4] = {a_pf[p].x, a_pf[p].y, a_pf[p].z, a_pf[p].w};
      const float sb[4] = {__uint_as_float(b_pf[p].x << 16), __uint_as_float(b_pf[p].x & 0xFFFF0000u), __uint_as_float(b_pf[p].y << 16),
                           __uint_as_float(b_pf[p].y & 0xFFFF0000u)};
      const float mean = st_pf[p].x, rstd = st_pf[p].y;
      float gy[4], xh[4], m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!live) d[k] = 0.f;
        xh[k] = (sa[k] + sb[k] - mean) * rstd;
        gy[k] = d[k] * g[k];
        m1 += gy[k];
        m2 = fmaf(gy[k], xh[k], m2);
        dg[k] = fmaf(d[k], xh[k], dg[k]);
        db[k] += d[k];
      }
      m1 = tg_group_sum<LPR>(m1) * (1.f / ND);
      m2 = tg_group_sum<LPR>(m2) * (1.f / ND);
      if (!live) continue;
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[k] = rstd * (gy[k] - m1 - xh[k] * m2);
        dsx[k] += o[k];
      }
      TG_ST_F4(A.y + e, o[0], o[1], o[2], o[3]);
      if (A.y_bf) {
        uint2 q;
        q.x = tg_pack2(o[0], o[1]); q.y = tg_pack2(o[2], o[3]);
        *(uint2*)(A.y_bf + e) = q;
      }
    }
    // one partial row per workgroup: the RPP row slots of a column meet in LDS in a fixed order
    __syncthreads();                                   // every thread is done with the staging tile
    float* red = reinterpret_cast<float*>(lds);        // [RPP][3][ND]
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[(r * 3 + 0) * ND + c0 + k] = dg[k];
      red[(r * 3 + 1) * ND + c0 + k] = db[k];
      red[(r * 3 + 2) * ND + c0 + k] = dsx[k];
    }
    __syncthreads();
    for (int c = tid; c < 3 * ND; c += 512) {
      float acc1 = 0.f;
#pragma unroll
      for (int q = 0; q < RPP; ++q) acc1 += red[q * 3 * ND + c];
      A.part[tile * 3 * ND + c] = acc1;
    }
  } else {
    constexpr int CPR = ND / 8;
    constexpr int RPP = 512 / CPR;
    const int c = tid % CPR, r = tid / CPR;
#pragma unroll
    for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
      const int rl = p * RPP + r;
      const long long e = (row0 + rl) * ND + c * 8;
      const uint4 q = *(const uint4*)(lds + rl * SP + c * 16);
      if (EPI == TG_PLAIN) {
        *(uint4*)(A.out0 + (row0 + rl) * ldo + c * 8) = q;
      } else if (EPI == TG_GELU) {
        *(uint4*)(A.out0 + e) = q;
        float v[8];
        tg_unpack8(q, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = tg_gelu(v[j]);
        *(uint4*)(A.out1 + e) = tg_pack8(v);
      } else {   // TG_GELU_BWD: dh = dg * (Phi(h) + h * phi(h));  optionally gelu(h) next to it (the operand of the weight
                 // gradient of the linear layer behind the GELU, when the forward kept it in LDS only: k_tok_ffn)
        const uint4 hq = tg_ld_u4_once(A.aux + e);
        float g[8], v[8];
        tg_unpack8(q, g);
        tg_unpack8(hq, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = g[j] * tg_gelu_grad(v[j]);
        *(uint4*)(A.out0 + e) = tg_pack8(g);
        if (A.out1) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = tg_gelu(v[j]);
          *(uint4*)(A.out1 + e) = tg_pack8(v);
        }
      }
    }
  }
}

template <int KD, int ND, int EPI>
__global__ __launch_bounds__(512, 4) void k_tok_gemm(TgArgs A) {
  tg_tile<KD, ND, EPI>(A, blockIdx.x, ND / 32, ND);
}

// Up to three plain products over the same row tiles in ONE launch (the q, k and v projections of a layer: q and k read the
// same operand rows).  The jobs of a row tile are adjacent workgroups of one XCD, so the shared operand tile is read from
// HBM once and from that XCD's L2 afterwards.
struct TgJob {
  const unsigned short* X;
  const uint4* Wp;              // first fragment of this job's channel slice
  const unsigned short* bias;   // of the slice, or null
  unsigned short* out;          // first column of the slice
  int mbs, ldo;
};
struct TgMulti {
  TgJob j0, j1, j2;
  int n_jobs;
  long long tiles, n_pad;
};

template <int KD, int ND>
__global__ __launch_bounds__(512, 4) void k_tok_gemm_multi(TgMulti M) {
  const int xcd = blockIdx.x & 7, g = blockIdx.x >> 3;
  const int job = g % M.n_jobs;
  const long long tile = (long long)(g / M.n_jobs) * 8 + xcd;
  if (tile >= M.tiles) return;
  TgArgs A = {};
  A.X = job == 0 ? M.j0.X : (job == 1 ? M.j1.X : M.j2.X);
  A.Wp = job == 0 ? M.j0.Wp : (job == 1 ? M.j1.Wp : M.j2.Wp);
  A.bias = job == 0 ? M.j0.bias : (job == 1 ? M.j1.bias : M.j2.bias);
  A.out0 = job == 0 ? M.j0.out : (job == 1 ? M.j1.out : M.j2.out);
  A.n = A.n_pad = M.n_pad;
  const int mbs = job == 0 ? M.j0.mbs : (job == 1 ? M.j1.mbs : M.j2.mbs);
  const int ldo = job == 0 ? M.j0.ldo : (job == 1 ? M.j1.ldo : M.j2.ldo);
  tg_tile<KD, ND, TG_PLAIN>(A, tile, mbs, ldo);
}

template <int KD, int ND>
static int tg_launch_multi(const TgMulti& M, hipStream_t st) {
  constexpr int ROWS = TgRowsE<ND, TG_PLAIN>::value;
  constexpr int lds = ROWS * ((KD > ND ? KD : ND) * 2 + 16);
  static bool once = false;
  if (!once) {
    GD_CHECK(hipFuncSetAttribute((const void*)k_tok_gemm_multi<KD, ND>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    once = true;
  }
  TgMulti A = M;
  A.tiles = M.n_pad / ROWS;
  const long long groups = (A.tiles + 7) / 8 * A.n_jobs;
  hipLaunchKernelGGL((k_tok_gemm_multi<KD, ND>), dim3((unsigned)(groups * 8)), dim3(512), lds, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}

template <int KD, int ND, int EPI>
static int tg_launch(const TgArgs& A, hipStream_t st) {
  constexpr int ROWS = TgRowsE<ND, EPI>::value;
  constexpr int lds_tile = ROWS * ((KD > ND ? KD : ND) * 2 + 16);
  constexpr int lds_red = EPI == TG_LN_BWD ? (512 / (ND / 4)) * 3 * ND * 4 : 0;
  constexpr int lds = lds_tile > lds_red ? lds_tile : lds_red;
  static bool once = false;
  if (!once) {
    GD_CHECK(hipFuncSetAttribute((const void*)k_tok_gemm<KD, ND, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    once = true;
  }
  hipLaunchKernelGGL((k_tok_gemm<KD, ND, EPI>), dim3((unsigned)(A.n_pad / ROWS)), dim3(512), lds, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}

template <int EPI>
static int tg_dispatch(int K, int N, const TgArgs& A, hipStream_t st) {
#define TG_CASE(k_, n_)                                       \
  if constexpr (!((EPI == TG_RES_LN || EPI == TG_LN_BWD) && n_ > 256)) { \
    if (K == k_ && N == n_) return tg_launch<k_, n_, EPI>(A, st); \
  }
  TG_CASE(128, 128);
  TG_CASE(128, 256);
  TG_CASE(256, 128);
  TG_CASE(256, 256);
  TG_CASE(256, 512);
  TG_CASE(512, 256);
#undef TG_CASE
  GD_REQUIRE(false, "tok_gemm: unsupported (K, N)");
}

// ------------------------------------------------------------------------------------------------
// The feed-forward block of a layer as ONE launch (sst_basic_block.py:79-84: linear1 -> GELU -> linear2 -> residual +
// LayerNorm 2):  h = X W1^T + b1 (bf16, stored: the backward differentiates the GELU at it),  g = gelu(h) lives in LDS only
// (32 rows x FF bf16 per workgroup: the second product reads its operand tile from there, the (n, FF) activation is neither
// written nor read back - the backward's GELU kernel re-creates it next to dh for the weight gradient),  then the RES_LN
// epilogue of the kernel above on  g W2^T + b2.  Same fragments, k order, rounding points and row arithmetic as
// k_tok_gemm<D, FF, TG_GELU> followed by k_tok_gemm<FF, D, TG_RES_LN>: bit-identical outputs.
// ------------------------------------------------------------------------------------------------
struct TgFfnArgs {
  const unsigned short* X;      // (n_pad, D) bf16
  const uint4* W1p;             // packed (FF, D)
  const unsigned short* b1;     // (FF) bf16 or null
  const uint4* W2p;             // packed (D, FF)
  const unsigned short* b2;     // (D) bf16 or null
  unsigned short* h;            // (n_pad, FF) bf16
  TgArgs L;                     // the RES_LN operands / outputs (X, Wp, bias, out* unused)
};

template <int D>
__global__ __launch_bounds__(512, 4) void k_tok_ffn(TgFfnArgs F) {
  constexpr int FF = 2 * D;
  constexpr int ROWS = D >= 256 ? 32 : 64;
  constexpr int XP = D * 2 + 16, HP = FF * 2 + 16, SP = D * 2 + 16;
  constexpr int KS1 = D / 16, MB1 = FF / 32, MPW1 = MB1 / TG_WAVES, NPW1 = ROWS / 32;     // product 1: every wavefront MPW1 channel blocks x all rows
  constexpr int KS2 = FF / 16, MB2 = D / 32;                                                // product 2: one block per wavefront
  static_assert(MPW1 * NPW1 == 2 && (MB2 == 8 || MB2 == 4), "tile shapes");
  extern __shared__ __align__(16) unsigned char lds[];
  unsigned char* const xl = lds;                     // activation tile, later the staging tile of product 2
  unsigned char* const hl = lds + ROWS * XP;         // h, then gelu(h): the operand tile of product 2
  const TgArgs& A = F.L;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long row0 = (long long)blockIdx.x * ROWS;

  // ---- weight fragments of product 1, first TG_PF k-steps
  const uint4* __restrict__ wp1 = F.W1p + (size_t)wv * 64 + lane;
  TgFrag wr[TG_PF + 1][MPW1];
#pragma unroll
  for (int ks = 0; ks < TG_PF; ++ks)
#pragma unroll
    for (int j = 0; j < MPW1; ++j) wr[ks][j].q = wp1[((size_t)ks * MB1 + j * TG_WAVES) * 64];
  // ---- LayerNorm operands (as tg_tile)
  constexpr int LPR = D / 4, LRPP = 512 / LPR, LPASS = ROWS / LRPP;
  const int lc0 = 4 * (tid % LPR), lr = tid / LPR;
  float4 res_pf[LPASS];
  int pos_pf[LPASS];
#pragma unroll
  for (int p = 0; p < LPASS; ++p) {
    const long long row = row0 + p * LRPP + lr;
    const long long rr = row < A.n ? row : A.n - 1;
    res_pf[p] = *(const float4*)(A.res + rr * D + lc0);
    pos_pf[p] = A.ypos_bf ? A.tok_pos[rr] : 0;
  }
  // ---- activation tile
  {
    constexpr int CPR = D / 8, RPP = 512 / CPR;
    const int c = tid % CPR, r = tid / CPR;
#pragma unroll
    for (int p = 0; p < ROWS / RPP; ++p) {
      const int row = p * RPP + r;
      *(uint4*)(xl + row * XP + c * 16) = *(const uint4*)(F.X + (row0 + row) * D + c * 8);
    }
  }
  __syncthreads();
  // ---- product 1
  f32x16 acc1[MPW1][NPW1];
#pragma unroll
  for (int j = 0; j < MPW1; ++j)
#pragma unroll
    for (int b = 0; b < NPW1; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc1[j][b][i] = 0.f;
  {
    const unsigned char* lb = xl + (lane & 31) * XP + (lane >> 5) * 16;
    TgFrag sf[2][NPW1];
#pragma unroll
    for (int b = 0; b < NPW1; ++b) sf[0][b].q = *(const uint4*)(lb + b * 32 * XP);
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      if (ks + TG_PF < KS1) {
#pragma unroll
        for (int j = 0; j < MPW1; ++j) wr[(ks + TG_PF) % (TG_PF + 1)][j].q = wp1[((size_t)(ks + TG_PF) * MB1 + j * TG_WAVES) * 64];
      }
      if (ks + 1 < KS1) {
#pragma unroll
        for (int b = 0; b < NPW1; ++b) sf[(ks + 1) & 1][b].q = *(const uint4*)(lb + b * 32 * XP + (ks + 1) * 32);
      }
#pragma unroll
      for (int j = 0; j < MPW1; ++j)
#pragma unroll
        for (int b = 0; b < NPW1; ++b)
          acc1[j][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[ks % (TG_PF + 1)][j].v, sf[ks & 1][b].v, acc1[j][b], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- weight fragments of product 2 (in flight behind the GELU pass)
  const int mb2 = MB2 >= TG_WAVES ? wv : (wv >> 1), nb2 = MB2 >= TG_WAVES ? 0 : (wv & 1);
  const uint4* __restrict__ wp2 = F.W2p + (size_t)mb2 * 64 + lane;
  TgFrag w2[TG_PF + 1];
#pragma unroll
  for (int ks = 0; ks < TG_PF; ++ks) w2[ks].q = wp2[(size_t)ks * MB2 * 64];
  // ---- h = bf16(acc + b1) -> LDS [row][channel]
#pragma unroll
  for (int j = 0; j < MPW1; ++j) {
    const int cb = (wv + j * TG_WAVES) * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int b = 0; b < NPW1; ++b) {
      const int row = b * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc1[j][b][4 * q + e];
        if (F.b1) {
          const uint2 bq = *(const uint2*)(F.b1 + cb + 8 * q);
          v[0] += __uint_as_float(bq.x << 16); v[1] += __uint_as_float(bq.x & 0xFFFF0000u);
          v[2] += __uint_as_float(bq.y << 16); v[3] += __uint_as_float(bq.y & 0xFFFF0000u);
        }
        uint2 o;
        o.x = tg_pack2(v[0], v[1]);
        o.y = tg_pack2(v[2], v[3]);
        *(uint2*)(hl + row * HP + (cb + 8 * q) * 2) = o;
      }
    }
  }
  __syncthreads();
  // ---- row pass: h leaves for HBM, gelu(h) replaces it in LDS
  {
    constexpr int CPR = FF / 8, RPP = 512 / CPR;
    const int c = tid % CPR, r = tid / CPR;
#pragma unroll
    for (int p = 0; p < ROWS / RPP; ++p) {
      const int rl = p * RPP + r;
      const uint4 q = *(const uint4*)(hl + rl * HP + c * 16);
      TG_ST_U4(F.h + (row0 + rl) * FF + c * 8, q);
      float v[8];
      tg_unpack8(q, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tg_gelu(v[j]);
      *(uint4*)(hl + rl * HP + c * 16) = tg_pack8(v);
    }
  }
  __syncthreads();
  // ---- product 2: operand tile = gelu(h) in LDS
  f32x16 acc2;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
  {
    const unsigned char* lb = hl + ((nb2 * 32) + (lane & 31)) * HP + (lane >> 5) * 16;
    TgFrag sf[2];
    sf[0].q = *(const uint4*)lb;
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {
      if (ks + TG_PF < KS2) w2[(ks + TG_PF) % (TG_PF + 1)].q = wp2[(size_t)(ks + TG_PF) * MB2 * 64];
      if (ks + 1 < KS2) sf[(ks + 1) & 1].q = *(const uint4*)(lb + (ks + 1) * 32);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2[ks % (TG_PF + 1)].v, sf[ks & 1].v, acc2, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- bf16(acc + b2) -> staging tile (the activation tile's region: nobody reads it any more)
  {
    const int cb = mb2 * 32 + 4 * (lane >> 5), row = nb2 * 32 + (lane & 31);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc2[4 * q + e];
      if (F.b2) {
        const uint2 bq = *(const uint2*)(F.b2 + cb + 8 * q);
        v[0] += __uint_as_float(bq.x << 16); v[1] += __uint_as_float(bq.x & 0xFFFF0000u);
        v[2] += __uint_as_float(bq.y << 16); v[3] += __uint_as_float(bq.y & 0xFFFF0000u);
      }
      uint2 o;
      o.x = tg_pack2(v[0], v[1]);
      o.y = tg_pack2(v[2], v[3]);
      *(uint2*)(xl + row * SP + (cb + 8 * q) * 2) = o;
    }
  }
  __syncthreads();
  // ---- residual + LayerNorm rows (the RES_LN epilogue of tg_tile)
  {
    const float4 g4 = *(const float4*)(A.gamma + lc0), b4 = *(const float4*)(A.beta + lc0);
    const float g[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int p = 0; p < LPASS; ++p) {
      const int rl = p * LRPP + lr;
      const long long row = row0 + rl;
      const bool live = row < A.n;
      const long long e = (live ? row : A.n - 1) * D + lc0;
      const float4 a4 = res_pf[p];
      const uint2 fq = *(const uint2*)(xl + rl * SP + lc0 * 2);
      if (A.f_out) TG_ST_U2(A.f_out + (row0 + rl) * D + lc0, fq);
      float sv[4] = {a4.x, a4.y, a4.z, a4.w};
      sv[0] += __uint_as_float(fq.x << 16); sv[1] += __uint_as_float(fq.x & 0xFFFF0000u);
      sv[2] += __uint_as_float(fq.y << 16); sv[3] += __uint_as_float(fq.y & 0xFFFF0000u);
      const float mean = tg_group_sum<LPR>((sv[0] + sv[1]) + (sv[2] + sv[3])) * (1.f / D);
      float sq = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dlt = sv[k] - mean;
        sq = fmaf(dlt, dlt, sq);
      }
      const float rstd = rsqrtf(tg_group_sum<LPR>(sq) * (1.f / D) + A.eps);
      if (!live) continue;
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = (sv[k] - mean) * rstd * g[k] + bt[k];
      TG_ST_F4(A.y + e, o[0], o[1], o[2], o[3]);
      if (A.y_bf) {
        uint2 q;
        q.x = tg_pack2(o[0], o[1]); q.y = tg_pack2(o[2], o[3]);
        *(uint2*)(A.y_bf + e) = q;
      }
      if (A.ypos_bf) {
        const float4 p4 = *(const float4*)(A.pos_table + (long long)pos_pf[p] * D + lc0);
        uint2 q;
        q.x = tg_pack2(o[0] + p4.x, o[1] + p4.y); q.y = tg_pack2(o[2] + p4.z, o[3] + p4.w);
        *(uint2*)(A.ypos_bf + e) = q;
      }
      if (lc0 == 0) *(float2*)(A.stats + row * 2) = make_float2(mean, rstd);
    }
  }
}

template <int D>
static int tg_launch_ffn(const TgFfnArgs& F, long long n_pad, hipStream_t st) {
  constexpr int ROWS = D >= 256 ? 32 : 64;
  constexpr int lds = ROWS * (D * 2 + 16) + ROWS * (4 * D + 16);
  static bool once = false;
  if (!once) {
    GD_CHECK(hipFuncSetAttribute((const void*)k_tok_ffn<D>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    once = true;
  }
  hipLaunchKernelGGL((k_tok_ffn<D>), dim3((unsigned)(n_pad / ROWS)), dim3(512), lds, st, F);
  GD_LAUNCH_CHECK();
  return 0;
}

// internal front end (encoder_layer.hip)
bool gd_tok_gemm_supported(int K, int N) {
  return (K == 128 && (N == 128 || N == 256)) || (K == 256 && (N == 128 || N == 256 || N == 512)) || (K == 512 && N == 256);
}
int gd_tok_gemm_plain(hipStream_t st, const void* X, const void* Wp, const void* bias, long long n_pad, int K, int N, void* out) {
  TgArgs A = {};
  A.X = (const unsigned short*)X; A.Wp = (const uint4*)Wp; A.bias = (const unsigned short*)bias; A.n = n_pad; A.n_pad = n_pad;
  A.out0 = (unsigned short*)out;
  GdTimed timed(GD_T_TOK_GEMM, st, 2.0 * n_pad * (K + N) + 2.0 * K * N, 2.0 * n_pad * K * N);
  return tg_dispatch<TG_PLAIN>(K, N, A, st);
}
// q | k = (X + pos) Wqk^T + b (n_pad, 2d) and v = X Wv^T + b (n_pad, d) in one launch; Wqk packed as one (2d, d) image
int gd_tok_gemm_qkv(hipStream_t st, const void* Xpos, const void* X, const void* Wp_qk, const void* Wp_v, const void* bias3,
                    long long n_pad, int d, void* qk, void* v) {
  TgMulti M = {};
  const unsigned short* b = (const unsigned short*)bias3;
  M.j0 = TgJob{(const unsigned short*)Xpos, (const uint4*)Wp_qk, b, (unsigned short*)qk, 2 * d / 32, 2 * d};
  M.j1 = TgJob{(const unsigned short*)Xpos, (const uint4*)Wp_qk + (size_t)(d / 32) * 64, b ? b + d : nullptr, (unsigned short*)qk + d,
               2 * d / 32, 2 * d};
  M.j2 = TgJob{(const unsigned short*)X, (const uint4*)Wp_v, b ? b + 2 * d : nullptr, (unsigned short*)v, d / 32, d};
  M.n_jobs = 3;
  M.n_pad = n_pad;
  GdTimed timed(GD_T_TOK_GEMM, st, 2.0 * n_pad * (2 * d + 3 * d) + 2.0 * 3 * d * d, 2.0 * n_pad * d * 3 * d);
  if (d == 128) return tg_launch_multi<128, 128>(M, st);
  if (d == 256) return tg_launch_multi<256, 256>(M, st);
  GD_REQUIRE(false, "tok_gemm_qkv: unsupported width");
}
int gd_tok_gemm_gelu(hipStream_t st, const void* X, const void* Wp, const void* bias, long long n_pad, int K, int N, void* h, void* gact) {
  TgArgs A = {};
  A.X = (const unsigned short*)X; A.Wp = (const uint4*)Wp; A.bias = (const unsigned short*)bias; A.n = n_pad; A.n_pad = n_pad;
  A.out0 = (unsigned short*)h; A.out1 = (unsigned short*)gact;
  GdTimed timed(GD_T_TOK_GEMM, st, 2.0 * n_pad * (K + 2 * N) + 2.0 * K * N, 2.0 * n_pad * K * N);
  return tg_dispatch<TG_GELU>(K, N, A, st);
}
int gd_tok_gemm_gelu_bwd(hipStream_t st, const void* dY, const void* Wp, const void* h, long long n_pad, int K, int N, void* dh, void* gact) {
  TgArgs A = {};
  A.X = (const unsigned short*)dY; A.Wp = (const uint4*)Wp; A.n = n_pad; A.n_pad = n_pad;
  A.out0 = (unsigned short*)dh; A.aux = (const unsigned short*)h; A.out1 = (unsigned short*)gact;
  GdTimed timed(GD_T_TOK_GEMM, st, 2.0 * n_pad * (K + (gact ? 3 : 2) * N) + 2.0 * K * N, 2.0 * n_pad * K * N);
  return tg_dispatch<TG_GELU_BWD>(K, N, A, st);
}
int gd_tok_gemm_res_ln(hipStream_t st, const void* X, const void* Wp, const void* bias, long long n, long long n_pad, int K, int N,
                       const float* res, const float* gamma, const float* beta, float eps, float* y, float* stats, void* y_bf,
                       const float* pos_table, const int* tok_pos, void* ypos_bf, void* f_out, int y_cached) {
  TgArgs A = {};
  A.f_out = (unsigned short*)f_out;
  A.y_cached = y_cached;
  A.X = (const unsigned short*)X; A.Wp = (const uint4*)Wp; A.bias = (const unsigned short*)bias; A.n = n; A.n_pad = n_pad;
  A.res = res; A.gamma = gamma; A.beta = beta; A.eps = eps; A.y = y; A.stats = stats; A.y_bf = (unsigned short*)y_bf;
  A.pos_table = pos_table; A.tok_pos = tok_pos; A.ypos_bf = (unsigned short*)ypos_bf;
  // product operand + fp32 residual in, fp32 rows + statistics out, + the optional bf16 copies (y, y + pos, rounded branch)
  GdTimed timed(GD_T_TOK_GEMM, st,
                2.0 * n_pad * K + (double)n * N * (4 + 4 + (y_bf ? 2 : 0) + (ypos_bf ? 2 : 0) + (f_out ? 2 : 0)) + 8.0 * n +
                    (ypos_bf ? 4.0 * n : 0.0) + 2.0 * K * N,
                2.0 * n_pad * K * N);
  return tg_dispatch<TG_RES_LN>(K, N, A, st);
}

// y = LayerNorm(res + bf16(gelu(bf16(X W1^T + b1)) W2^T + b2)) with every output of gd_tok_gemm_res_ln; h (n_pad, 2 d) bf16 is the
// only trace of the hidden activation in HBM (gd_tok_gemm_gelu_bwd re-creates gelu(h) for the weight gradient)
bool gd_tok_gemm_ffn_supported(int d, int ff) { return (d == 128 || d == 256) && ff == 2 * d; }
int gd_tok_gemm_ffn(hipStream_t st, const void* X, const void* W1p, const void* b1, const void* W2p, const void* b2, long long n,
                    long long n_pad, int d, void* h, const float* res, const float* gamma, const float* beta, float eps, float* y,
                    float* stats, void* y_bf, const float* pos_table, const int* tok_pos, void* ypos_bf, void* f_out) {
  TgFfnArgs F = {};
  F.X = (const unsigned short*)X; F.W1p = (const uint4*)W1p; F.b1 = (const unsigned short*)b1; F.W2p = (const uint4*)W2p;
  F.b2 = (const unsigned short*)b2; F.h = (unsigned short*)h;
  TgArgs& A = F.L;
  A.n = n; A.n_pad = n_pad; A.res = res; A.gamma = gamma; A.beta = beta; A.eps = eps; A.y = y; A.stats = stats;
  A.y_bf = (unsigned short*)y_bf; A.pos_table = pos_table; A.tok_pos = tok_pos; A.ypos_bf = (unsigned short*)ypos_bf;
  A.f_out = (unsigned short*)f_out;
  const int ff = 2 * d;
  // operand + residual in, h + LayerNorm rows (+ copies) out, both weight images
  GdTimed timed(GD_T_TOK_GEMM, st,
                2.0 * n_pad * d + 2.0 * n_pad * ff + (double)n * d * (4 + 4 + (y_bf ? 2 : 0) + (ypos_bf ? 2 : 0) + (f_out ? 2 : 0)) + 8.0 * n +
                    (ypos_bf ? 4.0 * n : 0.0) + 4.0 * d * ff,
                4.0 * n_pad * d * ff);
  if (d == 128) return tg_launch_ffn<128>(F, n_pad, st);
  if (d == 256) return tg_launch_ffn<256>(F, n_pad, st);
  GD_REQUIRE(false, "tok_gemm_ffn: d must be 128 or 256");
}

// dx (n, N) fp32 [+ dx_bf (n_pad, N) bf16] = LayerNorm backward of  g = dy + [dy2] + bf16(X Wp^T)  through LN(ln_a + ln_b) with the saved
// (mean, rstd) rows; part: (n_pad / gd_tok_gemm_rows(N), 3, N) fp32 partial rows of dgamma / dbeta / column sums of dx
int gd_tok_gemm_rows(int N) { return N >= 256 ? TgRows<256>::value : TgRows<128>::value; }
int gd_tok_gemm_ln_bwd(hipStream_t st, const void* X, const void* Wp, long long n, long long n_pad, int K, int N, const float* dy,
                       const void* dy2_bf, const float* ln_a, const void* ln_b_bf, const float* stats, const float* gamma, float* dx,
                       void* dx_bf, float* part) {
  TgArgs A = {};
  A.X = (const unsigned short*)X; A.Wp = (const uint4*)Wp; A.n = n; A.n_pad = n_pad;
  A.res = dy; A.aux = (const unsigned short*)dy2_bf; A.ln_a = ln_a; A.ln_b = (const unsigned short*)ln_b_bf;
  A.stats = const_cast<float*>(stats); A.gamma = gamma; A.y = dx; A.y_bf = (unsigned short*)dx_bf; A.part = part;
  // product operand, dy (fp32) [+ dy2 (bf16)], both LayerNorm addends (fp32 + bf16), statistics in; dx (fp32) [+ bf16 copy] and the
  // per-workgroup partial rows out
  GdTimed timed(GD_T_TOK_GEMM, st,
                2.0 * n_pad * K + (double)n * N * (4 + (dy2_bf ? 2 : 0) + 4 + 2 + 4 + (dx_bf ? 2 : 0)) + 8.0 * n +
                    12.0 * N * (double)(n_pad / gd_tok_gemm_rows(N)) + 2.0 * K * N,
                2.0 * n_pad * K * N);
  return tg_dispatch<TG_LN_BWD>(K, N, A, st);
}
extern "C" int gdmae_tok_gemm_qkv(const void* Xpos, const void* X, const void* Wp_qk, const void* Wp_v, const void* bias3, long long n_pad,
                                  int d, void* qk, void* v, void* stream) {
  GD_REQUIRE(n_pad > 0 && n_pad % TG_ROWS == 0, "tok_gemm: rows must be padded to a multiple of 64");
  GD_REQUIRE(d == 128 || d == 256, "tok_gemm_qkv: d must be 128 or 256");
  return gd_tok_gemm_qkv((hipStream_t)stream, Xpos, X, Wp_qk, Wp_v, bias3, n_pad, d, qk, v);
}
extern "C" int gdmae_tok_gemm_ffn(const void* X, const void* W1p, const void* b1, const void* W2p, const void* b2, long long n, long long n_pad,
                                  int d, void* h, const float* res, const float* gamma, const float* beta, float eps, float* y, float* stats,
                                  void* y_bf16, const float* pos_table, const int* tok_pos, void* ypos_bf16, void* f_out, void* stream) {
  GD_REQUIRE(n_pad > 0 && n_pad % TG_ROWS == 0 && n <= n_pad && n >= 1, "tok_gemm: rows must be padded to a multiple of 64");
  GD_REQUIRE(d == 128 || d == 256, "tok_gemm_ffn: d must be 128 or 256");
  return gd_tok_gemm_ffn((hipStream_t)stream, X, W1p, b1, W2p, b2, n, n_pad, d, h, res, gamma, beta, eps, y, stats, y_bf16, pos_table, tok_pos,
                         ypos_bf16, f_out);
}
extern "C" int gdmae_tok_gemm_ln_bwd_rows(int N) { return gd_tok_gemm_rows(N); }
extern "C" int gdmae_tok_gemm_ln_bwd(const void* X, const void* Wp, long long n, long long n_pad, int K, int N, const float* dy,
                                     const void* dy2_bf16, const float* ln_a, const void* ln_b_bf16, const float* stats,
                                     const float* gamma, float* dx, void* dx_bf16, float* part, void* stream) {
  GD_REQUIRE(n_pad > 0 && n_pad % TG_ROWS == 0 && n <= n_pad && n >= 1, "tok_gemm: rows must be padded to a multiple of 64");
  GD_REQUIRE(gd_tok_gemm_supported(K, N) && N <= 256, "tok_gemm_ln_bwd: unsupported (K, N)");
  return gd_tok_gemm_ln_bwd((hipStream_t)stream, X, Wp, n, n_pad, K, N, dy, dy2_bf16, ln_a, ln_b_bf16, stats, gamma, dx, dx_bf16, part);
}

// ------------------------------------------------------------------------------------------------
// C ABI (tests and stand-alone use): Y = epilogue(X Wp^T + bias), see the table at the top.  epilogue: 0 plain, 1 GELU
// (out0 = h, out1 = gelu(h)), 2 GELU backward (aux = h; out1 optional: gelu(h)), 3 residual + LayerNorm (out0 optional: the
// rounded product).
// ------------------------------------------------------------------------------------------------
extern "C" int gdmae_tok_gemm(const void* X, const void* Wp, const void* bias, long long n, long long n_pad, int K, int N,
                              int epilogue, void* out0, void* out1, const void* aux, const float* res, const float* gamma,
                              const float* beta, float eps, float* y, float* stats, void* y_bf16, const float* pos_table,
                              const int* tok_pos, void* ypos_bf16, void* stream) {
  GD_REQUIRE(n_pad > 0 && n_pad % TG_ROWS == 0 && n <= n_pad && n >= 1, "tok_gemm: rows must be padded to a multiple of 64");
  GD_REQUIRE(gd_tok_gemm_supported(K, N), "tok_gemm: unsupported (K, N)");
  hipStream_t st = (hipStream_t)stream;
  switch (epilogue) {
    case TG_PLAIN: return gd_tok_gemm_plain(st, X, Wp, bias, n_pad, K, N, out0);
    case TG_GELU: return gd_tok_gemm_gelu(st, X, Wp, bias, n_pad, K, N, out0, out1);
    case TG_GELU_BWD: return gd_tok_gemm_gelu_bwd(st, X, Wp, aux, n_pad, K, N, out0, out1);
    case TG_RES_LN:
      return gd_tok_gemm_res_ln(st, X, Wp, bias, n, n_pad, K, N, res, gamma, beta, eps, y, stats, y_bf16, pos_table, tok_pos, ypos_bf16,
                                out0, 0);
  }
  GD_REQUIRE(false, "tok_gemm: unknown epilogue");
}
