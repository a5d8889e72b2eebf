// Encoder layer around its bytes (reference pcdet/models/model_utils/sst_basic_block.py:57-84 EncoderLayer,
// cosine_msa.py:56-62 packed in-projection): everything of a post-norm layer that is not the windowed attention itself is THREE
// launches, and the residual stream between them is bf16 - a token row crosses HBM once per tensor the backward really needs.
//
//   forward   k_layer_fwd      a = o Wo^T + bo;  x1 = LN1(x + a);  h = x1 W1^T + b1;  f = gelu(h) W2^T + b2;  y = LN2(x1 + f)
//                              in : o, x (bf16)                         out: a, x1, h, f (bf16, kept for the backward), row statistics,
//                                                                            y (bf16) and y + pos (bf16) = the next layer's operands
//   backward  k_layer_bwd_ffn  dh = (df W2) * gelu'(h);  g1 = df + dh W1;  da = LN1'(g1);  do = da Wo
//                              in : df, h, x, a (bf16), statistics      out: dh, gelu(h) (operands of the weight gradients), da, do
//             k_layer_bwd_in   gx = da + [dqk | dv] Win;  df' = LN2'(gx) of the layer BELOW (or dx, fp32, at the bottom of a stage)
//                              in : dqk, dv, da, x1', f' (bf16)         out: df' (bf16) + its dgamma / dbeta / column-sum partial rows
//             k_ln2_bwd_top    df = LN2'(dy) for the top layer of a stage (dy arrives in fp32 from the sparse-conv block above)
//
// Against the launch-per-product sequence of tok_gemm.hip (out-projection + LN1, feed-forward block, GELU backward, two LayerNorm
// backward epilogues, two plain input-gradient products) a layer moves 60 d bytes per token instead of 106 d: the fp32 copies of
// the residual stream and of its gradient are gone (LayerNorm re-normalises every half layer, so the stream is O(1) and its bf16
// rounding is the rounding every GEMM operand already had), LN1's output lives in LDS between the out-projection and the
// feed-forward block, and the gradient of the residual branch is the SAME tensor as the gradient of the branch (df is read once
// as GEMM operand and once as addend from the tile in LDS).  Fused phases share one row tile per workgroup: 32 rows for d = 256,
// 64 for d = 128, 8 wavefronts, weights streamed as MFMA A operands from the fragment-ordered images of tok_gemm_pack.
#include "tok_tiles.h"
#include "layer_tail.h"

// experiment switches (tools/build_variant.sh): weight prefetch distance, row tile of the d = 256 stages, resident workgroups
#ifndef TL_ROWS256
#define TL_ROWS256 32
#endif
#ifndef TL_MINW
#define TL_MINW 4
#endif
// TL_HALVES = 2: one workgroup = two independent 8-wavefront row tiles that share barriers (and thereby the timing of their weight
// fragment loads: the second request for a fragment finds the line in the CU's L1); same wavefronts per CU as two workgroups
#ifndef TL_HALVES
#define TL_HALVES 1
#endif

#ifdef TL_TRACE      // experiment build (tools/build_variant.sh ... -DTL_TRACE=<workgroup>): per-phase time stamps of one workgroup of k_layer_fwd, 100 MHz ticks
__device__ unsigned long long tl_trace_buf[8 * 32];
extern "C" int gdmae_debug_tl_trace(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tl_trace_buf), sizeof(tl_trace_buf)) == hipSuccess ? 0 : 1;
}
#define TL_MARK(i) do { if (blockIdx.x == TL_TRACE && (threadIdx.x & 63) == 0) tl_trace_buf[(threadIdx.x >> 6) * 32 + (i)] = wall_clock64(); } while (0)
#else
#define TL_MARK(i) do { } while (0)
#endif

namespace {

template <int D>
struct TlRows {
  static constexpr int ROWS = D >= 256 ? TL_ROWS256 : 64;
  static constexpr int LPR = D / 4;                 // lanes per row in the LayerNorm passes (4 consecutive columns per lane)
  static constexpr int LRPP = TL_THREADS / LPR;     // rows per pass
  static constexpr int LPASS = ROWS / LRPP;
};

// LayerNorm forward of one row piece: s[4] -> o[4]; returns (mean, rstd)
template <int D>
__device__ __forceinline__ float2 tl_ln_fwd(const float (&s)[4], const float (&g)[4], const float (&bt)[4], float eps, float (&o)[4]) {
  constexpr int LPR = TlRows<D>::LPR;
  const float mean = tg_group_sum<LPR>((s[0] + s[1]) + (s[2] + s[3])) * (1.f / D);
  float sq = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float dlt = s[k] - mean;
    sq = fmaf(dlt, dlt, sq);
  }
  const float rstd = rsqrtf(tg_group_sum<LPR>(sq) * (1.f / D) + eps);
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = (s[k] - mean) * rstd * g[k] + bt[k];
  return make_float2(mean, rstd);
}

// LayerNorm backward over the rows of a tile: per-thread accumulators of dgamma / dbeta / column sums of dx, one partial row
// per workgroup (arithmetic and association order of k_add_ln_bwd, layernorm.hip)
template <int D>
struct TlLnBwd {
  float dg[4], db[4], dsx[4];
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int k = 0; k < 4; ++k) dg[k] = db[k] = dsx[k] = 0.f;
  }
  // d: gradient of the LayerNorm output (zeroed here when the row is padding); xa + xb: the LayerNorm input
  __device__ __forceinline__ void row(float (&d)[4], bool live, const uint2& xa, const uint2& xb, const float2& st, const float (&g)[4],
                                      float (&o)[4]) {
    constexpr int LPR = TlRows<D>::LPR;
    float sa[4], sb[4];
    tl_unpack4(xa, sa);
    tl_unpack4(xb, sb);
    const float mean = st.x, rstd = st.y;
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
    m1 = tg_group_sum<LPR>(m1) * (1.f / D);
    m2 = tg_group_sum<LPR>(m2) * (1.f / D);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[k] = live ? rstd * (gy[k] - m1 - xh[k] * m2) : 0.f;
      dsx[k] += o[k];
    }
  }
  // red: LDS, LRPP x 3 x D floats, free for use by every thread once the caller's barrier has passed; contains two barriers
  __device__ __forceinline__ void finish(float* red, float* part_row, int tid) {
    constexpr int LPR = TlRows<D>::LPR, LRPP = TlRows<D>::LRPP;
    const int c0 = 4 * (tid % LPR), r = tid / LPR;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[(r * 3 + 0) * D + c0 + k] = dg[k];
      red[(r * 3 + 1) * D + c0 + k] = db[k];
      red[(r * 3 + 2) * D + c0 + k] = dsx[k];
    }
    __syncthreads();
    for (int c = tid; c < 3 * D; c += TL_THREADS) {
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < LRPP; ++q) a += red[q * 3 * D + c];
      part_row[c] = a;
    }
  }
};

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
struct LfArgs {
  const unsigned short* o;       // (n_pad, D) attention output
  const unsigned short* x;       // (n_pad, D) layer input = residual of LayerNorm 1
  const uint4 *Wo, *W1, *W2;     // packed (D, D), (FF, D), (D, FF)
  const unsigned short *bo, *b1, *b2;
  const float *g1, *be1, *g2, *be2;
  float eps;
  long long n, n_pad;
  unsigned short *a, *x1, *h, *f;   // kept for the backward: branch outputs (the LayerNorm addends), LN1 output, pre-activation
  float *st1, *st2;                 // (n, 2) mean, rstd
  float* y;                         // optional (n, D) fp32: output of the stage's last layer
  unsigned short* y_bf;             // optional (n_pad, D): the next layer's v / residual operand
  unsigned short* ypos_bf;          // optional (n_pad, D): the next layer's q / k operand, bf16(y + pos_table[tok_pos[row]])
  const float* pos_table;
  const int* tok_pos;
  const unsigned short* res0;       // optional (n_pad, D): input rows of the STAGE; res_out = bf16(res0 + y), the block residual
  unsigned short* res_out;          // optional (n, D)
  // QKV = true: the in-projection of the NEXT layer on the rows this tile has just produced (q | k = (y + pos) Wqk^T + b, v = y Wv^T + b)
  const uint4 *Wqk_n, *Wv_n;        // packed (2 D, D), (D, D) of the next layer
  const unsigned short* bin_n;      // (3 D)
  unsigned short *qk_n, *v_n;       // (n_pad, 2 D), (n_pad, D)
};

template <int D, bool QKV>
__global__ __launch_bounds__(TL_THREADS * TL_HALVES, TL_MINW) void k_layer_fwd(LfArgs A) {
  constexpr int FF = 2 * D;
  using R = TlRows<D>;
  constexpr int ROWS = R::ROWS, LPR = R::LPR, LRPP = R::LRPP, LPASS = R::LPASS;
  constexpr int XP = D * 2 + 16, HP = FF * 2 + 16;
  extern __shared__ __align__(16) unsigned char lds_all[];
  const int tile_in_wg = threadIdx.x / TL_THREADS;   // TL_HALVES row tiles per workgroup, TL_THREADS threads each
  unsigned char* const lds = lds_all + tile_in_wg * (ROWS * XP + ROWS * HP);
  unsigned char* const xl = lds;                     // o tile -> a (staging) -> x1 tile -> f (staging)
  unsigned char* const hl = lds + ROWS * XP;         // h (staging) -> gelu(h) tile
  const int tid = threadIdx.x % TL_THREADS, lane = tid & 63, wv = tid >> 6;
  const long long tile_id = (long long)blockIdx.x * TL_HALVES + tile_in_wg;
  const long long row0 = tile_id * ROWS;
  const int lc0 = 4 * (tid % LPR), lr = tid / LPR;

  // Optional outputs select POINTERS once (to something readable when the output is absent); the loads themselves are unconditional:
  // a uniform branch around a load is a join with a load in flight, and the compiler drains the load counter there - one serial
  // round trip per optional load (this kernel had twelve of them per tile: four token positions at the top, four block-residual
  // rows and four position-table rows at the bottom).
  const bool has_pos = A.ypos_bf != nullptr, has_res = A.res_out != nullptr;
  const int* __restrict__ tokp = has_pos ? A.tok_pos : (const int*)A.x;
  const float* __restrict__ ptab = has_pos ? A.pos_table : A.g1;
  const unsigned short* __restrict__ res0p = has_res ? A.res0 : A.x;
  TL_MARK(0);
  TlProd<D, D, ROWS> pa;
  pa.prefetch(A.Wo, nullptr, wv, lane);
  uint2 res_pf[LPASS];
  int pos_pf[LPASS];
#pragma unroll
  for (int p = 0; p < LPASS; ++p) {
    const long long row = row0 + p * LRPP + lr;
    const long long rr = row < A.n ? row : A.n - 1;
    res_pf[p] = *(const uint2*)(A.x + rr * D + lc0);
    pos_pf[p] = tokp[rr];
  }
  tl_load_tile<D, ROWS>(A.o, row0, xl, XP, 0, tid);
  // LayerNorm weights of the row pass behind the product: requested here (a load behind the barrier that opens the pass is a whole
  // L2 round trip in front of its arithmetic)
  const float4 g4_1 = *(const float4*)(A.g1 + lc0), b4_1 = *(const float4*)(A.be1 + lc0);
  __syncthreads();
  TL_MARK(1);
  // ---- a = o Wo^T + bo
  {
    f32x16 acc[TlShape<D, D, ROWS>::MPW][TlShape<D, D, ROWS>::NPW];
    tl_zero(acc);
    TlBias<D, D, ROWS> bia;
    bia.load(A.bo, wv, lane);                        // lands behind the product
    pa.run(xl, XP, wv, lane, acc);
    TL_MARK(2);
    __syncthreads();                                 // every wavefront is done with the o tile
    tl_stage<D, D, ROWS>(acc, bia, xl, XP, wv, lane);
  }
  TlProd<D, FF, ROWS> pb;
  pb.prefetch(A.W1, nullptr, wv, lane);
  __syncthreads();
  TL_MARK(3);
  // ---- x1 = LN1(x + a): a leaves for HBM, x1 replaces it in LDS (operand tile of linear1) and is the residual of LN2
  uint2 x1_keep[LPASS];
  {
    const float4 g4 = g4_1, b4 = b4_1;
    const float g[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
    for (int p = 0; p < LPASS; ++p) {
      const int rl = p * LRPP + lr;
      const long long row = row0 + rl;
      const bool live = row < A.n;
      const uint2 fq = *(const uint2*)(xl + rl * XP + lc0 * 2);
      float s[4], r4[4], o[4];
      tl_unpack4(fq, s);
      tl_unpack4(res_pf[p], r4);
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] += r4[k];
      const float2 st = tl_ln_fwd<D>(s, g, bt, A.eps, o);
      uint2 q = tl_pack4(o);
      if (!live) q = make_uint2(0u, 0u);             // pad rows: zero operand rows (garbage attention rows stay out of the tile)
      *(uint2*)(xl + rl * XP + lc0 * 2) = q;
      x1_keep[p] = q;
      if (live) {
        TG_ST_U2(A.a + row * D + lc0, fq);
        TG_ST_U2(A.x1 + row * D + lc0, q);
        if (lc0 == 0) *(float2*)(A.st1 + row * 2) = st;
      }
    }
  }
  __syncthreads();
  TL_MARK(4);
  // ---- h = x1 W1^T + b1
  {
    f32x16 acc[TlShape<D, FF, ROWS>::MPW][TlShape<D, FF, ROWS>::NPW];
    tl_zero(acc);
    pb.run(xl, XP, wv, lane, acc);
    TL_MARK(5);
    tl_stage<D, FF, ROWS>(acc, A.b1, hl, HP, wv, lane);     // (two channel blocks per wavefront: no registers for an early bias)
  }
  TlProd<FF, D, ROWS> pc;
  pc.prefetch(A.W2, nullptr, wv, lane);
  __syncthreads();
  TL_MARK(6);
  // ---- h leaves for HBM (the backward differentiates the GELU at it), gelu(h) replaces it in LDS
  {
    constexpr int CPR = FF / 8, RPP = TL_THREADS / CPR;
    const int c = tid % CPR, r = tid / CPR;
#pragma unroll
    for (int p = 0; p < ROWS / RPP; ++p) {
      const int rl = p * RPP + r;
      const uint4 q = *(const uint4*)(hl + rl * HP + c * 16);
      TG_ST_U4(A.h + (row0 + rl) * FF + c * 8, q);
      float v[8];
      tg_unpack8(q, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tg_gelu(v[j]);
      *(uint4*)(hl + rl * HP + c * 16) = tg_pack8(v);
    }
  }
  __syncthreads();
  TL_MARK(7);
  // ---- f = gelu(h) W2^T + b2  (staged over the x1 tile: nobody reads it any more)
  float4 g4_2, b4_2;                                   // LayerNorm-2 weights, requested in front of the product
  {
    f32x16 acc[TlShape<FF, D, ROWS>::MPW][TlShape<FF, D, ROWS>::NPW];
    tl_zero(acc);
    TlBias<FF, D, ROWS> bic;
    bic.load(A.b2, wv, lane);
    g4_2 = *(const float4*)(A.g2 + lc0);
    b4_2 = *(const float4*)(A.be2 + lc0);
    pc.run(hl, HP, wv, lane, acc);
    TL_MARK(8);
    tl_stage<FF, D, ROWS>(acc, bic, xl, XP, wv, lane);
  }
  // QKV: y + pos replaces f in its slot of the x tile (operand of the q | k product), y goes to a tile behind the hidden tile's first
  // ROWS * HP bytes (operand of the v product, then its staging tile); the q | k rows are staged over the front of the buffer
  unsigned char* const yl = lds + ROWS * HP;
  __syncthreads();
  TL_MARK(9);
  // ---- y = LN2(x1 + f)
  {
    const float4 g4 = g4_2, b4 = b4_2;
    const float g[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
    uint2 r0q[LPASS];
    float4 p4q[LPASS];
#pragma unroll
    for (int p = 0; p < LPASS; ++p) {                // the optional operands of every pass, requested together
      const long long row = row0 + p * LRPP + lr;
      const long long rr = row < A.n ? row : A.n - 1;
      asm volatile("" : "+v"(pos_pf[p]));            // looked at here, not where it was loaded
      r0q[p] = *(const uint2*)(res0p + rr * D + lc0);
      p4q[p] = *(const float4*)(ptab + (long long)(has_pos ? pos_pf[p] : 0) * D + lc0);
    }
#pragma unroll
    for (int p = 0; p < LPASS; ++p) {
      const int rl = p * LRPP + lr;
      const long long row = row0 + rl;
      const bool live = row < A.n;
      const uint2 fq = *(const uint2*)(xl + rl * XP + lc0 * 2);
      float s[4], r4[4], o[4];
      tl_unpack4(fq, s);
      tl_unpack4(x1_keep[p], r4);
#pragma unroll
      for (int k = 0; k < 4; ++k) s[k] += r4[k];
      const float2 st = tl_ln_fwd<D>(s, g, bt, A.eps, o);
      if constexpr (QKV) {                             // operand tiles of the next layer's in-projection (pad rows: zero)
        const float op[4] = {o[0] + p4q[p].x, o[1] + p4q[p].y, o[2] + p4q[p].z, o[3] + p4q[p].w};
        *(uint2*)(xl + rl * XP + lc0 * 2) = live ? tl_pack4(op) : make_uint2(0u, 0u);
        *(uint2*)(yl + rl * XP + lc0 * 2) = live ? tl_pack4(o) : make_uint2(0u, 0u);
      }
      if (!live) continue;
      const long long e = row * D + lc0;
      TG_ST_U2(A.f + e, fq);
      if (A.y) TG_ST_F4(A.y + e, o[0], o[1], o[2], o[3]);
      if (has_res) {
        float r0[4];
        tl_unpack4(r0q[p], r0);
        const float rs[4] = {r0[0] + o[0], r0[1] + o[1], r0[2] + o[2], r0[3] + o[3]};
        *(uint2*)(A.res_out + e) = tl_pack4(rs);
      }
      if (A.y_bf) *(uint2*)(A.y_bf + e) = tl_pack4(o);
      if (has_pos) {
        const float op[4] = {o[0] + p4q[p].x, o[1] + p4q[p].y, o[2] + p4q[p].z, o[3] + p4q[p].w};
        *(uint2*)(A.ypos_bf + e) = tl_pack4(op);
      }
      if (lc0 == 0) *(float2*)(A.st2 + row * 2) = st;
    }
  }
  TL_MARK(10);
#ifdef TL_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TL_MARK(11);
#endif
  if constexpr (QKV) {
    // ---- the next layer's in-projection: same fragments, k order and rounding points as its own launch (k_tok_gemm_multi)
    TlProd<D, D, ROWS> pv;
    pv.prefetch(A.Wv_n, nullptr, wv, lane);
    __syncthreads();                                   // both operand tiles are complete
    f32x16 av[TlShape<D, D, ROWS>::MPW][TlShape<D, D, ROWS>::NPW];
    tl_zero(av);
    pv.run(yl, XP, wv, lane, av);
    TlProd<D, 2 * D, ROWS> pq;
    pq.prefetch(A.Wqk_n, nullptr, wv, lane);
    __syncthreads();                                   // every wavefront is done with the y tile: v is staged over it
    tl_stage<D, D, ROWS>(av, A.bin_n + 2 * D, yl, XP, wv, lane);
    f32x16 aq[TlShape<D, 2 * D, ROWS>::MPW][TlShape<D, 2 * D, ROWS>::NPW];
    tl_zero(aq);
    pq.run(xl, XP, wv, lane, aq);
    __syncthreads();                                   // ... and with the y + pos tile; the v tile is complete
    tl_stage<D, 2 * D, ROWS>(aq, A.bin_n, lds, HP, wv, lane);
    {
      constexpr int CPR = D / 8, RPP = TL_THREADS / CPR;
      const int c = tid % CPR, r = tid / CPR;
#pragma unroll
      for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
        const int rl = p * RPP + r;
        if (RPP > ROWS && rl >= ROWS) break;
        *(uint4*)(A.v_n + (row0 + rl) * D + c * 8) = *(const uint4*)(yl + rl * XP + c * 16);
      }
    }
    __syncthreads();
    {
      constexpr int CPR = 2 * D / 8, RPP = TL_THREADS / CPR;
      const int c = tid % CPR, r = tid / CPR;
#pragma unroll
      for (int p = 0; p < ROWS / RPP; ++p) {
        const int rl = p * RPP + r;
        *(uint4*)(A.qk_n + (row0 + rl) * (2 * D) + c * 8) = *(const uint4*)(lds + rl * HP + c * 16);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, feed-forward block + LayerNorm 1 + out-projection
// ------------------------------------------------------------------------------------------------
struct LbArgs {
  const unsigned short* df;      // (n_pad, D) gradient of the LayerNorm-2 input: of f AND of the residual x1
  const unsigned short* h;       // (n_pad, FF)
  const unsigned short *x, *a;   // LayerNorm-1 addends
  const float* st1;
  const float* g1;
  const uint4 *W2t, *W1t, *Wot;  // packed (FF, D), (D, FF), (D, D): transposes of W2, W1, Wo
  long long n, n_pad;
  unsigned short *dh, *gact;     // (n_pad, FF): operands of the weight gradients of linear1 / linear2
  unsigned short *da, *d_o;      // (n_pad, D): gradient of the LayerNorm-1 input (of a and of the residual x), and of the attention output
  float* part;                   // (n_pad / ROWS, 3, D) partial rows of dgamma1 / dbeta1 / column sums of da
};

template <int D>
__global__ __launch_bounds__(TL_THREADS * TL_HALVES, TL_MINW) void k_layer_bwd_ffn(LbArgs A) {
  constexpr int FF = 2 * D;
  using R = TlRows<D>;
  constexpr int ROWS = R::ROWS, LPR = R::LPR, LRPP = R::LRPP, LPASS = R::LPASS;
  constexpr int XP = D * 2 + 16, HP = FF * 2 + 16;
  static_assert(LRPP * 3 * D * 4 <= ROWS * HP, "partial-row reduction fits the hidden tile");
  extern __shared__ __align__(16) unsigned char lds_all[];
  const int tile_in_wg = threadIdx.x / TL_THREADS;
  unsigned char* const lds = lds_all + tile_in_wg * (ROWS * XP + ROWS * HP);
  unsigned char* const gl = lds;                     // df tile -> dx1 (staging) -> da tile -> do (staging)
  unsigned char* const hl = lds + ROWS * XP;         // dg (staging) -> dh tile -> reduction scratch
  const int tid = threadIdx.x % TL_THREADS, lane = tid & 63, wv = tid >> 6;
  const long long tile_id = (long long)blockIdx.x * TL_HALVES + tile_in_wg;
  const long long row0 = tile_id * ROWS;
  const int lc0 = 4 * (tid % LPR), lr = tid / LPR;

  TlProd<D, FF, ROWS> pa;
  pa.prefetch(A.W2t, nullptr, wv, lane);
  // operands of the row passes that do not depend on the products: requested now
  constexpr int HCPR = FF / 8, HRPP = TL_THREADS / HCPR, HPASS = ROWS / HRPP;
  const int hc = tid % HCPR, hr = tid / HCPR;
  uint4 h_pf[HPASS];
#pragma unroll
  for (int p = 0; p < HPASS; ++p) h_pf[p] = tg_ld_u4_once(A.h + (row0 + p * HRPP + hr) * FF + hc * 8);
  tl_load_tile<D, ROWS>(A.df, row0, gl, XP, 0, tid);
  __syncthreads();
  // ---- dg = df W2
  {
    f32x16 acc[TlShape<D, FF, ROWS>::MPW][TlShape<D, FF, ROWS>::NPW];
    tl_zero(acc);
    pa.run(gl, XP, wv, lane, acc);
    tl_stage<D, FF, ROWS>(acc, nullptr, hl, HP, wv, lane);
  }
  TlProd<FF, D, ROWS> pb;
  pb.prefetch(A.W1t, nullptr, wv, lane);
  // operands of the LayerNorm pass: in flight behind the GELU pass and the second product (requested here, not at the top,
  // to stay inside 128 registers during the first product)
  uint2 dy_pf[LPASS], xa_pf[LPASS], xb_pf[LPASS];
  float2 st_pf[LPASS];
#pragma unroll
  for (int p = 0; p < LPASS; ++p) {
    const long long row = row0 + p * LRPP + lr;
    const long long rr = row < A.n ? row : A.n - 1;
    dy_pf[p] = *(const uint2*)(A.df + rr * D + lc0);
    xa_pf[p] = tg_ld_u2_once(A.x + rr * D + lc0);
    xb_pf[p] = tg_ld_u2_once(A.a + rr * D + lc0);
    st_pf[p] = *(const float2*)(A.st1 + rr * 2);
  }
  __syncthreads();
  // ---- dh = dg * gelu'(h) (operand tile of the next product, and of linear1's weight gradient); gelu(h) for linear2's
#pragma unroll
  for (int p = 0; p < HPASS; ++p) {
    const int rl = p * HRPP + hr;
    const long long e = (row0 + rl) * FF + hc * 8;
    const uint4 q = *(const uint4*)(hl + rl * HP + hc * 16);
    float g[8], v[8];
    tg_unpack8(q, g);
    tg_unpack8(h_pf[p], v);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] = g[j] * tg_gelu_grad(v[j]);
    const uint4 dq = tg_pack8(g);
    *(uint4*)(hl + rl * HP + hc * 16) = dq;
    *(uint4*)(A.dh + e) = dq;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = tg_gelu(v[j]);
    *(uint4*)(A.gact + e) = tg_pack8(v);
  }
  __syncthreads();
  const float4 g4_1 = *(const float4*)(A.g1 + lc0);   // LayerNorm weights of the pass behind the product (see k_layer_fwd)
  // ---- dx1 = dh W1 (staged over the df tile: its rows are in registers)
  {
    f32x16 acc[TlShape<FF, D, ROWS>::MPW][TlShape<FF, D, ROWS>::NPW];
    tl_zero(acc);
    pb.run(hl, HP, wv, lane, acc);
    tl_stage<FF, D, ROWS>(acc, nullptr, gl, XP, wv, lane);
  }
  TlProd<D, D, ROWS> pc;
  pc.prefetch(A.Wot, nullptr, wv, lane);
  __syncthreads();                                   // staging complete; every wavefront is done with the dh tile
  // ---- da = LN1'(df + dx1)
  {
    const float4 g4 = g4_1;
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    TlLnBwd<D> L;
    L.init();
#pragma unroll
    for (int p = 0; p < LPASS; ++p) {
      const int rl = p * LRPP + lr;
      const long long row = row0 + rl;
      const bool live = row < A.n;
      const uint2 fq = *(const uint2*)(gl + rl * XP + lc0 * 2);
      float d[4], d2[4], o[4];
      tl_unpack4(dy_pf[p], d);
      tl_unpack4(fq, d2);
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] += d2[k];
      L.row(d, live, xa_pf[p], xb_pf[p], st_pf[p], g, o);
      const uint2 q = tl_pack4(o);
      *(uint2*)(gl + rl * XP + lc0 * 2) = q;
      if (live) *(uint2*)(A.da + row * D + lc0) = q;
    }
    L.finish(reinterpret_cast<float*>(hl), A.part + tile_id * 3 * D, tid);   // its barrier also publishes the da tile
  }
  // ---- do = da Wo
  {
    f32x16 acc[TlShape<D, D, ROWS>::MPW][TlShape<D, D, ROWS>::NPW];
    tl_zero(acc);
    pc.run(gl, XP, wv, lane, acc);
    __syncthreads();                                 // every wavefront is done with the da tile
    tl_stage<D, D, ROWS>(acc, nullptr, gl, XP, wv, lane);
  }
  __syncthreads();
  {
    constexpr int CPR = D / 8, RPP = TL_THREADS / CPR;
    const int c = tid % CPR, r = tid / CPR;
#pragma unroll
    for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
      const int rl = p * RPP + r;
      if (RPP > ROWS && rl >= ROWS) break;
      *(uint4*)(A.d_o + (row0 + rl) * D + c * 8) = *(const uint4*)(gl + rl * XP + c * 16);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward, in-projection + the LayerNorm-2 backward of the layer below (or the fp32 input gradient of the stage)
// ------------------------------------------------------------------------------------------------
struct LiArgs {
  const unsigned short *dqk, *dv;   // (n_pad, 2 D), (n_pad, D)
  const uint4 *Wqkt, *Wvt;          // packed (D, 2 D), (D, D): transposes of Win[:2d], Win[2d:]
  const unsigned short* dres;       // (n_pad, D) gradient through the residual branch (da of this layer)
  long long n, n_pad;
  // LN = true: LayerNorm 2 of the layer below
  const unsigned short *ln_a, *ln_b;
  const float* st;
  const float* gamma;
  unsigned short* dout;             // (n_pad, D) gradient of that LayerNorm's input
  float* part;
  // LN = false: gradient of the stage input, fp32 (dx) or - block-residual mode - bf16 with the gradient of the skip path added
  float* dx;                        // (n, D) fp32
  const unsigned short* dtop;       // (n, D) bf16 gradient of the block residual
  unsigned short* dx_bf;            // (n, D) bf16 = dtop + dres + product
};

// Tail blocks (layer_tail.h): the first Tm.total workgroups of the launch do the layer's closing reductions, one 256-thread tail block each
// (the other wavefronts of such a workgroup leave at once: s_barrier waits for the surviving waves only); they start with the launch and
// the row tiles fill in behind them.  Tm.start[y] = first workgroup of section y.
struct TailMap {
  unsigned start[12];
  unsigned total;
  int ny;
};
template <int D, bool LN>
__global__ __launch_bounds__(TL_THREADS * TL_HALVES, TL_MINW) void k_layer_bwd_in(LiArgs A, TailJobs T, TailMap Tm) {
  if (blockIdx.x < Tm.total) {          // uniform per workgroup
    if (threadIdx.x >= 256) return;
    int y = 0;
    while (y + 1 < Tm.ny && blockIdx.x >= Tm.start[y + 1]) ++y;
    tail_block(T, blockIdx.x - Tm.start[y], y, (int)threadIdx.x);
    return;
  }
  const unsigned blk = blockIdx.x - Tm.total;
  using R = TlRows<D>;
  constexpr int ROWS = R::ROWS, LPR = R::LPR, LRPP = R::LRPP, LPASS = R::LPASS;
  constexpr int KD = 3 * D, XP = KD * 2 + 16, SP = D * 2 + 16;
  static_assert(LRPP * 3 * D * 4 + ROWS * SP <= ROWS * XP, "staging tile + reduction scratch fit the operand tile");
  extern __shared__ __align__(16) unsigned char lds_all[];
  const int tile_in_wg = threadIdx.x / TL_THREADS;
  unsigned char* const lds = lds_all + tile_in_wg * (ROWS * XP);
  const int tid = threadIdx.x % TL_THREADS, lane = tid & 63, wv = tid >> 6;
  const long long tile_id = (long long)blk * TL_HALVES + tile_in_wg;
  const long long row0 = tile_id * ROWS;
  const int lc0 = 4 * (tid % LPR), lr = tid / LPR;

  TlProd<KD, D, ROWS, 2 * D / 16> pa;
  pa.prefetch(A.Wqkt, A.Wvt, wv, lane);
  uint2 dy_pf[LPASS], xa_pf[LN ? LPASS : 1], xb_pf[LN ? LPASS : 1];
  float2 st_pf[LN ? LPASS : 1];
#pragma unroll
  for (int p = 0; p < LPASS; ++p) {
    const long long row = row0 + p * LRPP + lr;
    const long long rr = row < A.n ? row : A.n - 1;
    dy_pf[p] = *(const uint2*)(A.dres + rr * D + lc0);
    if (LN) {
      xa_pf[p] = tg_ld_u2_once(A.ln_a + rr * D + lc0);
      xb_pf[p] = tg_ld_u2_once(A.ln_b + rr * D + lc0);
      st_pf[p] = *(const float2*)(A.st + rr * 2);
    }
  }
  tl_load_tile<2 * D, ROWS>(A.dqk, row0, lds, XP, 0, tid);
  tl_load_tile<D, ROWS>(A.dv, row0, lds, XP, 4 * D, tid);
  const float4 g4_ln = *(const float4*)((LN ? A.gamma : (const float*)A.dres) + lc0);
  __syncthreads();
  {
    f32x16 acc[TlShape<KD, D, ROWS>::MPW][TlShape<KD, D, ROWS>::NPW];
    tl_zero(acc);
    pa.run(lds, XP, wv, lane, acc);
    __syncthreads();
    tl_stage<KD, D, ROWS>(acc, nullptr, lds, SP, wv, lane);
  }
  __syncthreads();
  if (LN) {
    const float4 g4 = g4_ln;
    const float g[4] = {g4.x, g4.y, g4.z, g4.w};
    TlLnBwd<D> L;
    L.init();
#pragma unroll
    for (int p = 0; p < LPASS; ++p) {
      const int rl = p * LRPP + lr;
      const long long row = row0 + rl;
      const bool live = row < A.n;
      const uint2 fq = *(const uint2*)(lds + rl * SP + lc0 * 2);
      float d[4], d2[4], o[4];
      tl_unpack4(dy_pf[p], d);
      tl_unpack4(fq, d2);
#pragma unroll
      for (int k = 0; k < 4; ++k) d[k] += d2[k];
      L.row(d, live, xa_pf[p], xb_pf[p], st_pf[p], g, o);
      if (live) *(uint2*)(A.dout + row * D + lc0) = tl_pack4(o);
    }
    L.finish(reinterpret_cast<float*>(lds + ROWS * SP), A.part + tile_id * 3 * D, tid);
  } else {
#pragma unroll
    for (int p = 0; p < LPASS; ++p) {
      const int rl = p * LRPP + lr;
      const long long row = row0 + rl;
      if (row >= A.n) continue;
      const uint2 fq = *(const uint2*)(lds + rl * SP + lc0 * 2);
      float d[4], d2[4];
      tl_unpack4(dy_pf[p], d);
      tl_unpack4(fq, d2);
      if (A.dx_bf) {
        float d3[4];
        tl_unpack4(*(const uint2*)(A.dtop + row * D + lc0), d3);
        const float o[4] = {d3[0] + (d[0] + d2[0]), d3[1] + (d[1] + d2[1]), d3[2] + (d[2] + d2[2]), d3[3] + (d[3] + d2[3])};
        *(uint2*)(A.dx_bf + row * D + lc0) = tl_pack4(o);
      } else {
        TG_ST_F4(A.dx + row * D + lc0, d[0] + d2[0], d[1] + d2[1], d[2] + d2[2], d[3] + d2[3]);
      }
    }
  }
}

// df = LN2'(dy) for the top layer of a stage: dy (n, D) fp32, LayerNorm input = x1 + f (bf16), same row tiles / partial rows
struct LtArgs {
  const float* dy;                  // (n, D) fp32, or
  const unsigned short* dy_bf;      // (n, D) bf16 when dy is null
  const unsigned short *ln_a, *ln_b;
  const float* st;
  const float* gamma;
  long long n;
  unsigned short* dout;
  float* part;
};
template <int D>
__global__ __launch_bounds__(TL_THREADS) void k_ln2_bwd_top(LtArgs A) {
  using R = TlRows<D>;
  constexpr int ROWS = R::ROWS, LPR = R::LPR, LRPP = R::LRPP, LPASS = R::LPASS;
  __shared__ float red[LRPP * 3 * D];
  const int tid = threadIdx.x;
  const long long row0 = (long long)blockIdx.x * ROWS;
  const int lc0 = 4 * (tid % LPR), lr = tid / LPR;
  const float4 g4 = *(const float4*)(A.gamma + lc0);
  const float g[4] = {g4.x, g4.y, g4.z, g4.w};
  TlLnBwd<D> L;
  L.init();
#pragma unroll
  for (int p = 0; p < LPASS; ++p) {
    const long long row = row0 + p * LRPP + lr;
    const bool live = row < A.n;
    const long long rr = live ? row : A.n - 1;
    float d[4], o[4];
    if (A.dy) {
      const float4 d4 = *(const float4*)(A.dy + rr * D + lc0);
      d[0] = d4.x; d[1] = d4.y; d[2] = d4.z; d[3] = d4.w;
    } else {
      tl_unpack4(*(const uint2*)(A.dy_bf + rr * D + lc0), d);
    }
    L.row(d, live, tg_ld_u2_once(A.ln_a + rr * D + lc0), tg_ld_u2_once(A.ln_b + rr * D + lc0), *(const float2*)(A.st + rr * 2), g, o);
    if (live) *(uint2*)(A.dout + row * D + lc0) = tl_pack4(o);
  }
  L.finish(red, A.part + (long long)blockIdx.x * 3 * D, tid);
}

// operands of the first layer of a stage whose input rows are bf16: xb = x (padded buffer), xpb = bf16(x + pos_table[tok_pos[row]])
__global__ __launch_bounds__(256) void k_prep_tokens_bf(const unsigned short* __restrict__ x, const float* __restrict__ pos_table,
                                                        const int* __restrict__ tok_pos, long long n, int d, unsigned short* __restrict__ xb,
                                                        unsigned short* __restrict__ xpb) {
  const int cpr = d >> 3;
  const long long total = n * cpr;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / cpr;
    const int c = (int)(i - row * cpr) << 3;
    const uint4 q = *(const uint4*)(x + row * d + c);
    *(uint4*)(xb + row * d + c) = q;
    float v[8];
    tg_unpack8(q, v);
    const float* pr = pos_table + (long long)tok_pos[row] * d + c;
    const float4 p0 = *(const float4*)pr, p1 = *(const float4*)(pr + 4);
    v[0] += p0.x; v[1] += p0.y; v[2] += p0.z; v[3] += p0.w; v[4] += p1.x; v[5] += p1.y; v[6] += p1.z; v[7] += p1.w;
    *(uint4*)(xpb + row * d + c) = tg_pack8(v);
  }
}

template <typename K>
int set_lds(K kernel, int bytes) {
  GD_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// internal front end (encoder_layer.hip)
// ------------------------------------------------------------------------------------------------
bool gd_layer_fused_supported(int d, int ff) { return (d == 128 || d == 256) && ff == 2 * d; }
int gd_layer_fused_rows(int d) { return d >= 256 ? TL_ROWS256 : 64; }

int gd_layer_fused_fwd(hipStream_t st, int d, const void* o, const void* x, const void* Wo, const void* W1, const void* W2, const void* bo,
                       const void* b1, const void* b2, const float* g1, const float* be1, const float* g2, const float* be2, float eps,
                       long long n, long long n_pad, void* a, void* x1, void* h, void* f, float* st1, float* st2, float* y, void* y_bf,
                       void* ypos_bf, const float* pos_table, const int* tok_pos, const void* res0, void* res_out, const void* Wqk_n,
                       const void* Wv_n, const void* bin_n, void* qk_n, void* v_n) {
  LfArgs A = {};
  const bool qkv = Wqk_n != nullptr;
  GD_REQUIRE(!qkv || (Wv_n && bin_n && qk_n && v_n && y_bf && ypos_bf), "layer_fused_fwd: the fused in-projection needs the next layer's operands");
  A.Wqk_n = (const uint4*)Wqk_n; A.Wv_n = (const uint4*)Wv_n; A.bin_n = (const unsigned short*)bin_n;
  A.qk_n = (unsigned short*)qk_n; A.v_n = (unsigned short*)v_n;
  A.res0 = (const unsigned short*)res0; A.res_out = (unsigned short*)res_out;
  A.o = (const unsigned short*)o; A.x = (const unsigned short*)x;
  A.Wo = (const uint4*)Wo; A.W1 = (const uint4*)W1; A.W2 = (const uint4*)W2;
  A.bo = (const unsigned short*)bo; A.b1 = (const unsigned short*)b1; A.b2 = (const unsigned short*)b2;
  A.g1 = g1; A.be1 = be1; A.g2 = g2; A.be2 = be2; A.eps = eps; A.n = n; A.n_pad = n_pad;
  A.a = (unsigned short*)a; A.x1 = (unsigned short*)x1; A.h = (unsigned short*)h; A.f = (unsigned short*)f; A.st1 = st1; A.st2 = st2;
  A.y = y; A.y_bf = (unsigned short*)y_bf; A.ypos_bf = (unsigned short*)ypos_bf; A.pos_table = pos_table; A.tok_pos = tok_pos;
  const int rows = gd_layer_fused_rows(d), ff = 2 * d;
  const int lds = TL_HALVES * (rows * (d * 2 + 16) + rows * (ff * 2 + 16));
  // bf16 rows: o, x in; a, x1, h, f, y (or x + y) out; three weight images.  Side: fp32 y at the stage boundary, the y + pos copy,
  // the stage input re-read for the block residual, statistics rows, position ids
  GdTimed timed(GD_T_TOK_GEMM, st,
                2.0 * n_pad * d + (double)n * d * (2 + 2 + 2 + 2 + ((y_bf || res_out) ? 2 : 0)) + 2.0 * n_pad * ff + 2.0 * (d * d + 2.0 * d * ff) +
                    (qkv ? 2.0 * n_pad * 3 * d + 2.0 * 3 * d * d : 0.0),
                2.0 * n_pad * (d * d + 2.0 * d * ff) + (qkv ? 2.0 * n_pad * d * 3 * d : 0.0),
                (double)n * d * ((y ? 4 : 0) + (ypos_bf ? 2 : 0) + (res_out ? 2 : 0)) + 16.0 * n + (ypos_bf ? 4.0 * n : 0.0));
  static bool once[4] = {false, false, false, false};
  const dim3 grid((unsigned)(n_pad / (rows * TL_HALVES))), block(TL_THREADS * TL_HALVES);
#define LF_CASE(D_, Q_, idx)                                                                              \
  {                                                                                                       \
    if (!once[idx]) { if (int rc = set_lds(k_layer_fwd<D_, Q_>, lds)) return rc; once[idx] = true; }      \
    hipLaunchKernelGGL((k_layer_fwd<D_, Q_>), grid, block, lds, st, A);                                   \
  }
  if (d == 128 && qkv) LF_CASE(128, true, 0)
  else if (d == 128) LF_CASE(128, false, 1)
  else if (d == 256 && qkv) LF_CASE(256, true, 2)
  else if (d == 256) LF_CASE(256, false, 3)
  else GD_REQUIRE(false, "layer_fused_fwd: d must be 128 or 256");
#undef LF_CASE
  GD_LAUNCH_CHECK();
  return 0;
}

int gd_layer_fused_bwd_ffn(hipStream_t st, int d, const void* df, const void* h, const void* x, const void* a, const float* st1, const float* g1,
                           const void* W2t, const void* W1t, const void* Wot, long long n, long long n_pad, void* dh, void* gact, void* da,
                           void* d_o, float* part) {
  LbArgs A = {};
  A.df = (const unsigned short*)df; A.h = (const unsigned short*)h; A.x = (const unsigned short*)x; A.a = (const unsigned short*)a;
  A.st1 = st1; A.g1 = g1; A.W2t = (const uint4*)W2t; A.W1t = (const uint4*)W1t; A.Wot = (const uint4*)Wot; A.n = n; A.n_pad = n_pad;
  A.dh = (unsigned short*)dh; A.gact = (unsigned short*)gact; A.da = (unsigned short*)da; A.d_o = (unsigned short*)d_o; A.part = part;
  const int rows = gd_layer_fused_rows(d), ff = 2 * d;
  const int lds = TL_HALVES * (rows * (d * 2 + 16) + rows * (ff * 2 + 16));
  // bf16 rows: df, h, x, a in; dh, gelu(h), da, do out; three weight images.  Side: statistics rows, partial rows
  GdTimed timed(GD_T_TOK_GEMM, st,
                2.0 * n_pad * d + 2.0 * n_pad * ff + (double)n * d * (2 + 2 + 2) + 2.0 * n_pad * d + 4.0 * n_pad * ff + 2.0 * (d * d + 2.0 * d * ff),
                2.0 * n_pad * (d * d + 2.0 * d * ff), 8.0 * n + 12.0 * d * (double)(n_pad / rows));
  static bool once[2] = {false, false};
  if (d == 128) {
    if (!once[0]) { if (int rc = set_lds(k_layer_bwd_ffn<128>, lds)) return rc; once[0] = true; }
    hipLaunchKernelGGL(k_layer_bwd_ffn<128>, dim3((unsigned)(n_pad / (rows * TL_HALVES))), dim3(TL_THREADS * TL_HALVES), lds, st, A);
  } else if (d == 256) {
    if (!once[1]) { if (int rc = set_lds(k_layer_bwd_ffn<256>, lds)) return rc; once[1] = true; }
    hipLaunchKernelGGL(k_layer_bwd_ffn<256>, dim3((unsigned)(n_pad / (rows * TL_HALVES))), dim3(TL_THREADS * TL_HALVES), lds, st, A);
  } else {
    GD_REQUIRE(false, "layer_fused_bwd_ffn: d must be 128 or 256");
  }
  GD_LAUNCH_CHECK();
  return 0;
}

// LN: dout (n_pad, d) bf16 + part; otherwise dx (n, d) fp32
int gd_layer_fused_bwd_in(hipStream_t st, int d, const void* dqk, const void* dv, const void* Wqkt, const void* Wvt, const void* dres, long long n,
                          long long n_pad, const void* ln_a, const void* ln_b, const float* stats, const float* gamma, void* dout, float* part,
                          float* dx, const void* dtop, void* dx_bf, const TailJobs* tail) {
  LiArgs A = {};
  TailJobs T = {};
  TailMap Tm = {};
  double tail_bytes = 0.0;
  if (tail) {       // the layer's closing reductions as the first workgroups of this launch, sections packed back to back
    T = *tail;
    Tm.ny = T.J.count + 2;
    GD_REQUIRE(Tm.ny <= 11, "layer tail: sections");
    unsigned at = 0;
    for (int y = 0; y < Tm.ny; ++y) {
      Tm.start[y] = at;
      if (y < T.J.count) {
        at += (unsigned)((T.J.P4[y] + 255) / 256);
        tail_bytes += 16.0 * T.J.P4[y] * (T.J.S[y] + 2);
      } else if (y == T.J.count) {
        int cols = 0;
        for (int q = 0; q < T.a.count; ++q) {
          cols += T.a.len[q];
          tail_bytes += 4.0 * T.a.len[q] * (T.a.nblk[q] + 2);
        }
        at += (unsigned)((cols + 15) / 16);
      } else {
        at += 1;
        tail_bytes += 4.0 * T.n_part;
      }
    }
    Tm.start[Tm.ny] = at;
    Tm.total = at;
  }
  A.dtop = (const unsigned short*)dtop; A.dx_bf = (unsigned short*)dx_bf;
  A.dqk = (const unsigned short*)dqk; A.dv = (const unsigned short*)dv; A.Wqkt = (const uint4*)Wqkt; A.Wvt = (const uint4*)Wvt;
  A.dres = (const unsigned short*)dres; A.n = n; A.n_pad = n_pad; A.ln_a = (const unsigned short*)ln_a; A.ln_b = (const unsigned short*)ln_b;
  A.st = stats; A.gamma = gamma; A.dout = (unsigned short*)dout; A.part = part; A.dx = dx;
  const bool ln = dx == nullptr && dx_bf == nullptr;
  const int rows = gd_layer_fused_rows(d);
  const int lds = TL_HALVES * rows * (3 * d * 2 + 16);
  // bf16 rows: dqk, dv, da in (+ both LayerNorm addends in, df out | the skip-path gradient in, dx out); two weight images.  Side:
  // statistics + partial rows, or the fp32 dx rows
  GdTimed timed(GD_T_TOK_GEMM, st, 2.0 * n_pad * 3 * d + (double)n * d * (2 + (ln ? 2 + 2 + 2 : (dx_bf ? 2 + 2 : 0))) + 2.0 * 3 * d * d,
                2.0 * n_pad * 3 * d * d, (ln ? 8.0 * n + 12.0 * d * (double)(n_pad / rows) : (dx_bf ? 0.0 : 4.0 * n * d)) + tail_bytes);
  static bool once[4] = {false, false, false, false};
#define LI_CASE(D_, LN_, idx)                                                                                   \
  {                                                                                                             \
    if (!once[idx]) { if (int rc = set_lds(k_layer_bwd_in<D_, LN_>, lds)) return rc; once[idx] = true; }        \
    hipLaunchKernelGGL((k_layer_bwd_in<D_, LN_>), dim3((unsigned)(n_pad / (rows * TL_HALVES)) + Tm.total), dim3(TL_THREADS * TL_HALVES), lds, st, A, T, Tm);       \
  }
  if (d == 128 && ln) LI_CASE(128, true, 0)
  else if (d == 128) LI_CASE(128, false, 1)
  else if (d == 256 && ln) LI_CASE(256, true, 2)
  else if (d == 256) LI_CASE(256, false, 3)
  else GD_REQUIRE(false, "layer_fused_bwd_in: d must be 128 or 256");
#undef LI_CASE
  GD_LAUNCH_CHECK();
  return 0;
}

int gd_layer_fused_prep_bf(hipStream_t st, const void* x, const float* pos_table, const int* tok_pos, long long n, int d, void* xb, void* xpb) {
  if (n <= 0) return 0;
  long long g = (n * (d / 8) + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_prep_tokens_bf, dim3((unsigned)g), dim3(256), 0, st, (const unsigned short*)x, pos_table, tok_pos, n, d, (unsigned short*)xb,
                     (unsigned short*)xpb);
  GD_LAUNCH_CHECK();
  return 0;
}

// dy: (n, d) fp32, or null and dy_bf: (n, d) bf16
int gd_layer_fused_ln2_top(hipStream_t st, int d, const float* dy, const void* dy_bf, const void* ln_a, const void* ln_b, const float* stats,
                           const float* gamma, long long n, long long n_pad, void* dout, float* part) {
  LtArgs A = {dy, (const unsigned short*)dy_bf, (const unsigned short*)ln_a, (const unsigned short*)ln_b, stats, gamma, n, (unsigned short*)dout, part};
  const int rows = gd_layer_fused_rows(d);
  GdTimed timed(GD_T_TOK_GEMM, st, (double)n * d * ((dy ? 0 : 2) + 2 + 2 + 2), 0.0, (dy ? 4.0 * n * d : 0.0) + 8.0 * n + 12.0 * d * (double)(n_pad / rows));
  if (d == 128) hipLaunchKernelGGL(k_ln2_bwd_top<128>, dim3((unsigned)(n_pad / rows)), dim3(TL_THREADS), 0, st, A);
  else if (d == 256) hipLaunchKernelGGL(k_ln2_bwd_top<256>, dim3((unsigned)(n_pad / rows)), dim3(TL_THREADS), 0, st, A);
  else GD_REQUIRE(false, "layer_fused_ln2_top: d must be 128 or 256");
  GD_LAUNCH_CHECK();
  return 0;
}
