// Encoder layer in registers (reference pcdet/models/model_utils/sst_basic_block.py:57-84 EncoderLayer): the
// same three launches per layer and direction as layer_fused.hip, same tensors, same rounding points - but a token row never passes
// through LDS.  One WAVEFRONT owns 16 token rows from the first product of a launch to its last store:
//
//   * products are v_mfma_f32_16x16x32_bf16 with the weights as A operand (16 output channels x 32 k) and the token rows as B operand
//     (32 k x 16 rows): lane (n = lane % 16, g = lane / 16) then holds, for every 16-channel block of the output, channels 4 g .. 4 g + 3
//     of row n.  A whole row lives in the four lanes n, n + 16, n + 32, n + 48: LayerNorm statistics are a register sum and two
//     v_permlane swaps, bias / GELU / rounding are lane-local.
//   * the output of one product IS the B operand of the next: a k-step of the next product takes k-slots 8 g .. 8 g + 7 from lane
//     (n, g) - the lane's four channels of block 2 s and of block 2 s + 1, packed to bf16.  The k order this implies
//     (slot 8 g + j of step s = channel 32 s + 16 (j / 4) + 4 g + j % 4) is baked into the weight images (gdmae_tok_gemm_pack, "chained"
//     format); a sum over k does not care about its order.  No staging tile, no barrier, no layout conversion between products.
//   * weights reach the four wavefronts of a workgroup through a ring of three 16 KB LDS slots (16 fragments of 1 KB each, in
//     exactly the order the products consume them: the image of a launch is ONE linear stream).  Every thread fetches 4 x 16 bytes
//     of slot q + 3 into registers while slot q is multiplied and commits slot q + 1; one s_barrier per slot is the only
//     synchronisation of the kernel, and no load or store of token rows is ordered by it.
//   * the feed-forward block runs in hidden chunks of 128 channels: h chunk = x1 W1[chunk]^T -> bias, rounding, store, GELU ->
//     B operand of f += gelu(h chunk) W2[:, chunk]^T; the hidden row never exists as a whole.
//
// Against layer_fused.hip's row tiles (32 / 64 rows per 8-wavefront workgroup, three LDS staging round trips and twelve barriers per
// tile): per token the weights cross L2 -> CU once per 64 rows instead of once per 32, operands are read from LDS once per 16 rows (the
// 16 x 16 shape), and the row passes have no latency chain - every load and store of a wavefront is independent of its neighbours.
#include "tok_tiles.h"
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef V3_TRACE      // experiment build (tools/build_variant.sh ... -DV3_TRACE): per-phase time stamps of one workgroup, 100 MHz ticks
__device__ unsigned long long v3_trace_buf[8 * 64];
extern "C" int gdmae_debug_v3_trace(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(v3_trace_buf), sizeof(v3_trace_buf)) == hipSuccess ? 0 : 1;
}
#define V3_MARK(i) do { if (blockIdx.x == V3_TRACE && (threadIdx.x & 63) == 0) v3_trace_buf[(threadIdx.x >> 6) * 64 + (i)] = wall_clock64(); } while (0)
#else
#define V3_MARK(i) do { } while (0)
#endif

#ifndef V3_STAGGER
#define V3_STAGGER 127      // s_sleep units of 64 cycles
#endif

namespace {

#ifndef V3_NWAVES
#define V3_NWAVES 4         // wavefronts (16 rows each) per workgroup: 4 = two workgroups per CU, 8 = one (experiment switch)
#endif
#ifndef V3_NRING
#define V3_NRING 4          // LDS ring slots of 16 KB
#endif
constexpr int V3_WAVES = V3_NWAVES, V3_THREADS = 64 * V3_WAVES, V3_TROWS = 16, V3_ROWS = V3_WAVES * V3_TROWS;
constexpr int V3_DPW = 16 / V3_WAVES;       // DMA instructions (1 KB fragments) per wavefront and slot
constexpr int V3_SLOT_FR = 16, V3_SLOT_B = V3_SLOT_FR * 1024, V3_RING = V3_NRING;

template <int D>
struct V3S {
  static constexpr int FF = 2 * D, NB = D / 16, KS = D / 32;
  static constexpr int HC = 128, HB = HC / 16, HKS = HC / 32, NCH = FF / HC;       // hidden chunk: channels, blocks, k-steps
  static constexpr int PO_SL = KS * NB / V3_SLOT_FR, P1_SL = KS * HB / V3_SLOT_FR, P2_SL = HKS * NB / V3_SLOT_FR;
  static constexpr int SLOTS = PO_SL + NCH * (P1_SL + P2_SL);
  };

// ---- the weight stream ---------------------------------------------------------------------------------------------
// LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane straight into LDS, no staging registers): wavefront w moves fragments w, w + 4,
// w + 8, w + 12 of a slot, 1 KB per instruction.  Slot j of the stream lives in ring slot j % V3_RING.  Schedule of a step q:
//     W  s_waitcnt vmcnt((V3_RING - 2) * 4)   this wavefront's part of slot q has landed (only the DMA of slots q + 1 ... may be younger)
//     B  s_barrier                             every part has, and every wavefront is done with slot q - 1
//     I  DMA of slot q + V3_RING - 1           into the ring slot that q - 1 occupied
//     C  the products of slot q
// The load counter is in order and counts stores: a wait that lets N younger operations stay in flight must not have stores among them
// that it does not know of.  Row passes (everything between two products: they issue the kernel's loads and stores of token rows) therefore
// START with a full drain (v3_drain: the V3_RING - 1 slots in flight have had a product's time to land), after which the next
// V3_RING - 1 steps need no W at all, and the stores of the pass have those steps to drain before a W can see them.
// The DMA is issued through inline assembly: the builtin makes the compiler treat every later ds_read as a reader of the DMA's LDS
// bytes and drain the load counter (s_waitcnt vmcnt(0)) in front of it - i.e. wait for the slot that was requested a moment ago.  An
// operation the compiler does not know of only makes ITS counted waits stricter (the counter is in order), never wrong.
struct V3Stream {
  const unsigned char* src;   // stream + 1024 * wavefront + 16 * lane
  unsigned ring;              // LDS byte address of the ring + 1024 * wavefront (wave-uniform)
  int q, slot, last;          // q: stream index of the slot that is multiplied next, slot = q % V3_RING
  __device__ __forceinline__ void dma(int qq, int sl) {
    const int qc = qq < last ? qq : last;        // past the end: the last slot once more, into a ring slot nobody reads again
    const unsigned char* g = src + (size_t)qc * V3_SLOT_B;
    const unsigned l = __builtin_amdgcn_readfirstlane(ring + sl * V3_SLOT_B);
#pragma unroll
    for (int i = 0; i < V3_DPW; ++i)
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l + i * 1024 * V3_WAVES), "v"(g + i * 1024 * V3_WAVES) : "memory");
  }
  __device__ __forceinline__ void start(const uint4* stream, unsigned char* lds, int wv, int lane, int slots) {
    src = (const unsigned char*)stream + 1024 * wv + 16 * lane;
    ring = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds + 1024 * wv;
    q = 0; slot = 0; last = slots - 1;
#pragma unroll
    for (int j = 0; j < V3_RING - 1; ++j) dma(j, j);
  }
};
__device__ __forceinline__ void v3_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ f32x4 v3_mma(const uint4& a, const uint4& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One ring slot: fragments F0 .. F0 + 15 of a product whose image has MBN channel blocks per k-step (fragment f = k-step f / MBN,
// block f % MBN).  J: index of the step within its product (the first V3_RING - 1 steps behind a row pass do not wait, see above).
template <int J, int F0, int MBN, int NKS, int NAC>
__device__ __forceinline__ void v3_step(V3Stream& S, const unsigned char* ring, const uint4 (&bop)[NKS], f32x4 (&acc)[NAC], int lane) {
  if constexpr (J >= V3_RING - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((V3_RING - 2) * V3_DPW) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  S.dma(S.q + V3_RING - 1, S.slot == 0 ? V3_RING - 1 : S.slot - 1);
  const unsigned char* sl = ring + S.slot * V3_SLOT_B + lane * 16;
  // fragments in groups of four, two groups of LDS reads in flight under each group of products
  uint4 a[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) a[0][j] = *(const uint4*)(sl + j * 1024);
#pragma unroll
  for (int j = 0; j < 4; ++j) a[1][j] = *(const uint4*)(sl + (4 + j) * 1024);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    if (gq + 2 < 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a[(gq + 2) % 3][j] = *(const uint4*)(sl + ((gq + 2) * 4 + j) * 1024);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = F0 + gq * 4 + j;               // compile-time after unrolling
      acc[f % MBN] = v3_mma(a[gq % 3][j], bop[f / MBN], acc[f % MBN]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  S.q += 1;
  S.slot = S.slot == V3_RING - 1 ? 0 : S.slot + 1;
}

// compile-time loop: f(integral_constant<int, I>) for I = 0 .. N - 1
template <int I, int N, typename F>
__device__ __forceinline__ void v3_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    v3_for<I + 1, N>(f);
  }
}

template <int N>
__device__ __forceinline__ void v3_zero(f32x4 (&acc)[N]) {
#pragma unroll
  for (int b = 0; b < N; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// sum over the four lanes n, n + 16, n + 32, n + 48 that share a token row
__device__ __forceinline__ float v3_row_sum(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// LayerNorm of a row held as NB x 4 values per lane: s -> o, returns (mean, rstd).  gamma / beta: + 4 g already applied.
template <int D>
__device__ __forceinline__ float2 v3_ln(const float (&s)[D / 16][4], const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                        float (&o)[D / 16][4]) {
  constexpr int NB = D / 16;
  float t = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b) t += (s[b][0] + s[b][1]) + (s[b][2] + s[b][3]);
  const float mean = v3_row_sum(t) * (1.f / D);
  float sq = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dlt = s[b][i] - mean;
      sq = fmaf(dlt, dlt, sq);
    }
  const float rstd = rsqrtf(v3_row_sum(sq) * (1.f / D) + eps);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float4 g4 = *(const float4*)(gamma + 16 * b), b4 = *(const float4*)(beta + 16 * b);
    o[b][0] = (s[b][0] - mean) * rstd * g4.x + b4.x;
    o[b][1] = (s[b][1] - mean) * rstd * g4.y + b4.y;
    o[b][2] = (s[b][2] - mean) * rstd * g4.z + b4.z;
    o[b][3] = (s[b][3] - mean) * rstd * g4.w + b4.w;
  }
  return make_float2(mean, rstd);
}

__device__ __forceinline__ void v3_unpack4(const uint2& q, float (&f)[4]) { tl_unpack4(q, f); }

// ---- forward --------------------------------------------------------------------------------------------------------
struct V3Fwd {
  const unsigned short* o;       // (n_pad, D) attention output
  const unsigned short* x;       // (n_pad, D) layer input = residual of LayerNorm 1
  const uint4* W;                // the launch's weight stream: Wo (natural k) | per hidden chunk: W1[chunk] | W2[:, chunk] (chained k)
  const unsigned short *bo, *b1, *b2;
  const float *g1, *be1, *g2, *be2;
  float eps;
  long long n, n_pad;
  unsigned short *a, *x1, *h, *f;   // kept for the backward (layer_fused.hip LfArgs)
  float *st1, *st2;
  float* y;
  unsigned short* y_bf;
  unsigned short* ypos_bf;
  const float* pos_table;
  const int* tok_pos;
  const unsigned short* res0;
  unsigned short* res_out;
};

template <int D>
__global__ __launch_bounds__(V3_THREADS, 8 / V3_WAVES) void k_layer_fwd_v3(V3Fwd A) {
  using C = V3S<D>;
  constexpr int FF = C::FF, NB = C::NB, KS = C::KS, HB = C::HB, HKS = C::HKS;
  extern __shared__ __align__(16) unsigned char v3_ring[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int n = lane & 15, g = lane >> 4;
  const long long row = (long long)blockIdx.x * V3_ROWS + wv * V3_TROWS + n;
  const bool live = row < A.n;
  const long long rr = live ? row : A.n - 1;
  const int c0 = 4 * g;                               // the lane's channels of block b: 16 b + c0 .. + 3

  V3_MARK(0);
#ifndef V3_NO_STAGGER
  // The two workgroups of a CU start together and would run their row passes (VALU) and their products (matrix pipe, mostly waiting for
  // the weight stream) at the same moments; the one in the odd workgroup slot of its CU (HW_ID.TG_ID) starts half a hidden chunk later,
  // so that one multiplies while the other runs a row pass.
  if (__builtin_amdgcn_s_getreg(4 | (16 << 6) | (3 << 11)) & 1) __builtin_amdgcn_s_sleep(V3_STAGGER);
#endif
  V3Stream S;
  S.start(A.W, v3_ring, __builtin_amdgcn_readfirstlane(wv), lane, C::SLOTS);
  // the layer's vectors behind the ring: biases (bf16) | LayerNorm weights (fp32); the row passes read them with ds_read (every lane of
  // a 16-lane row group the same address) instead of 3 x D / 16 small global loads each that queue behind the pass's own row traffic
  unsigned char* const prm = v3_ring + V3_RING * V3_SLOT_B;
  const unsigned short* const l_bo = (const unsigned short*)prm;
  const unsigned short* const l_b1 = l_bo + D;
  const unsigned short* const l_b2 = l_b1 + FF;
  const float* const l_g1 = (const float*)(l_b2 + D);
  const float *const l_be1 = l_g1 + D, *const l_g2 = l_g1 + 2 * D, *const l_be2 = l_g1 + 3 * D;
  {
    unsigned* pw = (unsigned*)prm;
    for (int i = tid; i < D / 2; i += V3_THREADS) pw[i] = ((const unsigned*)A.bo)[i];
    for (int i = tid; i < FF / 2; i += V3_THREADS) pw[D / 2 + i] = ((const unsigned*)A.b1)[i];
    for (int i = tid; i < D / 2; i += V3_THREADS) pw[D / 2 + FF / 2 + i] = ((const unsigned*)A.b2)[i];
    float* fw = (float*)(pw + D + FF / 2);
    for (int i = tid; i < D; i += V3_THREADS) {
      fw[i] = A.g1[i]; fw[D + i] = A.be1[i]; fw[2 * D + i] = A.g2[i]; fw[3 * D + i] = A.be2[i];
    }
  }
  // optional operands select pointers once; the loads themselves are unconditional (layer_fused.hip)
  const bool has_pos = A.ypos_bf != nullptr, has_res = A.res_out != nullptr;
  const int* __restrict__ tokp = has_pos ? A.tok_pos : (const int*)A.x;
  const float* __restrict__ ptab = has_pos ? A.pos_table : A.g1;
  const unsigned short* __restrict__ res0p = has_res ? A.res0 : A.x;
  uint4 bo[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) bo[ks] = *(const uint4*)(A.o + row * D + 32 * ks + 8 * g);
  uint2 xr[NB];
#pragma unroll
  for (int b = 0; b < NB; ++b) xr[b] = *(const uint2*)(A.x + rr * D + 16 * b + c0);
  const int pos = tokp[rr];
  v3_drain();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the vectors are in LDS before the first barrier
  V3_MARK(1);

  // ---- a = o Wo^T + bo;  x1 = LN1(x + a)
  uint2 x1p[NB];                                      // LayerNorm-1 output, bf16: operand of linear1 and residual of LayerNorm 2
  {
    f32x4 acc[NB];
    v3_zero(acc);
    v3_for<0, C::PO_SL>([&](auto q) {
      constexpr int Q = decltype(q)::value;
      v3_step<Q, Q * V3_SLOT_FR, NB, KS, NB>(S, v3_ring, bo, acc, lane);
      V3_MARK(2 + Q);
    });
    v3_drain();
    V3_MARK(10);
    float s[NB][4], o[NB][4];
    uint2 aq[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint2 bq = *(const uint2*)(l_bo + 16 * b + c0);
      float bb[4], r4[4], v[4];
      v3_unpack4(bq, bb);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = acc[b][i] + bb[i];
      aq[b] = tl_pack4(v);
      v3_unpack4(aq[b], v);                           // the branch output as the backward will read it
      v3_unpack4(xr[b], r4);
#pragma unroll
      for (int i = 0; i < 4; ++i) s[b][i] = v[i] + r4[i];
    }
    const float2 st = v3_ln<D>(s, l_g1 + c0, l_be1 + c0, A.eps, o);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      uint2 q = tl_pack4(o[b]);
      if (!live) q = make_uint2(0u, 0u);              // pad rows: zero operand rows
      x1p[b] = q;
    }
    if (live) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        *(uint2*)(A.a + row * D + 16 * b + c0) = aq[b];
        *(uint2*)(A.x1 + row * D + 16 * b + c0) = x1p[b];
      }
      if (g == 0) *(float2*)(A.st1 + row * 2) = st;
    }
  }
  V3_MARK(11);
  uint4 bx[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) bx[ks] = make_uint4(x1p[2 * ks].x, x1p[2 * ks].y, x1p[2 * ks + 1].x, x1p[2 * ks + 1].y);

  // ---- f = gelu(x1 W1^T + b1) W2^T + b2, in hidden chunks of 128 channels
  f32x4 accf[NB];
  v3_zero(accf);
#pragma unroll 1
  for (int c = 0; c < C::NCH; ++c) {
    f32x4 acch[HB];
    v3_zero(acch);
    v3_for<0, C::P1_SL>([&](auto q) {
      constexpr int Q = decltype(q)::value;
      v3_step<Q, Q * V3_SLOT_FR, HB, KS, HB>(S, v3_ring, bx, acch, lane);
    });
    V3_MARK(12 + 4 * c);
    v3_drain();
    V3_MARK(13 + 4 * c);
    uint2 gq[HB];
#pragma unroll
    for (int b = 0; b < HB; ++b) {
      const uint2 bq = *(const uint2*)(l_b1 + c * C::HC + 16 * b + c0);
      float bb[4], v[4];
      v3_unpack4(bq, bb);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = acch[b][i] + bb[i];
      const uint2 hq = tl_pack4(v);
      *(uint2*)(A.h + row * FF + c * C::HC + 16 * b + c0) = hq;       // every row of the padded buffer (layer_fused.hip)
      v3_unpack4(hq, v);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = tg_gelu(v[i]);
      gq[b] = tl_pack4(v);
    }
    V3_MARK(14 + 4 * c);
    uint4 bg[HKS];
#pragma unroll
    for (int ks = 0; ks < HKS; ++ks) bg[ks] = make_uint4(gq[2 * ks].x, gq[2 * ks].y, gq[2 * ks + 1].x, gq[2 * ks + 1].y);
    v3_for<0, C::P2_SL>([&](auto q) {
      constexpr int Q = decltype(q)::value;
      v3_step<Q, Q * V3_SLOT_FR, NB, HKS, NB>(S, v3_ring, bg, accf, lane);
    });
    V3_MARK(15 + 4 * c);
  }

  // ---- y = LN2(x1 + f)
  {
    v3_drain();
    V3_MARK(30);
    // the position rows of the next layer's q / k operand, requested before anything is stored (the load counter is in order: a load
    // behind the pass's stores would wait for them)
    float4 p4q[NB];                                     // (each one takes over the registers of the accumulator block it follows)
    const float* __restrict__ prow = ptab + (long long)(has_pos ? pos : 0) * D + c0;
    float s[NB][4], o[NB][4];
    const long long e0 = row * D + c0;
    {
      uint2 fq[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const uint2 bq = *(const uint2*)(l_b2 + 16 * b + c0);
        float bb[4], r4[4], v[4];
        v3_unpack4(bq, bb);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = accf[b][i] + bb[i];
        fq[b] = tl_pack4(v);
        v3_unpack4(fq[b], v);
        v3_unpack4(x1p[b], r4);
#pragma unroll
        for (int i = 0; i < 4; ++i) s[b][i] = v[i] + r4[i];
        p4q[b] = *(const float4*)(prow + 16 * b);
      }
      if (live) {
#pragma unroll
        for (int b = 0; b < NB; ++b) *(uint2*)(A.f + e0 + 16 * b) = fq[b];
      }
    }
    const float2 st = v3_ln<D>(s, l_g2 + c0, l_be2 + c0, A.eps, o);
    if (live) {
      if (A.y) {
#pragma unroll
        for (int b = 0; b < NB; ++b) *(float4*)(A.y + e0 + 16 * b) = make_float4(o[b][0], o[b][1], o[b][2], o[b][3]);
      }
      if (has_res) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float r0[4];
          v3_unpack4(*(const uint2*)(res0p + e0 + 16 * b), r0);
          const float rs[4] = {r0[0] + o[b][0], r0[1] + o[b][1], r0[2] + o[b][2], r0[3] + o[b][3]};
          *(uint2*)(A.res_out + e0 + 16 * b) = tl_pack4(rs);
        }
      }
      if (A.y_bf) {
#pragma unroll
        for (int b = 0; b < NB; ++b) *(uint2*)(A.y_bf + e0 + 16 * b) = tl_pack4(o[b]);
      }
      if (has_pos) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const float4 p4 = p4q[b];
          const float op[4] = {o[b][0] + p4.x, o[b][1] + p4.y, o[b][2] + p4.z, o[b][3] + p4.w};
          *(uint2*)(A.ypos_bf + e0 + 16 * b) = tl_pack4(op);
        }
      }
      if (g == 0) *(float2*)(A.st2 + row * 2) = st;
    }
  }
  V3_MARK(31);
#ifdef V3_TRACE
  v3_drain();
  V3_MARK(32);
#endif
}

template <typename K>
int v3_set_lds(K kernel, int bytes) {
  GD_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// internal front end (encoder_layer.hip)
// ------------------------------------------------------------------------------------------------
int gd_layer_v3_rows() { return V3_ROWS; }
// elements of a launch's forward weight stream
size_t gd_layer_v3_fwd_stream_elems(int d, int ff) { return (size_t)d * d + 2 * (size_t)d * ff; }

// Pack jobs of the forward stream ({src, dst, M, K, ld, flags}, tok_gemm.hip k_tg_pack): Wo in the natural k order (its B operand
// comes from HBM), then per 128-channel hidden chunk W1[chunk rows] and W2[:, chunk columns] in the chained k order.
// Returns the number of jobs written (<= 1 + 2 ff / 128).
int gd_layer_v3_fwd_pack_jobs(const float* Wo, const float* W1, const float* W2, int d, int ff, void* stream_img, long long* jobs) {
  const long long V3 = 1ll << 20, CH = 1ll << 21;
  unsigned short* dst = (unsigned short*)stream_img;
  int nj = 0;
  auto put = [&](const float* src, long long M, long long K, long long ld, long long flags) {
    long long* J = jobs + 6 * nj++;
    J[0] = (long long)src; J[1] = (long long)dst; J[2] = M; J[3] = K; J[4] = ld; J[5] = flags;
    dst += M * K;
  };
  put(Wo, d, d, d, V3);
  for (int c = 0; c < ff / 128; ++c) {
    put(W1 + (size_t)128 * c * d, 128, d, d, V3 | CH);
    put(W2 + (size_t)128 * c, d, 128, ff, V3 | CH);
  }
  return nj;
}

int gd_layer_v3_fwd(hipStream_t st, int d, const void* o, const void* x, const void* Wstream, const void* bo, const void* b1, const void* b2,
                    const float* g1, const float* be1, const float* g2, const float* be2, float eps, long long n, long long n_pad, void* a,
                    void* x1, void* h, void* f, float* st1, float* st2, float* y, void* y_bf, void* ypos_bf, const float* pos_table,
                    const int* tok_pos, const void* res0, void* res_out) {
  V3Fwd A = {};
  A.o = (const unsigned short*)o; A.x = (const unsigned short*)x; A.W = (const uint4*)Wstream;
  A.bo = (const unsigned short*)bo; A.b1 = (const unsigned short*)b1; A.b2 = (const unsigned short*)b2;
  A.g1 = g1; A.be1 = be1; A.g2 = g2; A.be2 = be2; A.eps = eps; A.n = n; A.n_pad = n_pad;
  A.a = (unsigned short*)a; A.x1 = (unsigned short*)x1; A.h = (unsigned short*)h; A.f = (unsigned short*)f; A.st1 = st1; A.st2 = st2;
  A.y = y; A.y_bf = (unsigned short*)y_bf; A.ypos_bf = (unsigned short*)ypos_bf; A.pos_table = pos_table; A.tok_pos = tok_pos;
  A.res0 = (const unsigned short*)res0; A.res_out = (unsigned short*)res_out;
  GD_REQUIRE(n_pad % V3_ROWS == 0, "layer_v3_fwd: rows must be padded to 64");
  const int ff = 2 * d, lds = V3_RING * V3_SLOT_B + 2 * (2 * d + ff) + 16 * d;
  GdTimed timed(GD_T_TOK_GEMM, st,
                2.0 * n_pad * d + (double)n * d * (2 + 2 + 2 + 2 + ((y_bf || res_out) ? 2 : 0)) + 2.0 * n_pad * ff + 2.0 * (d * d + 2.0 * d * ff),
                2.0 * n_pad * (d * d + 2.0 * d * ff),
                (double)n * d * ((y ? 4 : 0) + (ypos_bf ? 2 : 0) + (res_out ? 2 : 0)) + 16.0 * n + (ypos_bf ? 4.0 * n : 0.0));
  static bool once[2] = {false, false};
  const dim3 grid((unsigned)(n_pad / V3_ROWS)), block(V3_THREADS);
  if (d == 128) {
    if (!once[0]) { if (int rc = v3_set_lds(k_layer_fwd_v3<128>, lds)) return rc; once[0] = true; }
    hipLaunchKernelGGL(k_layer_fwd_v3<128>, grid, block, lds, st, A);
  } else if (d == 256) {
    if (!once[1]) { if (int rc = v3_set_lds(k_layer_fwd_v3<256>, lds)) return rc; once[1] = true; }
    hipLaunchKernelGGL(k_layer_fwd_v3<256>, grid, block, lds, st, A);
  } else {
    GD_REQUIRE(false, "layer_v3_fwd: d must be 128 or 256");
  }
  GD_LAUNCH_CHECK();
  return 0;
}
