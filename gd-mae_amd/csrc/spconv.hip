// 2-D sparse convolution (k3: submanifold, strided s2 p1, and their input gradients) as an IMPLICIT GEMM over the rulebook
// (reference call sites pcdet/utils/spconv_utils.py:37-56, spt_backbone.py:206,217 - spconv's
// gather-GEMM-scatter there, un-vendored):
//
//     Y[r, :] = sum_{tap = 0..8}  W_tap (COUT, CIN)  X[nbr[r, tap], :]          (nbr < 0: the tap has no active input)
//
// Round 2 materialised the im2col matrix (n x 9 CIN bf16: 207 MB for the 256 -> 256 block of stage 2), ran a library GEMM
// over it and kept it for the weight gradient; the input gradient did the same over the transposed rulebook.  Here the
// gathered rows never leave the CU: a workgroup (8 wavefronts) owns 32, 64 or 128 output rows (SpRows) and the FULL output width; per tap
// it gathers its rows' CIN-vectors through the rulebook straight into an LDS tile (the next tap's rows are in flight in
// registers while the current tap is multiplied), and accumulates Y^T = W X^T with v_mfma_f32_32x32x16_bf16 exactly as
// the token GEMMs do (A = weights streamed in fragment order from a packed image - 9 per-tap images back to back, refreshed
// once per optimizer step -, B = the LDS tile).  The same kernel is the input gradient: X = dY rows, nbr = the transposed
// rulebook, W = the per-tap transposed weights.  fp32 source rows (the first convolution after an fp32 stage output) are
// rounded to bf16 on their way into LDS.
//
// Bytes per output row: 9 gathered rows x CIN x 2 B (L2 hits: every input row is read by up to 9 outputs) + COUT x 2 B
// written, against 9 CIN x 2 B written + read for the im2col matrix before.
#include "../../include/gdmae_hip.h"
#include "common.h"
#include "dw_grouped.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

union SpFrag {
  uint4 q;
  bf16x8 v;
};
__device__ inline unsigned short sp_f2bf(float f) { return gd_to_bf16(f); }
__device__ inline unsigned sp_pack2(float lo, float hi) { return gd_pack_bf16(lo, hi); }

struct SpArgs {
  const void* X;          // (n_src, CIN) bf16 or fp32 rows
  const int* nbr;         // (n, 9) source row of every (output row, tap), < 0: none
  const uint4* Wp;        // 9 packed (COUT, CIN) images
  unsigned short* Y;      // (n, COUT) bf16
  long long n;
  float* part;            // optional (gridDim.x, 2, COUT): per-workgroup column sums of Y and Y^2 (the bf16-rounded values) - the
                          // BatchNorm statistics of a conv block as this launch's epilogue instead of a pass over Y
  // optional rider: the split-K reduce of the block's weight gradient (conv_block.hip: dW[(o * 9 + k) * cin + i] += sum_s part[k][s][o][i],
  // slices in order) as the first `ride_blocks` workgroups of this launch - the input-gradient convolution follows the grouped
  // weight-gradient launch anyway and touches neither its partial tiles nor dW
  const float* ride_part;
  float* ride_dW;
  int ride_S, ride_cout, ride_cin;
  unsigned ride_blocks;
};

__device__ __forceinline__ void sp_ride_dw_reduce(const SpArgs& A) {
  const int S = A.ride_S, cout = A.ride_cout, cin = A.ride_cin;
  const long long per_tap = (long long)S * cout * cin, mn4 = (long long)cout * cin / 4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < 9 * mn4; e += (long long)A.ride_blocks * blockDim.x) {
    const int k = (int)(e / mn4);
    const long long r = e % mn4;
    const float4* p = reinterpret_cast<const float4*>(A.ride_part + k * per_tap) + r;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 8 <= S; s += 8) {                   // eight slices in flight, added in slice order (as k_spconv_dw_reduce)
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(long long)(s + j) * mn4];
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    for (; s < S; ++s) {
      const float4 v = p[(long long)s * mn4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const long long o = (r * 4) / cin, i = (r * 4) % cin;
    float4* d = reinterpret_cast<float4*>(A.ride_dW + (o * 9 + k) * cin + i);
    float4 w = *d;
    w.x += acc.x; w.y += acc.y; w.z += acc.z; w.w += acc.w;
    *d = w;
  }
}

constexpr int kWaves = 8;

// Rows per workgroup.  The weights are the dominant stream (every workgroup walks all 9 tap images: 1.2 MB at 256 x 256, from L2),
// so 64 rows instead of round 3's 32 at COUT = 256 halve it (k_spconv forward 244 -> 202 us per step); fp32 rows at 256 x 256 keep
// 32 (the unconverted rows of a 64-row tile do not fit 128 registers).
#ifndef SP_ROWS256
#define SP_ROWS256 64        // experiment switch
#endif
template <int CIN, int COUT, bool SRC_F32>
struct SpRows {
#ifndef SP_ROWS128
#define SP_ROWS128 128       // rows per workgroup at 128 -> 128 with bf16 rows (experiment switch; 64 = before)
#endif
  static constexpr int value = COUT >= 256 ? ((SRC_F32 && CIN >= 256) ? 32 : SP_ROWS256) : ((CIN == 128 && !SRC_F32) ? SP_ROWS128 : 64);
};

template <int CIN, int COUT, bool SRC_F32>
__global__ __launch_bounds__(512, 2) void k_spconv(SpArgs A) {
  constexpr int ROWS = SpRows<CIN, COUT, SRC_F32>::value;
  constexpr int KS = CIN / 16;                     // k-steps per tap
  constexpr int MB = COUT / 32;                    // 32-channel blocks
  constexpr int MPW = MB >= kWaves ? MB / kWaves : 1;
  constexpr int WPB = MB >= kWaves ? 1 : kWaves / MB;         // wavefronts that share a 32-channel block (they split the rows)
  constexpr int NPW = (ROWS / 32) / WPB;                      // 32-row blocks per wavefront (they share its weight fragments)
  static_assert(NPW >= 1 && NPW * WPB * 32 == ROWS, "row blocks");
  constexpr int XP = CIN * 2 + 16;                 // LDS row pitch of a gathered tile (conflict-free ds_read_b128)
  constexpr int SP = COUT * 2 + 16;                // ... of the output staging tile
  constexpr int CPR = CIN / 8;                     // 16-byte chunks per gathered row
  constexpr int RPP = 512 / CPR;                   // rows per pass of the 512 threads
  constexpr int P = ROWS / RPP;                    // passes per tap
  static_assert(P >= 1 && ROWS % RPP == 0, "unsupported shape");
  extern __shared__ __align__(16) unsigned char lds[];
  if (blockIdx.x < A.ride_blocks) return sp_ride_dw_reduce(A);      // uniform per workgroup
  const unsigned blk = blockIdx.x - A.ride_blocks;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long row0 = (long long)blk * ROWS;
  const int mb0 = wv / WPB;
  const int nb0 = (wv % WPB) * NPW;
  const uint4* __restrict__ wp = A.Wp + (size_t)mb0 * 64 + lane;        // + (gstep * MB + j * kWaves) * 64

  // ---- weight fragments: a ring of ONE WHOLE TAP (KS k-steps).  Step ks of tap t multiplies slot ks and reloads it with the
  // fragment of tap t + 1 right away.  The load counter retires in order, so waiting for a fragment also waits for every load
  // issued before it: with a short ring the gather of the next tap's rows (issued at the top of a tap) was drained by the first
  // fragment requested after it, a few k-steps later; now every fragment a tap consumes was requested BEFORE that tap's gather, and
  // the gather has the whole tap of MFMAs to land.
  // (CIN = 256 with 64-row tiles or fp32 rows: half a tap, both: a quarter - the whole one does not fit 128 registers next to the
  // gathered rows.)
  constexpr int RING = (KS == 16 && (ROWS == 64 || SRC_F32)) ? ((SRC_F32 && ROWS == 64) ? 4 : 8) : ((KS == 8 && ROWS == 128) ? 4 : KS);
  SpFrag wr[RING][MPW];
#pragma unroll
  for (int ks = 0; ks < RING; ++ks)
#pragma unroll
    for (int j = 0; j < MPW; ++j) wr[ks][j].q = wp[((size_t)ks * MB + j * kWaves) * 64];

  // ---- gather machinery: this thread's 16-byte chunk gc of rows grow[p].  Every load is unconditional (missing taps and rows past
  // the end read row 0 / the last rulebook row and are cleared by a mask when the tile is staged): a branch around a load makes the
  // compiler drain the load counter where the paths join.  The counter is in-order, so the first weight fragment needed after a
  // gather was issued waits for the gather as well (see the weight ring above).  Rows are fetched one tap ahead; fp32 rows are
  // converted when they are staged, not when they are loaded.
  constexpr int GD = 1;                            // gather distance in taps
  constexpr int RQ = SRC_F32 ? 2 : 1;              // 16-byte registers per gathered chunk
  const int gc = tid % CPR;
  long long nrow[P];                               // rulebook row (clamped)
  unsigned rvalid = 0u;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    const long long g = row0 + p * RPP + tid / CPR;
    rvalid |= g < A.n ? 1u << p : 0u;
    nrow[p] = (g < A.n ? g : A.n - 1) * 9;
  }
  int idx[P];                                      // sources of the next tap to fetch
  uint4 rq[1][P][RQ];
  unsigned keep[1] = {0u};                         // bit p: the chunk holds a real row
  auto load_idx = [&](int tap) {
#pragma unroll
    for (int p = 0; p < P; ++p) idx[p] = A.nbr[nrow[p] + tap];
  };
  auto fetch = [&](int slot) {
    unsigned k = 0u;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      asm volatile("" : "+v"(idx[p]));             // the index is looked at HERE, not where it was loaded (a compare hoisted to the
                                                   // load would wait for it a tap early)
      const bool ok = idx[p] >= 0 && ((rvalid >> p) & 1u);
      const long long j = ok ? idx[p] : 0;
      k |= ok ? 1u << p : 0u;
      if (SRC_F32) {
        const float* s = (const float*)A.X + j * CIN + gc * 8;
        rq[slot][p][0] = *reinterpret_cast<const uint4*>(s);
        rq[slot][p][RQ - 1] = *reinterpret_cast<const uint4*>(s + 4);
      } else {
        rq[slot][p][0] = *reinterpret_cast<const uint4*>((const unsigned short*)A.X + j * CIN + gc * 8);
      }
    }
    keep[slot] = k;
  };
  auto stage = [&](unsigned char* b, int slot) {
#pragma unroll
    for (int p = 0; p < P; ++p) {
      uint4 q;
      if (SRC_F32) {
        const uint4 a = rq[slot][p][0], c = rq[slot][p][RQ - 1];
        q.x = sp_pack2(__uint_as_float(a.x), __uint_as_float(a.y)); q.y = sp_pack2(__uint_as_float(a.z), __uint_as_float(a.w));
        q.z = sp_pack2(__uint_as_float(c.x), __uint_as_float(c.y)); q.w = sp_pack2(__uint_as_float(c.z), __uint_as_float(c.w));
      } else {
        q = rq[slot][p][0];
      }
      const unsigned m = ((keep[slot] >> p) & 1u) ? 0xFFFFFFFFu : 0u;
      q.x &= m; q.y &= m; q.z &= m; q.w &= m;
      *reinterpret_cast<uint4*>(b + (p * RPP + tid / CPR) * XP + gc * 16) = q;
    }
  };
  load_idx(0);
  fetch(0);
  load_idx(1);
  stage(lds, 0);
  __syncthreads();

  f32x16 acc[MPW][NPW];
#pragma unroll
  for (int j = 0; j < MPW; ++j)
#pragma unroll
    for (int b = 0; b < NPW; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][b][i] = 0.f;

#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {              // fully unrolled: every condition below is a compile-time constant
    if (tap + GD < 9) {
      fetch(0);                                    // in flight behind this tap's MFMAs
      if (tap + GD + 1 < 9) load_idx(tap + GD + 1);
    }
    __builtin_amdgcn_sched_barrier(0);             // ... and issued before them
    const unsigned char* lb = lds + (tap & 1) * (ROWS * XP) + ((nb0 * 32) + (lane & 31)) * XP + (lane >> 5) * 16;
    SpFrag sf[2][NPW];
#pragma unroll
    for (int b = 0; b < NPW; ++b) sf[0][b].q = *reinterpret_cast<const uint4*>(lb + b * 32 * XP);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
#pragma unroll
        for (int b = 0; b < NPW; ++b) sf[(ks + 1) & 1][b].q = *reinterpret_cast<const uint4*>(lb + b * 32 * XP + (ks + 1) * 32);
      }
#pragma unroll
      for (int j = 0; j < MPW; ++j)
#pragma unroll
        for (int b = 0; b < NPW; ++b) acc[j][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[ks % RING][j].v, sf[ks & 1][b].v, acc[j][b], 0, 0, 0);
      if (tap * KS + ks + RING < 9 * KS) {
#pragma unroll
        for (int j = 0; j < MPW; ++j) wr[ks % RING][j].q = wp[(((size_t)tap * KS + ks + RING) * MB + j * kWaves) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (tap < 8) stage(lds + ((tap + 1) & 1) * (ROWS * XP), 0);
    __syncthreads();
  }

  // ---- accumulators -> bf16 -> staging tile [row][channel] -> 16-byte row stores
#pragma unroll
  for (int j = 0; j < MPW; ++j) {
    const int cb = (mb0 + j * kWaves) * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int b = 0; b < NPW; ++b) {
      const int row = (nb0 + b) * 32 + (lane & 31);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint2 o;
        o.x = sp_pack2(acc[j][b][4 * q], acc[j][b][4 * q + 1]);
        o.y = sp_pack2(acc[j][b][4 * q + 2], acc[j][b][4 * q + 3]);
        *reinterpret_cast<uint2*>(lds + row * SP + (cb + 8 * q) * 2) = o;
      }
    }
  }
  __syncthreads();
  {
    constexpr int OCPR = COUT / 8;
    constexpr int ORPP = 512 / OCPR;
    const int c = tid % OCPR, r = tid / OCPR;
    float s1[8], s2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s1[k] = s2[k] = 0.f;
    const bool stats = A.part != nullptr;          // uniform
#pragma unroll
    for (int p = 0; p < (ROWS + ORPP - 1) / ORPP; ++p) {
      const int rl = p * ORPP + r;
      if (rl < ROWS && row0 + rl < A.n) {
        const uint4 q = *reinterpret_cast<const uint4*>(lds + rl * SP + c * 16);
        *reinterpret_cast<uint4*>(A.Y + (row0 + rl) * COUT + c * 8) = q;
        if (stats) {
          const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float lo = __uint_as_float(w[k] << 16), hi = __uint_as_float(w[k] & 0xFFFF0000u);
            s1[2 * k] += lo; s2[2 * k] = fmaf(lo, lo, s2[2 * k]);
            s1[2 * k + 1] += hi; s2[2 * k + 1] = fmaf(hi, hi, s2[2 * k + 1]);
          }
        }
      }
    }
    if (stats) {
      // the ORPP row groups of a channel chunk meet in LDS (fixed order: the partial row is bit-repeatable); ORPP x 2 x COUT floats fit
      // the tiles they replace
      static_assert(ORPP * 2 * COUT * 4 <= 2 * ROWS * XP || ORPP * 2 * COUT * 4 <= ROWS * SP, "statistics scratch fits the tiles");
      __syncthreads();                             // every thread is done with the staging tile
      float* red = reinterpret_cast<float*>(lds);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        red[(r * 2 + 0) * COUT + c * 8 + k] = s1[k];
        red[(r * 2 + 1) * COUT + c * 8 + k] = s2[k];
      }
      __syncthreads();
      for (int e = tid; e < 2 * COUT; e += 512) {
        float a = 0.f;
#pragma unroll
        for (int g = 0; g < ORPP; ++g) a += red[g * 2 * COUT + e];
        A.part[(long long)blk * 2 * COUT + e] = a;
      }
    }
  }
}

template <int CIN, int COUT, bool SRC_F32>
int sp_launch(const SpArgs& A, hipStream_t st) {
  constexpr int ROWS = SpRows<CIN, COUT, SRC_F32>::value;
  constexpr int tile = ROWS * (CIN * 2 + 16), stg = ROWS * (COUT * 2 + 16);
  constexpr int lds = 2 * tile > stg ? 2 * tile : stg;
  static bool once = false;
  if (!once) {
    GD_CHECK(hipFuncSetAttribute((const void*)k_spconv<CIN, COUT, SRC_F32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    once = true;
  }
  hipLaunchKernelGGL((k_spconv<CIN, COUT, SRC_F32>), dim3((unsigned)gd_div_up(A.n, ROWS) + A.ride_blocks), dim3(512), lds, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}

}  // namespace

bool gd_spconv_supported(int cin, int cout) { return (cin == 128 || cin == 256) && (cout == 128 || cout == 256); }
size_t gd_spconv_packed_bytes(int cin, int cout) { return (size_t)9 * cin * cout * 2; }

// Y (n, cout) bf16 = sum_tap Wp_tap X[nbr[:, tap]]; X bf16 or fp32 (x_f32) rows; Wp = 9 packed (cout, cin) images
// rows per workgroup of a launch = rows per statistics partial
int gd_spconv_rows(int cin, int cout, int x_f32) {
#define SP_CASE(ci, co) \
  if (cin == ci && cout == co) return x_f32 ? SpRows<ci, co, true>::value : SpRows<ci, co, false>::value;
  SP_CASE(128, 128);
  SP_CASE(128, 256);
  SP_CASE(256, 128);
  SP_CASE(256, 256);
#undef SP_CASE
  return 0;
}
// part (optional): (ceil(n / gd_spconv_rows), 2, cout) fp32 partial rows of the column sums of Y, Y^2
// ride_* (optional, ride_part != null): a conv block's weight-gradient reduce carried by this launch (SpArgs); refused (-2) when this
// launch does not happen (n <= 0) - the caller then reduces with a launch of its own
int gd_spconv(hipStream_t st, const void* X, int x_f32, const int* nbr, const void* Wp, long long n, int cin, int cout, void* Y, int slot,
              float* part, const float* ride_part, int ride_S, int ride_cout, int ride_cin, float* ride_dW) {
  if (n <= 0) return ride_part ? -2 : 0;
  GD_REQUIRE(gd_spconv_supported(cin, cout), "spconv: channels must be 128 or 256");
  SpArgs A{X, nbr, (const uint4*)Wp, (unsigned short*)Y, n, part, nullptr, nullptr, 0, 0, 0, 0u};
  if (ride_part) {
    A.ride_part = ride_part; A.ride_dW = ride_dW; A.ride_S = ride_S; A.ride_cout = ride_cout; A.ride_cin = ride_cin;
    A.ride_blocks = (unsigned)gd_div_up(9ll * ride_cout * ride_cin / 4, 512);
  }
  // gathered rows (L2) + the output rows + the weight image, per launch
  GdTimed timed(slot, st, (double)n * (9.0 * cin * (x_f32 ? 4 : 2) + 2.0 * cout + 36.0) + 18.0 * cin * cout, 2.0 * n * 9.0 * cin * cout);
#define SP_CASE(ci, co)                                                    \
  if (cin == ci && cout == co) return x_f32 ? sp_launch<ci, co, true>(A, st) : sp_launch<ci, co, false>(A, st);
  SP_CASE(128, 128);
  SP_CASE(128, 256);
  SP_CASE(256, 128);
  SP_CASE(256, 256);
#undef SP_CASE
  return -1;
}

// C ABI: see include/gdmae_hip.h
extern "C" size_t gdmae_spconv_packed_bytes(int cin, int cout) { return gd_spconv_packed_bytes(cin, cout); }

// pack-job table (gdmae_tok_gemm_pack format: {src, dst, M, K, ld, transpose} x 9) of a (cout, 3, 3, cin) fp32 weight:
// transposed = 0: the forward images (cout, cin) per tap; 1: the input-gradient images (cin, cout) per tap
extern "C" int gdmae_spconv_pack_jobs(const float* W, int cin, int cout, int transposed, void* packed, long long* jobs) {
  GD_REQUIRE(gd_spconv_supported(cin, cout), "spconv_pack_jobs: channels must be 128 or 256");
  for (int t = 0; t < 9; ++t) {
    long long* J = jobs + 6 * t;
    J[0] = (long long)(W + (size_t)t * cin);
    J[1] = (long long)((char*)packed + (size_t)t * cin * cout * 2);
    J[2] = transposed ? cin : cout;      // rows of the packed (M, K) matrix
    J[3] = transposed ? cout : cin;
    J[4] = 9ll * cin;                    // A[r][c] = transposed ? src[c * ld + r] : src[r * ld + c]
    J[5] = transposed;
  }
  return 0;
}

extern "C" int gdmae_spconv(const void* X, int x_f32, const int* nbr, const void* packed, long long n, int cin, int cout, void* Y,
                            int timing_slot, void* stream) {
  return gd_spconv((hipStream_t)stream, X, x_f32, nbr, packed, n, cin, cout, Y, timing_slot > 0 ? timing_slot : GD_T_SPCONV_FWD, nullptr, nullptr, 0, 0, 0, nullptr);
}
extern "C" int gdmae_spconv_stat_rows(int cin, int cout, int x_f32) { return gd_spconv_rows(cin, cout, x_f32); }
extern "C" int gdmae_spconv_stats(const void* X, int x_f32, const int* nbr, const void* packed, long long n, int cin, int cout, void* Y,
                                  float* part, void* stream) {
  GD_REQUIRE(part != nullptr, "spconv_stats: partial rows");
  return gd_spconv((hipStream_t)stream, X, x_f32, nbr, packed, n, cin, cout, Y, GD_T_SPCONV_FWD, part, nullptr, 0, 0, 0, nullptr);
}


// ------------------------------------------------------------------------------------------------
// Nine per-tap TN products over a rulebook as ONE grouped launch + a fixed-order reduce:
//     out[k][n][m_off + m] (+)= sum_t  G[t][m] * X[nbr[t][k]][n]         k = 0..8,  G (n_pad rows, M) bf16, X (.., N) bf16
// (the weight gradient of the decoder's 3x3 conv_out per source stage: G = the stage's BatchNorm/ReLU rows minus background, X =
// the tile-compact output gradient, out = dWk (9, C2, Cin) at the stage's column offset - spt_backbone_mae.py:46-52 backward).
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_tap_dw_reduce(const float* __restrict__ part, int S, int M, int N, float* __restrict__ out, int ld_out,
                                                       int m_off) {
  // part: [k][s][m][n] -> out[(k * N + n) * ld_out + m_off + m]
  const long long per_tap = (long long)S * M * N;
  const long long total = 9ll * N * M;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(e % N);                    // n fastest: the S partial reads of a thread group are coalesced
    const long long km = e / N;
    const int m = (int)(km % M), k = (int)(km / M);
    const float* p = part + k * per_tap + (long long)m * N + n;
    // eight slices requested before the first is added (same order of additions: bit-identical to the one-load-per-iteration loop,
    // which was S serial round trips per thread)
    const long long st = (long long)M * N;
    float acc = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(long long)(s + j) * st];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j];
    }
    for (; s < S; ++s) acc += p[(long long)s * st];
    out[((long long)k * N + n) * ld_out + m_off + m] += acc;
  }
}
int tap_dw_pick(long long n, int tiles, long long* n_pad) {
  static const int wgs = getenv("GDMAE_SPCONV_DW_WGS") ? atoi(getenv("GDMAE_SPCONV_DW_WGS")) : 640;
  return gd_dw_pick(n, tiles, wgs, n_pad);
}
}  // namespace

// rows the G operand must be allocated for (>= n: the row count padded to the slice grid the launch will use)
extern "C" long long gdmae_tap_dw_rows(long long n, int M, int N) {
  long long n_pad = 0;
  tap_dw_pick(n, 9 * (M / 128) * (N / 128), &n_pad);
  return n_pad;
}
extern "C" size_t gdmae_tap_dw_workspace_bytes(long long n, int M, int N) {
  long long n_pad = 0;
  return gd_align((size_t)tap_dw_pick(n, 9 * (M / 128) * (N / 128), &n_pad) * 9 * M * N * sizeof(float));
}
extern "C" int gdmae_tap_dw(const void* G, long long n, long long n_pad, int M, const void* X, const int* nbr, int N, float* out, int ld_out,
                            int m_off, void* workspace, void* stream) {
  GD_REQUIRE(M % 128 == 0 && N % 128 == 0 && n >= 0, "tap_dw: M, N multiples of 128");
  if (n == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  GdDwGroup Gp;
  Gp.n_jobs = 9;
  long long want = 0;
  const int S = tap_dw_pick(n, 9 * (M / 128) * (N / 128), &want);
  GD_REQUIRE(n_pad == want, "tap_dw: G must be allocated for gdmae_tap_dw_rows(n, M, N) rows");
  for (int k = 0; k < 9; ++k) Gp.job[k] = GdDwJob{G, X, M, N, (float*)workspace + (size_t)k * S * M * N, nullptr, 0, nbr + k, 9, 0};
  {
    GdTimed timed(GD_T_DEC_CONV_BWD, st, (double)n * (2.0 * M + 9.0 * 2.0 * N + 36.0) + 36.0 * S * M * N, 2.0 * n * 9.0 * M * N);
    int rc = gd_dw_grouped_s(st, Gp, n_pad, n, S);
    if (rc) return rc;
  }
  hipLaunchKernelGGL(k_tap_dw_reduce, dim3(gd_div_up(9ll * M * N, 256)), dim3(256), 0, st, (const float*)workspace, S, M, N, out, ld_out, m_off);
  GD_LAUNCH_CHECK();
  return 0;
}
