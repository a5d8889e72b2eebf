// Dense 3x3 convolution of channels-last bf16 maps on the bf16 matrix cores - forward, input gradient and weight gradient
// (the fine-tune detector: SSTBEVBackbone's four Conv2d(128, 128, 3, dilation 1 | 2) - reference
// pcdet/models/backbones_2d/sst_bev_backbone.py:6-42 -, CenterHead's shared / per-head convolutions - center_head.py:11-45 -
// and the dense decoder's conv_out - spt_backbone.py:282-303).  These ran through F.conv2d -> MIOpen until round 5: 26.7 of
// the 48 ms of config D's step, and a 5 - 7 minute kernel search on a fresh box.
//
//   Y[b, y, x, :] = sum_{ky, kx}  W[:, :, ky, kx]  X[b, y + (ky - 1) dil, x + (kx - 1) dil, :]        (zero outside the map)
//
// k_conv3x3_dense<CIB, CO, DIL>: one workgroup (4 wavefronts) = two 8 x 8 tiles of the map = 128 sites x CO output channels,
// K = phases x 9 taps x CIB input channels.  Per phase the (8 + 2 dil)^2 halo patches of both tiles are loaded once (16 bytes per
// thread and access, zero outside the map) into LDS - site pitch CIB * 2 + 16 bytes, row pitch = 128 (mod 256): conflict-free
// ds_read_b128 for the 4-rows-by-8-columns MFMA column blocks, the layout of k_conv3x3_tiles (conv_tiles.hip) - and all nine taps
// read that patch at a compile-time byte offset.  v_mfma_f32_32x32x16_bf16 computes Y^T: A = weights (32 output channels per
// wavefront, packed once per optimizer step in fragment order so a wavefront streams 1 KB per k-step from L2 into registers, eight
// steps ahead, no LDS and no barrier in the K loop), B = 32 sites from LDS.  CO = 128: wavefront w owns channel block w of all 128
// sites; CO = 64: channel block w & 1 of tile w >> 1; CO = 32: site block w.  Epilogue: (+ bias) -> bf16 -> LDS, site-major ->
// 16-byte row stores.
//
// The INPUT GRADIENT is the same kernel on the transposed, tap-flipped weight image (stride 1, padding = dilation).
//
// k_conv3x3_dense_dw<COB, DIL>: dW[co, ci, ky, kx] = sum_sites dY[site, co] X[site + tap, ci].  The contraction runs over the slow
// axis of both operands, so - as in dw_grouped.hip - the operands are staged row-major in LDS and the MFMA fragments are read
// transposed (ds_read_b64_tr_b16).  A workgroup owns a (COB output channels) x (64 input channels) x 9 taps block for a slice of
// the tiles: per tile the halo patch of X (64 channels) and the tile of dY are staged ONCE and all nine taps read the patch at
// shifted rows (nine 32 x 32 accumulators per wavefront); the next tile's rows are in flight in registers behind the products.
// fp32 partial blocks per slice, summed in a fixed order by k_cd_dw_reduce (deterministic) straight into the (cout, cin, 3, 3)
// gradient.  0.6 - 0.84 PFLOP/s at config D's map (tools/bench_conv_dense.py).  Measured and dropped (round 5): eight wavefronts of
// five accumulators (four per SIMD: 674 -> 712 us at 128 -> 128 - the loop is not latency-bound); both co blocks x five taps per
// wavefront (every X fragment feeds two products, 0.7 instead of 1.11 LDS fragment reads per product: 2.44 -> 2.63 ms at 384 -> 128).
#include "../../include/gdmae_hip.h"
#include "common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {
union CdFrag {
  uint4 q;
  bf16x8 v;
};
__device__ inline unsigned cd_pack2(float lo, float hi) { return gd_pack_bf16(lo, hi); }

template <int CIB, int DIL>
struct CdGeom {
  static constexpr int PW = 8 + 2 * DIL;                     // patch width = height (sites)
  static constexpr int SITE = CIB * 2 + 16;                  // bytes
  static constexpr int ROW_RAW = PW * SITE;
  static constexpr int ROW = ROW_RAW + ((128 - (ROW_RAW % 256)) + 256) % 256;       // = 128 (mod 256): patch rows alternate bank halves
  static constexpr int TILE = PW * ROW;
  static constexpr int KS = CIB / 16;                        // k-steps per tap
  static_assert(ROW % 256 == 128 && SITE % 128 != 0, "bank layout");
};

static inline int cd_pad32(int c) { return (c + 31) / 32 * 32; }
// input channels per phase of a convolution with cin_pad input channels
// 64 even where 128 divides: the patches of a 128-channel phase take 58.9 KB of LDS and 182 registers (two workgroups per CU), those
// of a 64-channel phase 33 KB / 156 (three): 128 -> 128 at config D's map size 633 -> 527 us, 128 -> 64 340 -> 264 us
static inline int cd_cib(int cin_pad, int dil) {
  (void)dil;
  return cin_pad % 64 == 0 ? 64 : 32;
}
static inline int cd_co(int cout_pad) { return cout_pad % 128 == 0 ? 128 : (cout_pad % 64 == 0 ? 64 : 32); }

// ------------------------------------------------------------------------------------------------
// weight images
// ------------------------------------------------------------------------------------------------
// W (cout, cin, 3, 3) fp32 -> fragments of the convolution O x I (O = out, I = in channels of the LAUNCH: swapped and tap-flipped when
// transposed): element (((g * 9 + t) * KS + ks) * MB + mb) * 64 + lane = A[o = 32 mb + (lane & 31)][i = g cib + 16 ks + 8 (lane >> 5) + j]
__global__ __launch_bounds__(256) void k_cd_pack(const float* __restrict__ W, int cout, int cin, int O, int I, int cib, int transposed,
                                                 uint4* __restrict__ Wp) {
  const int KS = cib / 16, MB = O / 32, phases = I / cib;
  const long long total = (long long)phases * 9 * KS * MB * 64;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(e & 63);
    long long r = e >> 6;
    const int mb = (int)(r % MB); r /= MB;
    const int ks = (int)(r % KS); r /= KS;
    const int t = (int)(r % 9);
    const int g = (int)(r / 9);
    const int o = mb * 32 + (lane & 31), i0 = g * cib + ks * 16 + (lane >> 5) * 8;
    const int ky = t / 3, kx = t - 3 * ky;
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + j;
      float v = 0.f;
      if (!transposed) {
        if (o < cout && i < cin) v = W[(((long long)o * cin + i) * 3 + ky) * 3 + kx];
      } else {
        if (o < cin && i < cout) v = W[(((long long)i * cin + o) * 3 + (2 - ky)) * 3 + (2 - kx)];
      }
      f[j] = v;
    }
    uint4 q;
    q.x = cd_pack2(f[0], f[1]); q.y = cd_pack2(f[2], f[3]); q.z = cd_pack2(f[4], f[5]); q.w = cd_pack2(f[6], f[7]);
    Wp[e] = q;
  }
}

// ------------------------------------------------------------------------------------------------
// forward / input gradient
// ------------------------------------------------------------------------------------------------
struct CdArgs {
  const unsigned short* X;      // (B, H, W, cin) bf16
  const uint4* Wp;
  const float* bias;            // (cout) fp32 or null
  unsigned short* Y;            // (B, H, W, cout) bf16 (OF32: fp32)
  int B, H, W, TH, TW, cin, cout, phases, mb_total;
  int n_tiles;
  int accumulate;               // OF32: the result is added to what Y holds
  int ncb;                      // CO blocks of the output (cout / CO)
  float* stat_part;             // optional (tile pairs, 2, cout) fp32: per-channel sum / sum of squares of the ROUNDED outputs of the
                                // workgroup's in-map sites (BatchNorm statistics of the layer that follows), bf16 output only
  const unsigned short* addend; // optional (B, H, W, cout) bf16 map added to the rounded result (bf16 output only): the gradient that
                                // reaches a layer's input along an identity shortcut joins the input gradient in its store pass
};

#ifndef CD_MPW128
#define CD_MPW128 1             // channel blocks per wavefront at CO = 128 (experiment switch; 2 = a 2 x 2 blocking: measured slower)
#endif

#define CD_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

// OF32: the accumulators (+ bias) leave as fp32 rows, optionally added to the map's previous content - the building block of the
// fp32-grade convolution of the parity mode (three launches on the bf16 value / remainder splits of both operands, conv3x3_f32)
template <int CIB, int CO, int DIL, bool OF32>
__global__ __launch_bounds__(256, 2) void k_conv3x3_dense(CdArgs A) {
  using Gm = CdGeom<CIB, DIL>;
  constexpr int PW = Gm::PW, SITE = Gm::SITE, ROW = Gm::ROW, TILE = Gm::TILE, KS = Gm::KS;
  constexpr int MBLK = CO / 32;                 // 32-channel blocks of the workgroup's output
  // a wavefront owns MPW channel blocks x NPW site blocks (4 wavefronts x MPW x NPW x 32 x 32 = CO x 128).  CO = 128: 2 x 2 - every
  // site fragment read from LDS feeds two products and every weight fragment two: with 1 x 4 (k_conv3x3_tiles' blocking) the four
  // ds_read_b128 per k-step of the eight resident wavefronts are exactly the CU's 128 B / clk of LDS bandwidth at the rate the matrix
  // cores could take them - the kernel ran at the LDS roof, half the MFMA one (796 TFLOP/s at 128 -> 128)
  constexpr int MPW = CD_MPW128 == 2 && CO == 128 ? 2 : 1;
  constexpr int MG = MBLK / MPW;                // wavefront groups along the channels
  constexpr int NPW = MBLK / MPW;               // 32-site blocks per wavefront
  constexpr int STEPS = 9 * KS;
  constexpr int CPS = CIB / 8;                  // 16-byte chunks per site
  constexpr int ENT = 2 * PW * PW;              // patch entries of both tiles
  constexpr int EPP = 256 / CPS;                // entries per pass of the 256 threads
  constexpr int NPASS = (ENT + EPP - 1) / EPP;
  constexpr int SP = CO * (OF32 ? 4 : 2) + 16;  // staging pitch of the epilogue
  constexpr int RING = STEPS > 8 ? 8 : STEPS - 1;       // weight prefetch distance in k-steps
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int mb = (wv % MG) * MPW, sb0 = (wv / MG) * NPW;
  // blockIdx -> (tile pair, CO block): the CO blocks of a tile pair read the same patches, so they sit next to each other on ONE XCD
  // (consecutive workgroup ids go round the eight XCDs, each with its own L2)
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int cb = q % A.ncb, tp = (q / A.ncb) * 8 + xcd;
  if (2 * tp >= A.n_tiles) return;
  int tb[2], ty0[2], tx0[2];
  bool have[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tile = tp * 2 + t;
    have[t] = tile < A.n_tiles;
    const int tl = have[t] ? tile : 0;
    tx0[t] = (tl % A.TW) * 8;
    const int r = tl / A.TW;
    ty0[t] = (r % A.TH) * 8;
    tb[t] = r / A.TH;
  }
  const int lc = tid % CPS, le = tid / CPS;
  f32x16 acc[MPW][NPW];
#pragma unroll
  for (int j = 0; j < MPW; ++j)
#pragma unroll
    for (int b = 0; b < NPW; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[j][b][i] = 0.f;
  const unsigned char* lb[NPW];
#pragma unroll
  for (int b = 0; b < NPW; ++b) {
    const int sb = sb0 + b;
    lb[b] = lds + (sb >> 1) * TILE + ((sb & 1) * 4 + ((lane & 31) >> 3)) * ROW + (lane & 7) * SITE + (lane >> 5) * 16;
  }

  for (int g = 0; g < A.phases; ++g) {
    // ---- both halo patches of this phase's CIB channels -> LDS.  Every load is unconditional (entries outside the map / past the
    // patches read site 0 and are cleared afterwards): a load behind its own branch is followed by a drain of the load counter.
    {
      uint4 q[NPASS];
      unsigned keep = 0u;
      const unsigned short* xg = A.X + g * CIB + lc * 8;
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int e = p * EPP + le;
        const int t = e / (PW * PW) < 2 ? e / (PW * PW) : 1;
        const int r = e - (e / (PW * PW)) * (PW * PW);
        const int py = r / PW, px = r - py * PW;
        const int y = ty0[t] + py - DIL, x = tx0[t] + px - DIL;
        const bool inb = e < ENT && have[t] && y >= 0 && y < A.H && x >= 0 && x < A.W;
        keep |= inb ? 1u << p : 0u;
        const long long site = inb ? ((long long)tb[t] * A.H + y) * A.W + x : 0;
        q[p] = *reinterpret_cast<const uint4*>(xg + site * A.cin);
      }
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int e = p * EPP + le;
        if (e >= ENT) continue;
        const int t = e / (PW * PW);
        const int r = e - t * (PW * PW);
        const int py = r / PW, px = r - py * PW;
        const unsigned m = ((keep >> p) & 1u) ? 0xFFFFFFFFu : 0u;
        uint4 o = q[p];
        o.x &= m; o.y &= m; o.z &= m; o.w &= m;
        *reinterpret_cast<uint4*>(lds + t * TILE + py * ROW + px * SITE + lc * 16) = o;
      }
    }
    __syncthreads();
    // ---- 9 taps x KS k-steps: weight fragments RING steps ahead through a register ring, site fragments one step ahead
    {
      const uint4* __restrict__ wp = A.Wp + ((size_t)g * STEPS * A.mb_total + cb * MBLK + mb) * 64 + lane;
      const size_t wstep = (size_t)A.mb_total * 64;
      CdFrag wr[RING + 1][MPW], sf[2][NPW];
#pragma unroll
      for (int st = 0; st < RING; ++st)
#pragma unroll
        for (int j = 0; j < MPW; ++j) wr[st][j].q = wp[st * wstep + j * 64];
#pragma unroll
      for (int b = 0; b < NPW; ++b) sf[0][b].q = *reinterpret_cast<const uint4*>(lb[b]);
#pragma unroll
      for (int st = 0; st < STEPS; ++st) {
        if (st + RING < STEPS) {
#pragma unroll
          for (int j = 0; j < MPW; ++j) wr[(st + RING) % (RING + 1)][j].q = wp[(st + RING) * wstep + j * 64];
        }
        if (st + 1 < STEPS) {
          const int tap = (st + 1) / KS, ks = (st + 1) - tap * KS;
          const int off = (tap / 3) * DIL * ROW + (tap % 3) * DIL * SITE + ks * 32;
#pragma unroll
          for (int b = 0; b < NPW; ++b) sf[(st + 1) & 1][b].q = *reinterpret_cast<const uint4*>(lb[b] + off);
        }
#pragma unroll
        for (int j = 0; j < MPW; ++j)
#pragma unroll
          for (int b = 0; b < NPW; ++b) acc[j][b] = CD_MFMA(wr[st % (RING + 1)][j].v, sf[st & 1][b].v, acc[j][b]);
        __builtin_amdgcn_sched_barrier(0);      // nothing moves across a step: the prefetch distances are what is written here
      }
    }
    __syncthreads();
  }

  // ---- epilogue: Y^T accumulators (row = channel by register, column = site by lane) (+ bias) -> bf16 site-major rows in LDS
  {
    const int n = lane & 31;
#pragma unroll
    for (int jm = 0; jm < MPW; ++jm) {
      const int c0 = (mb + jm) * 32 + 4 * (lane >> 5);
      float bv[16];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[4 * j + e] = A.bias ? A.bias[cb * CO + c0 + 8 * j + e] : 0.f;
#pragma unroll
      for (int b = 0; b < NPW; ++b) {
        const int site = (sb0 + b) * 32 + n;
        const f32x16& a = acc[jm][b];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (OF32) {
            *reinterpret_cast<float4*>(lds + site * SP + (c0 + 8 * j) * 4) =
                make_float4(a[4 * j] + bv[4 * j], a[4 * j + 1] + bv[4 * j + 1], a[4 * j + 2] + bv[4 * j + 2], a[4 * j + 3] + bv[4 * j + 3]);
          } else {
            uint2 o;
            o.x = cd_pack2(a[4 * j] + bv[4 * j], a[4 * j + 1] + bv[4 * j + 1]);
            o.y = cd_pack2(a[4 * j + 2] + bv[4 * j + 2], a[4 * j + 3] + bv[4 * j + 3]);
            *reinterpret_cast<uint2*>(lds + site * SP + (c0 + 8 * j) * 2) = o;
          }
        }
      }
    }
  }
  __syncthreads();
  {
    constexpr int OCPS = CO / (OF32 ? 4 : 8);       // 16-byte chunks per site
    for (int i = tid; i < 128 * OCPS; i += 256) {
      const int site = i / OCPS, c = i - site * OCPS;
      const int t = site >> 6, ss = site & 63;
      const int y = ty0[t] + (ss >> 3), x = tx0[t] + (ss & 7);
      if (have[t] && y < A.H && x < A.W) {
        const long long row = ((long long)tb[t] * A.H + y) * A.W + x;
        if constexpr (OF32) {
          float* dst = reinterpret_cast<float*>(A.Y) + row * A.cout + cb * CO + c * 4;
          float4 v = *reinterpret_cast<const float4*>(lds + site * SP + c * 16);
          if (A.accumulate) {
            const float4 o = *reinterpret_cast<const float4*>(dst);
            v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
          }
          *reinterpret_cast<float4*>(dst) = v;
        } else {
          uint4 v = *reinterpret_cast<const uint4*>(lds + site * SP + c * 16);
          if (A.addend != nullptr) {                 // uniform
            const uint4 o = *reinterpret_cast<const uint4*>(A.addend + row * A.cout + cb * CO + c * 8);
            const unsigned vw[4] = {v.x, v.y, v.z, v.w}, ow[4] = {o.x, o.y, o.z, o.w};
            unsigned r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              r[k] = cd_pack2(__uint_as_float(vw[k] << 16) + __uint_as_float(ow[k] << 16),
                              __uint_as_float(vw[k] & 0xFFFF0000u) + __uint_as_float(ow[k] & 0xFFFF0000u));
            v = make_uint4(r[0], r[1], r[2], r[3]);
          }
          *reinterpret_cast<uint4*>(A.Y + row * A.cout + cb * CO + c * 8) = v;
        }
      }
    }
  }
  // ---- BatchNorm statistics of the rounded outputs: a thread keeps ONE 8-channel chunk (256 % OCPS == 0) and walks the sites
  if constexpr (!OF32) {
    if (A.stat_part != nullptr) {
      constexpr int OCPS = CO / 8, TPC = 256 / OCPS;        // threads per chunk
      float sm[8], sq[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) sm[j] = sq[j] = 0.f;
      const int c = tid % OCPS;
      for (int site = tid / OCPS; site < 128; site += TPC) {
        const int t = site >> 6, ss = site & 63;
        const int y = ty0[t] + (ss >> 3), x = tx0[t] + (ss & 7);
        if (have[t] && y < A.H && x < A.W) {
          const uint4 v = *reinterpret_cast<const uint4*>(lds + site * SP + c * 16);
          const unsigned w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float lo = __uint_as_float(w4[k] << 16), hi = __uint_as_float(w4[k] & 0xFFFF0000u);
            sm[2 * k] += lo; sm[2 * k + 1] += hi;
            sq[2 * k] = fmaf(lo, lo, sq[2 * k]); sq[2 * k + 1] = fmaf(hi, hi, sq[2 * k + 1]);
          }
        }
      }
      __syncthreads();                                       // the staging tile has been read
      float* red = reinterpret_cast<float*>(lds);            // [256 threads][16]
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[tid * 16 + j] = sm[j];
        red[tid * 16 + 8 + j] = sq[j];
      }
      __syncthreads();
      if (tid < 2 * CO) {
        const int stat = tid / CO, ch = tid - stat * CO;
        float a = 0.f;
        for (int r = 0; r < TPC; ++r) a += red[(r * OCPS + (ch >> 3)) * 16 + stat * 8 + (ch & 7)];      // fixed order
        A.stat_part[((size_t)tp * 2 + stat) * A.cout + cb * CO + ch] = a;
      }
    }
  }
}

// rows [r * per, (r + 1) * per) of part (n_rows, W) summed in order -> out (n_out, W)
__global__ __launch_bounds__(256) void k_cd_stats_prereduce(const float* __restrict__ part, int n_rows, int Wd, int per, float* __restrict__ out) {
  const int r = blockIdx.x;
  for (int c = threadIdx.x; c < Wd; c += 256) {
    float a = 0.f;
    const int lo = r * per, hi = lo + per < n_rows ? lo + per : n_rows;
    int i = lo;
    for (; i + 8 <= hi; i += 8) {                  // eight rows in flight, added in row order
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(i + j) * Wd + c];
#pragma unroll
      for (int j = 0; j < 8; ++j) a += v[j];
    }
    for (; i < hi; ++i) a += part[(size_t)i * Wd + c];
    out[(size_t)r * Wd + c] = a;
  }
}

template <int CIB, int CO, int DIL, bool OF32>
int cd_launch(const CdArgs& A, hipStream_t st) {
  using Gm = CdGeom<CIB, DIL>;
  constexpr int patch = 2 * Gm::TILE, stg = 128 * (CO * (OF32 ? 4 : 2) + 16);
  constexpr int lds = patch > stg ? patch : stg;
  static bool once = false;
  if (!once) {
    GD_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_dense<CIB, CO, DIL, OF32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    once = true;
  }
  CdArgs B_ = A;
  B_.ncb = A.cout / CO;
  const unsigned ntp8 = (unsigned)gd_div_up(gd_div_up(A.n_tiles, 2), 8) * 8;           // tile pairs, padded to the XCD count
  hipLaunchKernelGGL((k_conv3x3_dense<CIB, CO, DIL, OF32>), dim3(ntp8 * (unsigned)B_.ncb), dim3(256), lds, st, B_);
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
struct CdDwArgs {
  const unsigned short* X;      // (B, H, W, cin) bf16
  const unsigned short* dY;     // (B, H, W, cout) bf16
  float* part;                  // (S, NY, 9, COB, 64) fp32
  int B, H, W, TH, TW, cin, cout, nci;
  int n_tiles, tiles_per_wg;
  int ny, n_slices;
};

// Staged 64-channel tiles ([site row][64 channels], 128-byte rows; rows = the sites of a patch / tile row by row): the two 64-byte
// segments of a row trade places on every other PAIR OF COLUMNS x of the patch, so that the four sites x ... x + 3 of a transposed read
// touch all four 64-byte bank groups - and so that the permutation depends on the column only: a lane's fragment addresses of all taps
// and k-steps are three registers (one per kx) plus compile-time offsets.
__device__ __forceinline__ int cd_off64(int row, int x, int c16) { return row * 128 + ((((c16 >> 2) ^ (x >> 1)) & 1) << 6) + ((c16 & 3) << 4); }
// one transposed 8 x 32 operand fragment = two ds_read_b64_tr_b16 at p and p + 4 sites
__device__ __forceinline__ bf16x8 cd_tr_pair(const unsigned char* __restrict__ p, int second) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + second));
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
}

template <int COB, int DIL>
__global__ __launch_bounds__(256, 2) void k_conv3x3_dense_dw(CdDwArgs A) {
  constexpr int PW = 8 + 2 * DIL, PS = PW * PW;
  constexpr int XCH = PS * 8;                             // 16-byte chunks of the X patch (64 channels)
  constexpr int XP = (XCH + 255) / 256;                   // per thread
  constexpr int GCH = 64 * (COB / 8);                     // ... of the dY tile
  constexpr int GP = GCH / 256;
  constexpr int NT = COB == 64 ? 9 : 5;                   // taps per wavefront
  static_assert(GP >= 1, "dY tile");
  extern __shared__ __align__(16) unsigned char lds[];
  unsigned char* ldx = lds;                                // patch: PS rows x 128 B
  unsigned char* ldg = lds + PS * 128;                     // dY tile: 64 rows x COB * 2 B
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // blockIdx -> (slice of tiles, (co, ci) block): the NY blocks of a slice walk the same tiles - next to each other on ONE XCD, whose L2
  // then serves the NY-fold re-reads of the slice's X patches and dY tiles
  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int by = q % A.ny, sl = (q / A.ny) * 8 + xcd;
  if (sl >= A.n_slices) return;
  const int cob = by / A.nci, cib = by - cob * A.nci;
  const int cw = COB == 64 ? (wv >> 1) : 0, iw = wv & 1;
  const int tap0 = COB == 64 ? 0 : (wv >> 1) * 5;          // COB = 32: wavefronts 0 / 1 take taps 0 .. 4, 2 / 3 taps 5 .. 8
  const int t_begin = sl * A.tiles_per_wg;
  const int t_end = t_begin + A.tiles_per_wg < A.n_tiles ? t_begin + A.tiles_per_wg : A.n_tiles;
  f32x16 acc[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
  const unsigned short* xbase = A.X + cib * 64;
  const unsigned short* gbase = A.dY + cob * COB;
  uint4 xq[XP], gq[GP];
  unsigned xkeep = 0u, gkeep = 0u;
  auto fetch = [&](int tile) {
    const int tl = tile < A.n_tiles ? tile : A.n_tiles - 1;
    const int tx0 = (tl % A.TW) * 8;
    const int r_ = tl / A.TW;
    const int ty0 = (r_ % A.TH) * 8, b = r_ / A.TH;
    xkeep = gkeep = 0u;
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int e = p * 256 + tid;
      const int s = e >> 3, c16 = e & 7;
      const int py = s / PW, px = s - py * PW;
      const int y = ty0 + py - DIL, x = tx0 + px - DIL;
      const bool inb = e < XCH && tile < A.n_tiles && y >= 0 && y < A.H && x >= 0 && x < A.W;
      xkeep |= inb ? 1u << p : 0u;
      const long long site = inb ? ((long long)b * A.H + y) * A.W + x : 0;
      xq[p] = *reinterpret_cast<const uint4*>(xbase + site * A.cin + c16 * 8);
    }
#pragma unroll
    for (int p = 0; p < GP; ++p) {
      const int e = p * 256 + tid;
      const int s = e / (COB / 8), c16 = e - s * (COB / 8);
      const int y = ty0 + (s >> 3), x = tx0 + (s & 7);
      const bool inb = tile < A.n_tiles && y < A.H && x < A.W;
      gkeep |= inb ? 1u << p : 0u;
      const long long site = inb ? ((long long)b * A.H + y) * A.W + x : 0;
      gq[p] = *reinterpret_cast<const uint4*>(gbase + site * A.cout + c16 * 8);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int e = p * 256 + tid;
      if (e >= XCH) continue;
      const unsigned m = ((xkeep >> p) & 1u) ? 0xFFFFFFFFu : 0u;
      uint4 o = xq[p];
      o.x &= m; o.y &= m; o.z &= m; o.w &= m;
      *reinterpret_cast<uint4*>(ldx + cd_off64(e >> 3, (e >> 3) % PW, e & 7)) = o;
    }
#pragma unroll
    for (int p = 0; p < GP; ++p) {
      const int e = p * 256 + tid;
      const int s = e / (COB / 8), c16 = e - s * (COB / 8);
      const unsigned m = ((gkeep >> p) & 1u) ? 0xFFFFFFFFu : 0u;
      uint4 o = gq[p];
      o.x &= m; o.y &= m; o.z &= m; o.w &= m;
      *reinterpret_cast<uint4*>(ldg + (COB == 64 ? cd_off64(s, s & 7, c16) : s * 64 + c16 * 16)) = o;
    }
  };
  // ---- this lane's fragment addresses.  A k-step = 16 sites = tile rows 2 ks, 2 ks + 1 (h = lane >> 5 picks the row), a transposed
  // read = sites x = (i >> 2) + {0, 4} of that row; the X operand of tap (ky, kx) sits (ky dil) patch rows and (kx dil) columns further
  const int fi = lane & 15, fgrp = (lane >> 4) & 1, fh = lane >> 5;
  const int fx = fi >> 2;
  const unsigned char* xa[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int px = fx + kx * DIL;
    xa[kx] = ldx + (fh * PW + px) * 128 + ((((iw ^ (px >> 1)) & 1)) << 6) + 32 * fgrp + 8 * (fi & 3);
  }
  const unsigned char* ga = COB == 64 ? ldg + (fh * 8 + fx) * 128 + ((((cw ^ (fx >> 1)) & 1)) << 6) + 32 * fgrp + 8 * (fi & 3)
                                      : ldg + (fh * 8 + fx) * 64 + 32 * fgrp + 8 * (fi & 3);
  constexpr int GROW = COB == 64 ? 128 : 64;             // bytes per staged dY site
  fetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    stage();
    __syncthreads();
    fetch(tile + 1 < t_end ? tile + 1 : tile);      // in flight behind this tile's products
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {                 // 16 sites = two tile rows per k-step
      const bf16x8 a = cd_tr_pair(ga + ks * 16 * GROW, 4 * GROW);
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const int tap = tap0 + k < 9 ? tap0 + k : 8;
        if constexpr (COB == 64) {
          const int ky = k / 3, kx = k - 3 * ky;       // compile-time: tap0 = 0
          acc[k] = CD_MFMA(a, cd_tr_pair(xa[kx] + (2 * ks + ky * DIL) * PW * 128, 4 * 128), acc[k]);
        } else {
          const int ky = tap / 3, kx = tap - 3 * ky;   // wave-uniform
          const unsigned char* xp = kx == 0 ? xa[0] : (kx == 1 ? xa[1] : xa[2]);
          acc[k] = CD_MFMA(a, cd_tr_pair(xp + (2 * ks + ky * DIL) * PW * 128, 4 * 128), acc[k]);
        }
      }
    }
    __syncthreads();
  }
  // ---- partial blocks: D[m = co by register][n = ci by lane]
  float* out = A.part + ((size_t)sl * A.ny + by) * 9 * COB * 64;
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    const int tap = tap0 + k;
    if (tap >= 9) continue;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = cw * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      out[((size_t)tap * COB + m) * 64 + iw * 32 + (lane & 31)] = acc[k][e];
    }
  }
}

// dW[co][ci][tap] += sum_s part[s][cob * nci + cib][tap][co % COB][ci % 64]      (fixed order over the slices)
__global__ __launch_bounds__(256) void k_cd_dw_reduce(const float* __restrict__ part, int S, int NY, int nci, int COB, int cout, int cin,
                                                      float* __restrict__ dW) {
  const long long total = (long long)cout * cin * 9;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int tap = (int)(e % 9);
    const long long r = e / 9;
    const int ci = (int)(r % cin), co = (int)(r / cin);
    const int y = (co / COB) * nci + ci / 64;
    const float* p = part + (((size_t)y * 9 + tap) * COB + co % COB) * 64 + ci % 64;
    const size_t stride = (size_t)NY * 9 * COB * 64;
    float a = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {                   // eight slices in flight, added in slice order
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(size_t)(s + j) * stride];
#pragma unroll
      for (int j = 0; j < 8; ++j) a += v[j];
    }
    for (; s < S; ++s) a += p[(size_t)s * stride];
    dW[e] += a;
  }
}

static int cd_dw_slices(int n_tiles, int NY, int* tiles_per_wg) {
  int S = gd_div_up(512, NY);                      // two workgroups per CU (what the registers allow) over all (co, ci) blocks
  if (S > n_tiles) S = n_tiles;
  if (S < 1) S = 1;
  const int tpw = gd_div_up(n_tiles, S);
  *tiles_per_wg = tpw;
  return gd_div_up(n_tiles, tpw);
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI (include/gdmae_hip.h)
// ------------------------------------------------------------------------------------------------
static bool cd_shape_ok(int cin, int cout, int dil) {
  return cin >= 1 && cout >= 1 && cin <= 1024 && cout <= 1024 && (dil == 1 || dil == 2);
}

extern "C" size_t gdmae_conv3x3_dense_packed_bytes(int cin, int cout) {
  return (size_t)9 * cd_pad32(cin) * cd_pad32(cout) * 2;
}

// weight (cout, cin, 3, 3) fp32.  transposed = 0: the image of the forward launch (cin_pad -> cout_pad channels); 1: the image of the
// input-gradient launch (cout_pad -> cin_pad channels, taps flipped).  Channel counts are padded to multiples of 32 with zeros.
extern "C" int gdmae_conv3x3_dense_pack(const float* weight, int cin, int cout, int dil, int transposed, void* packed, void* stream) {
  GD_REQUIRE(cd_shape_ok(cin, cout, dil), "conv3x3_dense_pack: 1..1024 channels, dilation 1 or 2");
  const int O = cd_pad32(transposed ? cin : cout), I = cd_pad32(transposed ? cout : cin);
  const int cib = cd_cib(I, dil);
  const long long total = (long long)9 * O * I / 8;
  hipLaunchKernelGGL(k_cd_pack, dim3((unsigned)gd_div_up(total, 256)), dim3(256), 0, (hipStream_t)stream, weight, cout, cin, O, I, cib, transposed,
                     (uint4*)packed);
  GD_LAUNCH_CHECK();
  return 0;
}

// Y (B, H, W, cout_l) = conv3x3(X (B, H, W, cin_l) bf16) (+ bias): cin_l / cout_l = the LAUNCH's channel counts, multiples of 32
// (the padded counts of gdmae_conv3x3_dense_pack; for the input gradient the roles of the layer's cin / cout are swapped).
// out_f32 = 0: Y bf16; 1: Y fp32, added to its previous content when accumulate != 0.
static int cd_conv(const void* X, int B, int H, int W, int cin_l, int cout_l, int dil, const void* packed, const float* bias, void* Y, int out_f32,
                   int accumulate, void* stream, float* stat_ws = nullptr, const void* addend = nullptr) {
  GD_REQUIRE(cin_l % 32 == 0 && cout_l % 32 == 0 && cd_shape_ok(cin_l, cout_l, dil), "conv3x3_dense: channel counts must be multiples of 32");
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  const int cib = cd_cib(cin_l, dil), co = cd_co(cout_l);
  CdArgs A{(const unsigned short*)X, (const uint4*)packed, bias, (unsigned short*)Y, B, H, W, (H + 7) / 8, (W + 7) / 8, cin_l, cout_l,
           cin_l / cib, cout_l / 32, 0, accumulate, 1, stat_ws, (const unsigned short*)addend};
  GD_REQUIRE(addend == nullptr || (!out_f32 && stat_ws == nullptr), "conv3x3_dense: an addend goes with a plain bf16 output");
  A.n_tiles = B * A.TH * A.TW;
  hipStream_t st = (hipStream_t)stream;
#define CD_CASE(ci, c, d) \
  if (cib == ci && co == c && dil == d) return out_f32 ? cd_launch<ci, c, d, true>(A, st) : cd_launch<ci, c, d, false>(A, st);
  CD_CASE(64, 128, 1)
  CD_CASE(64, 64, 1)
  CD_CASE(64, 32, 1)
  CD_CASE(32, 128, 1)
  CD_CASE(32, 64, 1)
  CD_CASE(32, 32, 1)
  CD_CASE(64, 128, 2)
  CD_CASE(64, 64, 2)
#undef CD_CASE
  GD_REQUIRE(false, "conv3x3_dense: unsupported shape (dilation 2 needs input channels in multiples of 64 and >= 64 output channels)");
}
extern "C" int gdmae_conv3x3_dense(const void* X, int B, int H, int W, int cin_l, int cout_l, int dil, const void* packed, const float* bias,
                                   void* Y, void* stream) {
  return cd_conv(X, B, H, W, cin_l, cout_l, dil, packed, bias, Y, 0, 0, stream);
}
// Y = bf16(bf16(conv(X) + bias) + addend): addend (B, H, W, cout_l) bf16 - what an elementwise addition of two bf16 maps after the
// convolution gives, without the pass (the shortcut gradient of a Conv-BN-ReLU block with an identity shortcut, sst_bev_backbone.py:36-40)
extern "C" int gdmae_conv3x3_dense_add(const void* X, int B, int H, int W, int cin_l, int cout_l, int dil, const void* packed, const float* bias,
                                       const void* addend, void* Y, void* stream) {
  GD_REQUIRE(addend != nullptr, "conv3x3_dense_add: addend");
  return cd_conv(X, B, H, W, cin_l, cout_l, dil, packed, bias, Y, 0, 0, stream, nullptr, addend);
}
// ... + the BatchNorm statistics of the layer that follows: stat_rows (GDMAE_CD_STAT_ROWS = 256, 2, cout_l) fp32 partial rows of the
// per-channel sum / sum of squares of the rounded outputs over all B H W sites (the format gdmae_bn_fold_partials takes) - the epilogue
// of the convolution instead of a pass over its output.  workspace: gdmae_conv3x3_dense_stats_workspace_bytes.
static const int kCdStatRows = 256;
extern "C" size_t gdmae_conv3x3_dense_stats_workspace_bytes(int B, int H, int W, int cout_l) {
  const long long ntp = gd_div_up((long long)B * ((H + 7) / 8) * ((W + 7) / 8), 2);
  return gd_align((size_t)ntp * 2 * cout_l * sizeof(float));
}
extern "C" int gdmae_conv3x3_dense_stats(const void* X, int B, int H, int W, int cin_l, int cout_l, int dil, const void* packed, const float* bias,
                                         void* Y, float* stat_rows, void* workspace, void* stream) {
  GD_REQUIRE(stat_rows != nullptr && workspace != nullptr, "conv3x3_dense_stats: statistics rows and workspace");
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  if (int rc = cd_conv(X, B, H, W, cin_l, cout_l, dil, packed, bias, Y, 0, 0, stream, (float*)workspace)) return rc;
  const int ntp = gd_div_up((long long)B * ((H + 7) / 8) * ((W + 7) / 8), 2);
  hipLaunchKernelGGL(k_cd_stats_prereduce, dim3(kCdStatRows), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, ntp, 2 * cout_l,
                     gd_div_up(ntp, kCdStatRows), stat_rows);
  GD_LAUNCH_CHECK();
  return 0;
}
extern "C" int gdmae_conv3x3_dense_stat_rows(void) { return kCdStatRows; }
extern "C" int gdmae_conv3x3_dense_f32out(const void* X, int B, int H, int W, int cin_l, int cout_l, int dil, const void* packed, const float* bias,
                                          float* Y, int accumulate, void* stream) {
  return cd_conv(X, B, H, W, cin_l, cout_l, dil, packed, bias, Y, 1, accumulate, stream);
}

static int cd_dw_cob(int cout_l) { return cout_l % 64 == 0 ? 64 : 32; }
extern "C" size_t gdmae_conv3x3_dense_dw_workspace_bytes(int B, int H, int W, int cin_l, int cout_l) {
  if (cin_l % 64 != 0 || cout_l % 32 != 0) return 0;
  const int cob = cd_dw_cob(cout_l), NY = (cout_l / cob) * (cin_l / 64);
  int tpw = 0;
  const int S = cd_dw_slices(B * ((H + 7) / 8) * ((W + 7) / 8), NY, &tpw);
  return gd_align((size_t)S * NY * 9 * cob * 64 * sizeof(float));
}
// dW (cout, cin, 3, 3) fp32 ACCUMULATED: X (B, H, W, cin_l) bf16, dY (B, H, W, cout_l) bf16 with cin_l = cin padded to a multiple of 64,
// cout_l = cout padded to a multiple of 32 (padding channels are not written)
extern "C" int gdmae_conv3x3_dense_bwd_weight(const void* X, const void* dY, int B, int H, int W, int cin_l, int cout_l, int cin, int cout, int dil,
                                              float* dW, void* workspace, void* stream) {
  GD_REQUIRE(cin_l % 64 == 0 && cout_l % 32 == 0 && cin <= cin_l && cout <= cout_l && cd_shape_ok(cin, cout, dil),
             "conv3x3_dense_bwd_weight: cin_l multiple of 64, cout_l multiple of 32");
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int cob = cd_dw_cob(cout_l), nci = cin_l / 64, NY = (cout_l / cob) * nci;
  CdDwArgs A{(const unsigned short*)X, (const unsigned short*)dY, (float*)workspace, B, H, W, (H + 7) / 8, (W + 7) / 8, cin_l, cout_l, nci, 0, 0, NY, 0};
  A.n_tiles = B * A.TH * A.TW;
  const int S = cd_dw_slices(A.n_tiles, NY, &A.tiles_per_wg);
  A.n_slices = S;
  const int PW = 8 + 2 * dil;
  const int lds = PW * PW * 128 + 64 * cob * 2;
#define CD_DW(c, d)                                                                                                               \
  if (cob == c && dil == d) {                                                                                                     \
    static bool once = false;                                                                                                     \
    if (!once) {                                                                                                                  \
      GD_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_dense_dw<c, d>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));    \
      once = true;                                                                                                                \
    }                                                                                                                             \
    hipLaunchKernelGGL((k_conv3x3_dense_dw<c, d>), dim3((unsigned)(gd_div_up(S, 8) * 8 * NY)), dim3(256), lds, st, A);            \
  }
  CD_DW(64, 1)
  CD_DW(64, 2)
  CD_DW(32, 1)
  CD_DW(32, 2)
#undef CD_DW
  GD_LAUNCH_CHECK();
  const long long total = (long long)cout * cin * 9;
  hipLaunchKernelGGL(k_cd_dw_reduce, dim3((unsigned)gd_div_up(total, 256)), dim3(256), 0, st, (const float*)workspace, S, NY, nci, cob, cout, cin, dW);
  GD_LAUNCH_CHECK();
  return 0;
}
