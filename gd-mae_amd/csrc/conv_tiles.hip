// Decoder conv_out (Conv2d 3x3, pad 1, no bias; reference pcdet/models/backbones_3d/spt_backbone_mae.py:46-52,
// 125-133) as a 16-bit-MFMA implicit GEMM (fp16 operands since round 6, bf16 output) over the ACTIVE TILES of the BEV map, with the BatchNorm2d statistics of its
// output fused (replaces MIOpen's dense forward convolution, the 1.35 GB background fill + three row scatters that
// built its 384-channel input map, and the statistics pass over its dense output).
//
// Input map, never materialised: source stage g contributes 128 channels; at a site covered by an active token of
// that stage the value is relu(a_g * P_g[row] + b_g) (ConvTranspose2d(k = s) output row of (token, dy, dx), folded
// BatchNorm2d + ReLU), at every other in-bounds site it is the per-channel constant relu(b_g), outside the map 0.
// Output: tile-compact rows (conv_tiles.h).  Sites outside the active tiles see only the constant input, so their
// output is one of 9 border-class constants (gdmae_conv3x3_tiles_pack) and enters the statistics in closed form.
//
// k_conv3x3_tiles: one workgroup (4 wavefronts) = two 8x8 tiles = 128 sites x 128 output channels, K = 9 taps x 384.
//   * per source stage ("phase"): the 10x10 halo patches of both tiles (128 channels) are gathered through the stage's
//     cell -> token map, normalised, rounded to fp16 and laid out in LDS with a 272-byte site pitch / 2944-byte row
//     pitch (conflict-free ds_read_b128 for the 4-rows-by-8-columns MFMA column blocks); all 9 taps read the same
//     patch at a compile-time byte offset.
//   * MFMA v_mfma_f32_32x32x16_f16 computes Y^T: A = weights (rows = 32 output channels of the wavefront, packed once
//     per step in fragment order so a wavefront streams 1 KB per k-step straight from L2 into VGPRs, no LDS, no
//     barrier in the K loop), B = 32 sites from LDS.  Wavefront w owns output channels [32 w, 32 w + 32) of all 128
//     sites: 4 accumulators, 1 weight fragment + 4 site fragments per 4 MFMAs.
//   * epilogue: accumulators -> bf16 -> LDS (site-major) -> 16-byte coalesced stores of the tile rows; the same pass
//     accumulates per-channel sum / sum of squares of the ROUNDED values over the tile's in-map sites.
#include "conv_tiles.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
union CtFrag {
  uint4 q;
  bf16x8 v;
  f16x8 h;
};

#define CT_C 128                 // channels per source stage and output channels
#define CT_SITE_PITCH 272        // staged OUTPUT rows: 128 bf16 + 16 bytes
// Channels of a source stage per phase (patch gather + 9 taps of products).  128: one phase per stage, 58.9 KB of patches, two
// workgroups per CU.  64 (experiment switch, round 5 - what paid in conv_dense.hip, 633 -> 527 us at 128 -> 128): two phases per stage
// on half the channels each - 33 KB of patches, three workgroups per CU, the same weight image (a phase walks the k-steps hf * 4 ...
// hf * 4 + 3 of every tap).  Measured HERE: 576 vs 565 us per step (the map gather + BatchNorm of the patch entries needs 213
// registers; at the 168 of three workgroups per CU 30 of them spill) - not the default.
#ifndef CT_CH
#define CT_CH 128
#endif
#ifndef CT_RING
#define CT_RING 8                                   // weight prefetch distance in k-steps
#endif
#define CT_HALVES (CT_C / CT_CH)
#define CT_KS (CT_CH / 16)                          // k-steps per tap and phase
#define CT_NSTEP (9 * CT_KS)
#define CT_CPS (CT_CH / 8)                          // 16-byte chunks per patch site
#define CT_PSITE (CT_CH * 2 + 16)                   // patch site pitch: consecutive sites advance one 16-byte bank quad
#define CT_ROW_PITCH (CT_CH == 128 ? 2944 : 1664)   // 10 sites, padded to 128 (mod 256): patch rows alternate bank halves
#define CT_TILE_PITCH (10 * CT_ROW_PITCH)           // 10 patch rows
// CT_TPW tiles per workgroup (experiment switch): 2 = four wavefronts, two workgroups per CU; 4 = eight wavefronts, ONE workgroup per
// CU whose wavefronts w and w + 4 stream the same weight fragments right after the same barrier (the second request meets the line
// in the CU's L1: half the L2 -> CU weight stream for the same wavefronts per CU)
#ifndef CT_TPW
#define CT_TPW 2
#endif
#define CT_THREADS (CT_TPW * 128)
#define CT_NWAVES (CT_TPW * 2)
#define CT_STAGE_PITCH 17408     // 64 sites x 272: epilogue staging of one tile
#define CT_RED_OFF (CT_TPW * CT_STAGE_PITCH)        // after the staging areas: (tiles, waves, 2 stats, 128) fp32
#define CT_EPI_BYTES (CT_RED_OFF + CT_TPW * CT_NWAVES * 2 * CT_C * 4)
#define CT_PATCH_BYTES (CT_TPW * CT_TILE_PITCH)     // the patches (58880 for two tiles of 128 channels, 33280 of 64)
#define CT_LDS_BYTES (CT_PATCH_BYTES > CT_EPI_BYTES ? CT_PATCH_BYTES : CT_EPI_BYTES)
#define CT_SPP (CT_THREADS / CT_CPS)                // patch entries per gather pass
#define CT_NPASS ((100 * CT_TPW + CT_SPP - 1) / CT_SPP)
#define CT_ESPP (CT_THREADS / 16)                   // output sites per epilogue pass (16 chunks per 128-channel row)
#define CT_MAX_SRC 3
#define CT_NRED 256              // rows the per-tile statistics partials are pre-reduced to

__device__ inline float ct_bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ inline unsigned short ct_f2bf(float f) { return gd_to_bf16(f); }
__device__ inline unsigned ct_pack2(float lo, float hi) { return gd_pack_bf16(lo, hi); }
__device__ inline float ct_f16r(float f) {          // f rounded to fp16 the way gd_pack_f16 rounds it, back in fp32
  const unsigned p = gd_pack_f16(f, 0.f);
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(p & 0xFFFFu));
}
__device__ inline void ct_unpack8(const uint4& u, float (&f)[8]) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(w[k] << 16);
    f[2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u);
  }
}

// ------------------------------------------------------------------------------------------------
// Active tiles (geometry only: built with the plan, off the training stream)
// ------------------------------------------------------------------------------------------------
struct CtMaps {
  const int* map[CT_MAX_SRC];
  int ls[CT_MAX_SRC];   // log2 of the stage's upsampling stride
  int k;
};

// one wavefront per tile: is any in-map site of the 10x10 halo patch covered by a token of any source stage?
__global__ __launch_bounds__(256) void k_ct_tile_flags(CtMaps Mp, int B, int H, int W, int TH, int TW, int* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (t >= B * TH * TW) return;
  const int tx = t % TW, r = t / TW, ty = r % TH, b = r / TH;
  bool act = false;
  for (int e = lane; e < 100; e += 64) {
    const int y = ty * 8 - 1 + e / 10, x = tx * 8 - 1 + e % 10;
    if (y < 0 || y >= H || x < 0 || x >= W) continue;
    for (int g = 0; g < Mp.k; ++g) {
      const int ls = Mp.ls[g];
      act |= Mp.map[g][(b * (H >> ls) + (y >> ls)) * (W >> ls) + (x >> ls)] >= 0;
    }
  }
  const unsigned long long m = __ballot(act);
  if (lane == 0) flag[t] = m != 0ull;
}

struct CtFlagLoad {
  const int* flag;
  __device__ int operator()(long long i) const { return flag[i]; }
};
struct CtSlotStore {
  int* slot;
  int* list;
  __device__ void operator()(long long i, int ex, int v) const {
    slot[i] = v ? ex : -1;
    if (v) list[ex] = (int)i;
  }
};

extern "C" size_t gdmae_decoder_tiles_workspace_bytes(int B, int H, int W) {
  const long long nt = (long long)B * ((H + 7) / 8) * ((W + 7) / 8);
  return gd_align(sizeof(int) * nt) + gd_align(sizeof(int) * (gd_scan_ws_elems(nt) + 2)) + 256;
}

// maps / strides: HOST arrays of k device pointers / ints (stride of stage g = H / its map's Y, a power of two).
// tile_slot (B*TH*TW), tile_list (capacity B*TH*TW, ascending tile ids), n_act: device int.
extern "C" int gdmae_decoder_tiles(const int* const* maps, const int* strides, int k, int B, int H, int W, int* tile_slot,
                                   int* tile_list, int* n_act, void* workspace, void* stream) {
  GD_REQUIRE(k >= 1 && k <= CT_MAX_SRC, "decoder_tiles: 1..3 source stages");
  GD_REQUIRE(H >= 2 && W >= 2 && B >= 1, "decoder_tiles: map too small");
  hipStream_t st = (hipStream_t)stream;
  CtMaps Mp;
  Mp.k = k;
  for (int g = 0; g < k; ++g) {
    const int s = strides[g];
    GD_REQUIRE(s == 1 || s == 2 || s == 4 || s == 8, "decoder_tiles: stride must be 1, 2, 4 or 8");
    GD_REQUIRE(H % s == 0 && W % s == 0, "decoder_tiles: stride must divide the map");
    Mp.map[g] = maps[g];
    Mp.ls[g] = s == 1 ? 0 : (s == 2 ? 1 : (s == 4 ? 2 : 3));
  }
  const int TH = (H + 7) / 8, TW = (W + 7) / 8;
  const long long nt = (long long)B * TH * TW;
  GdArena A(workspace, gdmae_decoder_tiles_workspace_bytes(B, H, W));
  int* flag = A.take<int>(nt);
  int* scan_ws = A.take<int>(gd_scan_ws_elems(nt) + 2);
  hipLaunchKernelGGL(k_ct_tile_flags, dim3(gd_div_up(nt, 4)), dim3(256), 0, st, Mp, B, H, W, TH, TW, flag);
  GD_LAUNCH_CHECK();
  return gd_device_scan<int>(nt, CtFlagLoad{flag}, CtSlotStore{tile_slot, tile_list}, n_act, scan_ws, st);
}

// The same with the single-launch scan (geometry plan, plan.hip): flag (nt ints) + a ZEROED look-back state of
// gd_decoder_tiles_lb_state_bytes(nt) bytes; two launches.
size_t gd_decoder_tiles_lb_state_bytes(long long nt) { return gd_scan_lb_state_bytes<int>(nt); }
int gd_decoder_tiles_lb(const int* const* maps, const int* strides, int k, int B, int H, int W, int* tile_slot, int* tile_list, int* n_act,
                        int* flag, void* lb_state, hipStream_t st) {
  GD_REQUIRE(k >= 1 && k <= CT_MAX_SRC, "decoder_tiles: 1..3 source stages");
  CtMaps Mp;
  Mp.k = k;
  for (int g = 0; g < k; ++g) {
    const int s = strides[g];
    GD_REQUIRE((s == 1 || s == 2 || s == 4 || s == 8) && H % s == 0 && W % s == 0, "decoder_tiles: stride must be 1, 2, 4 or 8 and divide the map");
    Mp.map[g] = maps[g];
    Mp.ls[g] = s == 1 ? 0 : (s == 2 ? 1 : (s == 4 ? 2 : 3));
  }
  const int TH = (H + 7) / 8, TW = (W + 7) / 8;
  const long long nt = (long long)B * TH * TW;
  hipLaunchKernelGGL(k_ct_tile_flags, dim3(gd_div_up(nt, 4)), dim3(256), 0, st, Mp, B, H, W, TH, TW, flag);
  GD_LAUNCH_CHECK();
  return gd_device_scan_lb<int>(nt, CtFlagLoad{flag}, CtSlotStore{tile_slot, tile_list}, GdNoTotal{}, n_act, lb_state, st);
}

// ------------------------------------------------------------------------------------------------
// Per-step preparation: weights in MFMA-fragment order, background row, border-class constants
// ------------------------------------------------------------------------------------------------
// Wp[((g * 72 + tap * 8 + ks) * 4 + w) * 64 + lane] = 8 fp16: output channel 32 w + (lane & 31), input channels
// g * 128 + ks * 16 + (lane >> 5) * 8 + j, tap = ky * 3 + kx of conv_w (C2, Cin, 3, 3)
__global__ __launch_bounds__(256) void k_ct_pack_weights(const float* __restrict__ w, int Cin, int nsteps, uint4* __restrict__ Wp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsteps * 4 * 64) return;
  const int lane = i & 63, wv = (i >> 6) & 3, st = i >> 8;
  const int ks = st & 7, tap = (st >> 3) % 9, g = (st >> 3) / 9;
  const int o = 32 * wv + (lane & 31);
  const int ci = g * CT_C + ks * 16 + (lane >> 5) * 8;
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = w[((long long)o * Cin + ci + j) * 9 + tap];
  uint4 q;                      // fp16 values (round 6): see gd_pack_f16, common.h
  q.x = gd_pack_f16(f[0], f[1]);
  q.y = gd_pack_f16(f[2], f[3]);
  q.z = gd_pack_f16(f[4], f[5]);
  q.w = gd_pack_f16(f[6], f[7]);
  Wp[i] = q;
}

struct CtBPtrs {
  const float* b[CT_MAX_SRC];
};

// one workgroup per output channel o: t[k] = sum_c f16(W[o][c][k]) * bg[c] with bg[c] = f16(relu(b[c])) - the operands the tile kernel
// multiplies -, then the 9 border-class constants ybg[cls][o] = bf16(sum of t[k] over the taps of the class that fall inside the map).
// Block 0 also writes the background row bgz (Cin) bf16 (what the backward subtracts from its bf16 operand rows).
__global__ __launch_bounds__(256) void k_ct_class_consts(const float* __restrict__ w, CtBPtrs Bp, int Cin, int C2,
                                                         unsigned short* __restrict__ bgz, unsigned short* __restrict__ ybg) {
  __shared__ double sh[9][256];
  const int o = blockIdx.x;
  double t[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) t[k] = 0.0;
  for (int c = threadIdx.x; c < Cin; c += 256) {
    const float bv = Bp.b[c / CT_C][c % CT_C];
    if (o == 0) bgz[c] = ct_f2bf(bv > 0.f ? bv : 0.f);
    const double bg = (double)ct_f16r(bv > 0.f ? bv : 0.f);
#pragma unroll
    for (int k = 0; k < 9; ++k) t[k] += (double)ct_f16r(w[((long long)o * Cin + c) * 9 + k]) * bg;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) sh[k][threadIdx.x] = t[k];
  __syncthreads();
  if (threadIdx.x < 9) {
    double s = 0.0;
    for (int i = 0; i < 256; ++i) s += sh[threadIdx.x][i];
    sh[threadIdx.x][0] = s;
  }
  __syncthreads();
  if (threadIdx.x < 9) {
    const int cy = threadIdx.x / 3, cx = threadIdx.x % 3;
    double s = 0.0;
    for (int ky = 0; ky < 3; ++ky) {
      if ((cy == 0 && ky == 0) || (cy == 2 && ky == 2)) continue;   // tap row outside the map
      for (int kx = 0; kx < 3; ++kx) {
        if ((cx == 0 && kx == 0) || (cx == 2 && kx == 2)) continue;
        s += sh[ky * 3 + kx][0];
      }
    }
    ybg[threadIdx.x * C2 + o] = ct_f2bf((float)s);
  }
}

extern "C" size_t gdmae_conv3x3_tiles_packed_bytes(int k) { return (size_t)k * 72 * 4 * 64 * 16; }

// conv_w (C2 = 128, Cin = 128 k, 3, 3) fp32; b: HOST array of k device pointers to the folded BatchNorm shifts (128 each).
// Wp: gdmae_conv3x3_tiles_packed_bytes(k); bgz (Cin) bf16; ybg (9, 128) bf16.
extern "C" int gdmae_conv3x3_tiles_pack(const float* conv_w, int C2, int Cin, const float* const* b, int k, void* Wp, void* bgz,
                                        void* ybg, void* stream) {
  GD_REQUIRE(C2 == CT_C && k >= 1 && k <= CT_MAX_SRC && Cin == CT_C * k, "conv3x3_tiles: 128 output channels, 128 per source");
  hipStream_t st = (hipStream_t)stream;
  const int nsteps = k * 72;
  hipLaunchKernelGGL(k_ct_pack_weights, dim3(nsteps), dim3(256), 0, st, conv_w, Cin, nsteps, (uint4*)Wp);
  GD_LAUNCH_CHECK();
  CtBPtrs Bp;
  for (int g = 0; g < CT_MAX_SRC; ++g) Bp.b[g] = g < k ? b[g] : nullptr;
  hipLaunchKernelGGL(k_ct_class_consts, dim3(C2), dim3(256), 0, st, conv_w, Bp, Cin, C2, (unsigned short*)bgz, (unsigned short*)ybg);
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// The convolution
// ------------------------------------------------------------------------------------------------
struct CtSrc {
  const unsigned short* P;   // (rows, 128) bf16 deconvolution outputs, row = token * s*s + dy * s + dx
  const int* map;            // (B * H/s * W/s) cell -> token / -1
  const float* a;            // folded BatchNorm scale (128)
  const float* b;            // folded BatchNorm shift (128)
  int ls;                    // log2 s
};
struct CtArgs {
  CtSrc src[CT_MAX_SRC];
  int k;
  const uint4* Wp;
  const int* tile_list;
  int n_act, H, W, TH, TW;
  unsigned short* Yc;        // (n_act * 64, 128) bf16
  float* part;               // (n_act, 2, 128) per-tile sum / sum of squares over the in-map sites
};

#define CT_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// byte offset of k-step st of a phase (tap st / CT_KS, 16-channel step st % CT_KS) inside the LDS patch
#define CT_OFF(st) ((((st) / CT_KS) / 3) * CT_ROW_PITCH + (((st) / CT_KS) % 3) * CT_PSITE + ((st) % CT_KS) * 32)
// the same k-step in the packed image of a stage (8 k-steps per tap)
#define CT_WSTEP(st, hf) ((((st) / CT_KS) * 8 + (hf) * CT_KS + ((st) % CT_KS)) * 256)

__global__ __launch_bounds__(CT_THREADS, CT_TPW == 2 ? (CT_CH == 64 ? 3 : 2) : 1) void k_conv3x3_tiles(CtArgs A) {
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wvg = tid >> 6;
  const int wv = wvg & 3;              // output-channel block of the wavefront
  const int tg = wvg >> 2;             // its pair of tiles
  const int slot0 = blockIdx.x * CT_TPW;
  int tb[CT_TPW], ty0[CT_TPW], tx0[CT_TPW];
  bool have[CT_TPW];
#pragma unroll
  for (int t = 0; t < CT_TPW; ++t) {
    have[t] = slot0 + t < A.n_act;
    const int tile = have[t] ? A.tile_list[slot0 + t] : 0;
    tx0[t] = (tile % A.TW) * 8;
    const int r = tile / A.TW;
    ty0[t] = (r % A.TH) * 8;
    tb[t] = r / A.TH;
  }
  const int lc = tid % CT_CPS;   // 16-byte chunk (8 channels) of a phase's CT_CH channels of a patch site
  const int lsg = tid / CT_CPS;  // patch entry within a pass
  // site operand: lane n = lane & 31 -> site (row n >> 3 of a 4-row half tile, column n & 7), k-group lane >> 5
  const unsigned char* lb = lds + tg * 2 * CT_TILE_PITCH + ((lane & 31) >> 3) * CT_ROW_PITCH + (lane & 7) * CT_PSITE + (lane >> 5) * 16;
  f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = acc2[i] = acc3[i] = 0.f;

  for (int g = 0; g < A.k; ++g) {
    const CtSrc S = A.src[g];
    const int ls = S.ls, sm = (1 << ls) - 1;
    const int Hs = A.H >> ls, Ws = A.W >> ls;
    // The cell -> token lookups of a thread are requested together and unconditionally (entries outside the map or past the
    // patches read cell 0 and ignore it), then the rows (row 0 of a readable stand-in for background / outside / unused entries
    // and for a stage without rows): behind its own `if` every load was followed by a drain of the load counter at the join.
    int code[CT_NPASS];   // >= 0: row of P, -1: background, -2: zero (outside the map), -3: nothing to write
    {
      int tokv[CT_NPASS];
#pragma unroll
      for (int p = 0; p < CT_NPASS; ++p) {
        const int e = p * CT_SPP + lsg;
        const int t = e / 100 < CT_TPW ? e / 100 : CT_TPW - 1;
        const int r = e - 100 * (e / 100);
        const int py = r / 10, px = r - py * 10;
        const int y = ty0[t] + py - 1, x = tx0[t] + px - 1;
        const bool inb = e < 100 * CT_TPW && have[t] && y >= 0 && y < A.H && x >= 0 && x < A.W;
        tokv[p] = S.map[inb ? (tb[t] * Hs + (y >> ls)) * Ws + (x >> ls) : 0];
      }
#pragma unroll
      for (int p = 0; p < CT_NPASS; ++p) {
        const int e = p * CT_SPP + lsg;
        int c = -3;
        if (e < 100 * CT_TPW) {
          const int t = e / 100;
          const int r = e - 100 * t;
          const int py = r / 10, px = r - py * 10;
          if (have[t]) {
            const int y = ty0[t] + py - 1, x = tx0[t] + px - 1;
            c = -2;
            if (y >= 0 && y < A.H && x >= 0 && x < A.W) {
              const int tok = tokv[p];
              c = tok < 0 ? -1 : (((tok << ls) + (y & sm)) << ls) + (x & sm);
            }
          }
        }
        code[p] = c;
      }
    }
    const unsigned short* __restrict__ Prows = S.P ? S.P : (const unsigned short*)S.a;      // a: 128 floats, readable as one row
#pragma unroll 1
    for (int hf = 0; hf < CT_HALVES; ++hf) {
      // ---- gather + normalise this phase's channels of the two halo patches into LDS
      {
        const int c0 = hf * CT_CH + lc * 8;
        float av[8], bv[8];
        {
          const float4 a0 = *(const float4*)(S.a + c0), a1 = *(const float4*)(S.a + c0 + 4);
          const float4 b0 = *(const float4*)(S.b + c0), b1 = *(const float4*)(S.b + c0 + 4);
          av[0] = a0.x; av[1] = a0.y; av[2] = a0.z; av[3] = a0.w; av[4] = a1.x; av[5] = a1.y; av[6] = a1.z; av[7] = a1.w;
          bv[0] = b0.x; bv[1] = b0.y; bv[2] = b0.z; bv[3] = b0.w; bv[4] = b1.x; bv[5] = b1.y; bv[6] = b1.z; bv[7] = b1.w;
        }
        uint4 bgq;
        {
          float r[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) r[j] = bv[j];
          bgq.x = gd_pack_f16_relu(r[0], r[1]); bgq.y = gd_pack_f16_relu(r[2], r[3]); bgq.z = gd_pack_f16_relu(r[4], r[5]); bgq.w = gd_pack_f16_relu(r[6], r[7]);
        }
        uint4 q[CT_NPASS];
#pragma unroll
        for (int p = 0; p < CT_NPASS; ++p) q[p] = *(const uint4*)(Prows + (long long)(code[p] >= 0 ? code[p] : 0) * CT_C + c0);
#pragma unroll
        for (int p = 0; p < CT_NPASS; ++p) {
          if (code[p] == -3) continue;
          const int e = p * CT_SPP + lsg;
          const int t = e / 100;
          const int r = e - 100 * t;
          const int py = r / 10, px = r - py * 10;
          uint4 o;
          if (code[p] >= 0) {
            float f[8];
            ct_unpack8(q[p], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaf(av[j], f[j], bv[j]);
            // ReLU + fp16 range clamp (one v_med3_f32) + conversion: the site operands are fp16 like the weight image
            o.x = gd_pack_f16_relu(f[0], f[1]); o.y = gd_pack_f16_relu(f[2], f[3]); o.z = gd_pack_f16_relu(f[4], f[5]); o.w = gd_pack_f16_relu(f[6], f[7]);
          } else if (code[p] == -1) {
            o = bgq;
          } else {
            o = make_uint4(0u, 0u, 0u, 0u);
          }
          *(uint4*)(lds + t * CT_TILE_PITCH + py * CT_ROW_PITCH + px * CT_PSITE + lc * 16) = o;
        }
      }
      __syncthreads();
      // ---- 9 taps x CT_KS k-steps: weight fragments stream from L2 eight steps ahead through a 9-slot register ring,
      //      site fragments come from the LDS patch one step ahead; per step 1 global load, 4 LDS reads, 4 MFMAs
      {
        const uint4* __restrict__ wp = A.Wp + ((size_t)g * 72 * 4 + wv) * 64 + lane + (size_t)hf * CT_KS * 256;
        CtFrag wr[CT_RING + 1], sf[2][4];
#pragma unroll
        for (int st = 0; st < CT_RING; ++st) wr[st].q = wp[CT_WSTEP(st, 0)];
        sf[0][0].q = *(const uint4*)(lb);
        sf[0][1].q = *(const uint4*)(lb + 4 * CT_ROW_PITCH);
        sf[0][2].q = *(const uint4*)(lb + CT_TILE_PITCH);
        sf[0][3].q = *(const uint4*)(lb + CT_TILE_PITCH + 4 * CT_ROW_PITCH);
#pragma unroll
        for (int st = 0; st < CT_NSTEP; ++st) {
          if (st + CT_RING < CT_NSTEP) wr[(st + CT_RING) % (CT_RING + 1)].q = wp[CT_WSTEP(st + CT_RING, 0)];
          if (st + 1 < CT_NSTEP) {
            sf[(st + 1) & 1][0].q = *(const uint4*)(lb + CT_OFF(st + 1));
            sf[(st + 1) & 1][1].q = *(const uint4*)(lb + 4 * CT_ROW_PITCH + CT_OFF(st + 1));
            sf[(st + 1) & 1][2].q = *(const uint4*)(lb + CT_TILE_PITCH + CT_OFF(st + 1));
            sf[(st + 1) & 1][3].q = *(const uint4*)(lb + CT_TILE_PITCH + 4 * CT_ROW_PITCH + CT_OFF(st + 1));
          }
          acc0 = CT_MFMA(wr[st % (CT_RING + 1)].h, sf[st & 1][0].h, acc0);
          acc1 = CT_MFMA(wr[st % (CT_RING + 1)].h, sf[st & 1][1].h, acc1);
          acc2 = CT_MFMA(wr[st % (CT_RING + 1)].h, sf[st & 1][2].h, acc2);
          acc3 = CT_MFMA(wr[st % (CT_RING + 1)].h, sf[st & 1][3].h, acc3);
          __builtin_amdgcn_sched_barrier(0);   // nothing moves across a step: the prefetch distances are what is written here
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: Y^T accumulators (row = channel by register, column = site by lane) -> bf16 site-major rows in LDS
  {
    const int n = lane & 31;
    const int cb = wv * 32 + 4 * (lane >> 5);
#define CT_STAGE(acc, sb)                                                                                              \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
    uint2 o;                                                                                                           \
    o.x = ct_pack2(acc[4 * j], acc[4 * j + 1]);                                                                        \
    o.y = ct_pack2(acc[4 * j + 2], acc[4 * j + 3]);                                                                    \
    *(uint2*)(lds + (tg * 2 + ((sb) >> 1)) * CT_STAGE_PITCH + (((sb) & 1) * 32 + n) * CT_SITE_PITCH + (cb + 8 * j) * 2) = o; \
  }
    CT_STAGE(acc0, 0)
    CT_STAGE(acc1, 1)
    CT_STAGE(acc2, 2)
    CT_STAGE(acc3, 3)
#undef CT_STAGE
  }
  __syncthreads();
  float* red = (float*)(lds + CT_RED_OFF);
  const int elc = tid & 15, elsg = tid >> 4;      // 16-byte chunk of a 128-channel output row, site within a pass
#pragma unroll
  for (int t = 0; t < CT_TPW; ++t) {
    if (!have[t]) continue;
    float s[8], q2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = q2[j] = 0.f;
#pragma unroll
    for (int p = 0; p < 64 / CT_ESPP; ++p) {
      const int site = p * CT_ESPP + elsg;
      const uint4 v = *(const uint4*)(lds + t * CT_STAGE_PITCH + site * CT_SITE_PITCH + elc * 16);
      *(uint4*)(A.Yc + ((long long)(slot0 + t) * GD_TILE_SITES + site) * CT_C + elc * 8) = v;
      if (ty0[t] + (site >> 3) < A.H && tx0[t] + (site & 7) < A.W) {
        float f[8];
        ct_unpack8(v, f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          s[j] += f[j];
          q2[j] = fmaf(f[j], f[j], q2[j]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s[j] += __shfl_xor(s[j], 16, 64);
      s[j] += __shfl_xor(s[j], 32, 64);
      q2[j] += __shfl_xor(q2[j], 16, 64);
      q2[j] += __shfl_xor(q2[j], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[((t * CT_NWAVES + wvg) * 2 + 0) * CT_C + elc * 8 + j] = s[j];
        red[((t * CT_NWAVES + wvg) * 2 + 1) * CT_C + elc * 8 + j] = q2[j];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int t = 0; t < CT_TPW; ++t) {
    if (!have[t] || tid >= 256) continue;
    const int stat = tid >> 7, ch = tid & 127;
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < CT_NWAVES; ++w) v += red[((t * CT_NWAVES + w) * 2 + stat) * CT_C + ch];
    A.part[(long long)(slot0 + t) * 256 + tid] = v;
  }
}

// Statistics: the n_act per-tile partial rows are pre-reduced to CT_NRED rows (fixed order), and two more rows carry
// the closed-form share of every site outside the active tiles (count per border class x class constant, split into a
// float hi / lo pair so the fp64 combine sees the exact product).
__global__ __launch_bounds__(256) void k_ct_stats_reduce(const float* __restrict__ part, int n_act, const int* __restrict__ tile_list,
                                                         const unsigned short* __restrict__ ybg, int B, int H, int W, int TH, int TW,
                                                         float* __restrict__ red) {
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < CT_NRED) {
    const int chunk = (n_act + CT_NRED - 1) / CT_NRED;
    const int r0 = blockIdx.x * chunk, r1 = r0 + chunk < n_act ? r0 + chunk : n_act;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int r = r0;
    for (; r + 15 < r1; r += 16) {                 // sixteen rows in flight, added in the order of the four-row loop below (same bits)
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = part[(long long)(r + u) * 256 + tid];
#pragma unroll
      for (int u = 0; u < 16; u += 4) {
        a0 += v[u];
        a1 += v[u + 1];
        a2 += v[u + 2];
        a3 += v[u + 3];
      }
    }
    for (; r + 3 < r1; r += 4) {
      a0 += part[(long long)r * 256 + tid];
      a1 += part[(long long)(r + 1) * 256 + tid];
      a2 += part[(long long)(r + 2) * 256 + tid];
      a3 += part[(long long)(r + 3) * 256 + tid];
    }
    for (; r < r1; ++r) a0 += part[(long long)r * 256 + tid];
    red[(long long)blockIdx.x * 256 + tid] = (a0 + a1) + (a2 + a3);
    return;
  }
  // sites of each border class inside the active tiles
  __shared__ int cnt[9];
  if (tid < 9) cnt[tid] = 0;
  __syncthreads();
  int c[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) c[k] = 0;
  for (int i0 = tid; i0 < n_act; i0 += 8 * 256) {        // eight list entries in flight per thread (one workgroup walks 12 k of them)
   int tl[8];
#pragma unroll
   for (int u = 0; u < 8; ++u) tl[u] = tile_list[i0 + u * 256 < n_act ? i0 + u * 256 : n_act - 1];
#pragma unroll
   for (int u = 0; u < 8; ++u) {
    if (i0 + u * 256 >= n_act) continue;
    const int tile = tl[u];
    const int tx = tile % TW, ty = (tile / TW) % TH;
    const int y0 = ty * 8, y1 = y0 + 8 < H ? y0 + 8 : H, x0 = tx * 8, x1 = x0 + 8 < W ? x0 + 8 : W;
    const int ry[3] = {y0 == 0 ? 1 : 0, 0, y1 == H ? 1 : 0};
    const int rx[3] = {x0 == 0 ? 1 : 0, 0, x1 == W ? 1 : 0};
    const int my = (y1 - y0) - ry[0] - ry[2], mx = (x1 - x0) - rx[0] - rx[2];
#pragma unroll
    for (int cy = 0; cy < 3; ++cy)
#pragma unroll
      for (int cx = 0; cx < 3; ++cx) c[cy * 3 + cx] += (cy == 1 ? my : ry[cy]) * (cx == 1 ? mx : rx[cx]);
   }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k)
    if (c[k]) atomicAdd(&cnt[k], c[k]);
  __syncthreads();
  const int stat = tid >> 7, ch = tid & 127;
  double acc = 0.0;
#pragma unroll
  for (int cy = 0; cy < 3; ++cy)
#pragma unroll
    for (int cx = 0; cx < 3; ++cx) {
      const long long total = (long long)B * (cy == 1 ? H - 2 : 1) * (cx == 1 ? W - 2 : 1);
      const double n = (double)(total - cnt[cy * 3 + cx]);
      const double v = (double)ct_bf2f(ybg[(cy * 3 + cx) * CT_C + ch]);
      acc += n * (stat ? v * v : v);
    }
  const float hi = (float)acc;
  red[(long long)CT_NRED * 256 + tid] = hi;
  red[(long long)(CT_NRED + 1) * 256 + tid] = (float)(acc - (double)hi);
}

int gd_bn_fold_from_partials(hipStream_t st, const float* part, int nblk, int C, double count, const float* gamma,
                             const float* beta, double eps, double momentum, float* running_mean, float* running_var,
                             long long* num_batches, double* stats, float* ab, float* mv);

extern "C" size_t gdmae_conv3x3_tiles_workspace_bytes(int n_act) {
  return gd_align((size_t)(n_act > 0 ? n_act : 1) * 256 * sizeof(float)) + gd_align((size_t)(CT_NRED + 2) * 256 * sizeof(float));
}

// P / map / a / b / strides: HOST arrays over the k source stages (device pointers).  Yc (n_act * 64, 128) bf16 out.
// BatchNorm2d (training) of the conv output over all B*H*W sites: stats f64[256] = mean | rstd, ab f32[256] = folded
// scale | shift, mv f32[256] = mean | biased variance; running statistics updated when running_mean != NULL.
extern "C" int gdmae_conv3x3_tiles_fwd(const void* const* P, const int* const* maps, const float* const* a, const float* const* b,
                                       const int* strides, int k, const void* Wp, const void* ybg, const int* tile_list, int n_act,
                                       int B, int H, int W, void* Yc, const float* gamma, const float* beta, double eps,
                                       double momentum, float* running_mean, float* running_var, long long* num_batches,
                                       double* stats, float* ab, float* mv, void* workspace, void* stream) {
  GD_REQUIRE(k >= 1 && k <= CT_MAX_SRC, "conv3x3_tiles: 1..3 source stages");
  GD_REQUIRE(H >= 2 && W >= 2, "conv3x3_tiles: map too small");
  hipStream_t st = (hipStream_t)stream;
  CtArgs A;
  A.k = k;
  for (int g = 0; g < CT_MAX_SRC; ++g) {
    if (g < k) {
      const int s = strides[g];
      GD_REQUIRE((s == 1 || s == 2 || s == 4 || s == 8) && H % s == 0 && W % s == 0, "conv3x3_tiles: stride");
      A.src[g].P = (const unsigned short*)P[g];
      A.src[g].map = maps[g];
      A.src[g].a = a[g];
      A.src[g].b = b[g];
      A.src[g].ls = s == 1 ? 0 : (s == 2 ? 1 : (s == 4 ? 2 : 3));
    } else {
      A.src[g].P = nullptr; A.src[g].map = nullptr; A.src[g].a = nullptr; A.src[g].b = nullptr; A.src[g].ls = 0;
    }
  }
  A.Wp = (const uint4*)Wp;
  A.tile_list = tile_list;
  A.n_act = n_act;
  A.H = H; A.W = W; A.TH = (H + 7) / 8; A.TW = (W + 7) / 8;
  A.Yc = (unsigned short*)Yc;
  GdArena ar(workspace, gdmae_conv3x3_tiles_workspace_bytes(n_act));
  float* part = ar.take<float>((size_t)(n_act > 0 ? n_act : 1) * 256);
  float* red = ar.take<float>((size_t)(CT_NRED + 2) * 256);
  A.part = part;
  if (n_act > 0) {
    static bool once = false;
    if (!once) {
      GD_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, CT_LDS_BYTES));
      once = true;
    }
    hipLaunchKernelGGL(k_conv3x3_tiles, dim3((n_act + CT_TPW - 1) / CT_TPW), dim3(CT_THREADS), CT_LDS_BYTES, st, A);
    GD_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_ct_stats_reduce, dim3(CT_NRED + 1), dim3(256), 0, st, (const float*)part, n_act, tile_list,
                     (const unsigned short*)ybg, B, H, W, A.TH, A.TW, red);
  GD_LAUNCH_CHECK();
  return gd_bn_fold_from_partials(st, red, CT_NRED + 2, CT_C, (double)B * H * W, gamma, beta, eps, momentum, running_mean,
                                  running_var, num_batches, stats, ab, mv);
}

// ------------------------------------------------------------------------------------------------
// Readers of the tile-compact map
// ------------------------------------------------------------------------------------------------
// out[i] = Y row of cell[i] ((b * H + y) * W + x), 16-byte chunks; ES = element bytes
template <int ES>
__global__ __launch_bounds__(256) void k_tiles_gather_rows(const void* __restrict__ Yc, GdTiles T, const int* __restrict__ cell,
                                                           long long n, int H, int W, int C, uint4* __restrict__ out) {
  const int cv = C * ES / 16;
  const long long total = n * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    // (32-bit division where the index fits: the emulated 64-bit one costs ~100 instructions per 16 bytes)
    const long long r = total < (1ll << 31) ? (long long)((unsigned)i / (unsigned)cv) : i / cv;
    const int v = (int)(i - r * cv);
    const int s = cell[r];
    const int x = s % W, q = s / W, y = q % H, b = q / H;
    out[i] = ((const uint4*)gd_y_row<ES>(Yc, T, b, y, x, H, W, C))[v];
  }
}

extern "C" int gdmae_tiles_gather_rows(const void* Yc, const int* tile_slot, const void* ybg, const int* cell, long long n, int H,
                                       int W, int C, int elem_bytes, void* out, void* stream) {
  GD_REQUIRE(tile_slot != nullptr && (elem_bytes == 2 || elem_bytes == 4) && (C * elem_bytes) % 16 == 0, "tiles_gather_rows");
  if (n <= 0) return 0;
  GdTiles T{tile_slot, ybg, (H + 7) / 8, (W + 7) / 8};
  long long g = (n * (C * elem_bytes / 16) + 255) / 256;
  if (g > 16384) g = 16384;
  if (elem_bytes == 2) hipLaunchKernelGGL(k_tiles_gather_rows<2>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, Yc, T, cell, n, H, W, C, (uint4*)out);
  else hipLaunchKernelGGL(k_tiles_gather_rows<4>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, Yc, T, cell, n, H, W, C, (uint4*)out);
  GD_LAUNCH_CHECK();
  return 0;
}

// dense (B*H*W, C) copy of the tile-compact map (only for callers that ask for the reference's dense spatial_features)
template <int ES>
__global__ __launch_bounds__(256) void k_tiles_to_dense(const void* __restrict__ Yc, GdTiles T, int B, int H, int W, int C,
                                                        uint4* __restrict__ out) {
  const int cv = C * ES / 16;
  const long long total = (long long)B * H * W * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long s = i / cv;
    const int v = (int)(i % cv);
    const int x = (int)(s % W);
    const long long q = s / W;
    const int y = (int)(q % H), b = (int)(q / H);
    out[i] = ((const uint4*)gd_y_row<ES>(Yc, T, b, y, x, H, W, C))[v];
  }
}

// ... and the reference's dense spatial_features in ONE pass: out (B*H*W, C) fp32 = relu(a[c] * y + b[c]) with y read from the
// tile-compact bf16 map (active tiles) or the border-class constants (everything else).  Replaces the dense bf16 copy + a cast, a
// multiply, an add and a ReLU pass over 1.75 M sites x 128 channels (the drop-in default dense_spatial_features = True).
__global__ __launch_bounds__(256) void k_tiles_to_dense_affine_relu(const void* __restrict__ Yc, GdTiles T, int B, int H, int W, int C,
                                                                    const float* __restrict__ a, const float* __restrict__ bsh,
                                                                    float* __restrict__ out) {
  const int cv = C / 8;                                  // 8 channels per thread: one 16-byte load, two 16-byte stores
  const long long total = (long long)B * H * W * cv;
  // index arithmetic in 32 bits (the launcher checks B H W C / 8 < 2^31): three 64-bit divisions per element are emulated (~120
  // instructions) and made this 0.9 GB store stream VALU-bound (3.2 TB/s)
  const unsigned total32 = (unsigned)total, ucv = (unsigned)cv, uW = (unsigned)W, uH = (unsigned)H;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total32; i += gridDim.x * blockDim.x) {
    const unsigned s32 = i / ucv;
    const int v = (int)(i - s32 * ucv);
    const unsigned q32 = s32 / uW;
    const int x = (int)(s32 - q32 * uW);
    const unsigned b32 = q32 / uH;
    const int y = (int)(q32 - b32 * uH), b = (int)b32;
    const long long s = s32;
    const uint4 u = ((const uint4*)gd_y_row<2>(Yc, T, b, y, x, H, W, C))[v];
    float f[8];
    ct_unpack8(u, f);
    const float4 a0 = *(const float4*)(a + 8 * v), a1 = *(const float4*)(a + 8 * v + 4);
    const float4 b0 = *(const float4*)(bsh + 8 * v), b1 = *(const float4*)(bsh + 8 * v + 4);
    float4 o0, o1;
    o0.x = fmaxf(fmaf(a0.x, f[0], b0.x), 0.f); o0.y = fmaxf(fmaf(a0.y, f[1], b0.y), 0.f);
    o0.z = fmaxf(fmaf(a0.z, f[2], b0.z), 0.f); o0.w = fmaxf(fmaf(a0.w, f[3], b0.w), 0.f);
    o1.x = fmaxf(fmaf(a1.x, f[4], b1.x), 0.f); o1.y = fmaxf(fmaf(a1.y, f[5], b1.y), 0.f);
    o1.z = fmaxf(fmaf(a1.z, f[6], b1.z), 0.f); o1.w = fmaxf(fmaf(a1.w, f[7], b1.w), 0.f);
    typedef float f4v __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f4v{o0.x, o0.y, o0.z, o0.w}, (f4v*)(out + s * C + 8 * v));          // written once, read by whoever asked for it
    __builtin_nontemporal_store(f4v{o1.x, o1.y, o1.z, o1.w}, (f4v*)(out + s * C + 8 * v + 4));
  }
}

extern "C" int gdmae_tiles_to_dense_affine_relu(const void* Yc, const int* tile_slot, const void* ybg, int B, int H, int W, int C, const float* a,
                                                const float* b, float* out, void* stream) {
  GD_REQUIRE(tile_slot != nullptr && C % 8 == 0 && a != nullptr && b != nullptr, "tiles_to_dense_affine_relu");
  GD_REQUIRE((long long)B * H * W * (C / 8) < (1ll << 31), "tiles_to_dense_affine_relu: B H W C / 8 must fit 31 bits");
  GdTiles T{tile_slot, ybg, (H + 7) / 8, (W + 7) / 8};
  hipLaunchKernelGGL(k_tiles_to_dense_affine_relu, dim3(16384), dim3(256), 0, (hipStream_t)stream, Yc, T, B, H, W, C, a, b, out);
  GD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gdmae_tiles_to_dense(const void* Yc, const int* tile_slot, const void* ybg, int B, int H, int W, int C, int elem_bytes,
                                    void* out, void* stream) {
  GD_REQUIRE(tile_slot != nullptr && (elem_bytes == 2 || elem_bytes == 4) && (C * elem_bytes) % 16 == 0, "tiles_to_dense");
  GdTiles T{tile_slot, ybg, (H + 7) / 8, (W + 7) / 8};
  if (elem_bytes == 2) hipLaunchKernelGGL(k_tiles_to_dense<2>, dim3(16384), dim3(256), 0, (hipStream_t)stream, Yc, T, B, H, W, C, (uint4*)out);
  else hipLaunchKernelGGL(k_tiles_to_dense<4>, dim3(16384), dim3(256), 0, (hipStream_t)stream, Yc, T, B, H, W, C, (uint4*)out);
  GD_LAUNCH_CHECK();
  return 0;
}
