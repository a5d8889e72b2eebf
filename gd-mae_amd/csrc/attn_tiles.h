// Shared pieces of the workgroup-cooperative bf16 attention kernels (attention_coop.hip): operand rows, swizzled row-major LDS tiles,
// MFMA wrappers (the operand scheme of attention_t32.hip) and the COOPERATIVE row transfer - a workgroup moves the rows of a window as
// whole 64 ... 256-byte segments, 16 bytes per lane, through the LDS tiles the products read anyway.
//
// Why: the per-(window, head) kernels load a token row as DH / 8 8-byte pieces per lane (the MFMA operand layout), i.e. every load
// instruction of a wavefront touches 32 different cache lines and uses 16 bytes of each.  With the arithmetic compiled out the T = 64
// forward still took 17.6 of its 19.1 us and the backward 30 of 39 us (2.9 - 3.0 TB/s, profiles/r05_attention_ablation.txt): the launches
// were bound by the ADDRESS rate of that pattern, not by latency or bytes.  Here a row segment of all heads of the workgroup is one
// contiguous run of 16-byte lanes (whole cache lines), loads and stores alike.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace attn {
constexpr float kInvEpsNorm = 1e12f;       // 1 / 1e-12 (F.normalize eps)
constexpr float kPadKey = -1e15f;          // normalised dot product of a padded key
constexpr float kLog2e = 1.4426950408889634f;
constexpr int kPitch = 32;                 // LDS tile row pitch in bf16 elements (64 B)

template <int NPC>
struct Row {            // this lane's 8-byte pieces of one token row (MFMA operand layout)
  uint2 p[NPC];
};

__device__ __forceinline__ int c_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ float lo_f(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_f(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ float half_sum(float x) {      // lane (rho, 0) + lane (rho, 1)
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float half_max(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// sum / max over the 4 lane groups (lane >> 4) that share a column (lane & 15) of a 16 x 16 tile
__device__ __forceinline__ float grp_sum(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float grp_max(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

// Tiles are row-major [token][32 bf16]; the eight 8-byte units of a row are XOR-permuted by a function of the row (as attention_t32.hip:
// conflict-free ds_write_b64 / ds_read_b64 of the operand pieces, transposed reads touch whole rows).  swz() is that function.
__device__ __forceinline__ int swz(int row) { return (((row >> 1) ^ (row >> 4)) & 1) | (((row >> 2) & 3) << 1); }
__device__ __forceinline__ int unit_off(int row, int unit) { return row * kPitch + 4 * (unit ^ swz(row)); }

template <int NPC>
__device__ __forceinline__ Row<NPC> lds_row(const unsigned short* __restrict__ tile, int row, int h) {
  Row<NPC> r;
#pragma unroll
  for (int t = 0; t < NPC; ++t) r.p[t] = *reinterpret_cast<const uint2*>(tile + unit_off(row, 2 * t + h));
  return r;
}
template <int NPC>
__device__ __forceinline__ void store_tile(unsigned short* __restrict__ tile, int row, int h, const Row<NPC>& r) {
#pragma unroll
  for (int t = 0; t < NPC; ++t) *reinterpret_cast<uint2*>(tile + unit_off(row, 2 * t + h)) = r.p[t];
}
template <int NPC>
__device__ __forceinline__ bf16x8 step_frag(const Row<NPC>& r, int s) {
  const uint4 u = make_uint4(r.p[2 * s].x, r.p[2 * s].y, r.p[2 * s + 1].x, r.p[2 * s + 1].y);
  return __builtin_bit_cast(bf16x8, u);
}
// acc += a-rows . b-rows^T over the head dim (32 x 32 tile)
template <int NPC>
__device__ __forceinline__ f32x16 mma_rows(const Row<NPC>& a, const Row<NPC>& b, f32x16 acc) {
#pragma unroll
  for (int s = 0; s < NPC / 2; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(step_frag<NPC>(a, s), step_frag<NPC>(b, s), acc, 0, 0, 0);
  return acc;
}
// A operand of a token-contracted step over tokens [base16, base16 + 16) of a row-major tile: row = dh (lane & 31)
__device__ __forceinline__ bf16x8 tr_a(const unsigned short* __restrict__ tile, int base16, int lane) {
  const int i = lane & 15, grp = (lane >> 4) & 1, h = lane >> 5;
  const int row = base16 + 4 * h + (i >> 2), unit = 4 * grp + (i & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + unit_off(row, unit)));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(tile + unit_off(row + 8, unit)));
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& a, int r0) {
  f32x8 t;
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = a[r0 + j];
  return __builtin_convertvector(t, bf16x8);
}
// out^T[dh][column] += sum over the 32 tokens of `tile32` rows: tile^T . b
__device__ __forceinline__ f32x16 mma_tokens(const unsigned short* __restrict__ tile32, const f32x16& b, f32x16 acc, int lane) {
#pragma unroll
  for (int t = 0; t < 2; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_a(tile32, 16 * t, lane), pack8(b, 8 * t), acc, 0, 0, 0);
  return acc;
}
// per-row scalars of one 32-token tile in C-layout order: x[r] = s[c_row(r, h)]
__device__ __forceinline__ void row_scalars(const float* __restrict__ s, int h, float (&x)[16]) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float4 a = *reinterpret_cast<const float4*>(s + 8 * t + 4 * h);
    x[4 * t] = a.x; x[4 * t + 1] = a.y; x[4 * t + 2] = a.z; x[4 * t + 3] = a.w;
  }
}
__device__ __forceinline__ uint2 pack_piece(float a, float b, float c, float d) {
  f32x4 t = {a, b, c, d};
  const bf16x4 o = __builtin_convertvector(t, bf16x4);
  return __builtin_bit_cast(uint2, o);
}
__device__ __forceinline__ void piece_f32(const uint2& w, float (&x)[4]) {
  x[0] = lo_f(w.x); x[1] = hi_f(w.x); x[2] = lo_f(w.y); x[3] = hi_f(w.y);
}
__device__ __forceinline__ f32x16 splat(float x) {
  f32x16 a;
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = x;
  return a;
}

// ---------------------------------------------------------------------------------------------------------------------
// Cooperative row transfer.  A workgroup of 256 threads owns ROWS window slots x HW heads; a row's HW * DH elements are one contiguous
// segment of HW * DH * 2 bytes = LPR 16-byte lanes.  Thread tid takes, in pass p, row p * RPI + tid / LPR, chunk tid % LPR
// (head tid % LPR / CPH, 16-byte piece % CPH of that head's DH elements).
// ---------------------------------------------------------------------------------------------------------------------
template <int DH, int HW, int ROWS>
struct Coop {
  static constexpr int NPC = DH / 8;
  static constexpr int LPR = HW * DH * 2 / 16;      // lanes per row
  static constexpr int CPH = DH / 8;                // 16-byte chunks per head
  static constexpr int RPI = 256 / LPR;             // rows per pass
  static constexpr int P = ROWS / RPI > 0 ? ROWS / RPI : 1;
  static_assert(LPR <= 256 && 256 % LPR == 0, "cooperative pass");
};
// 16-byte slot of chunk cq in the swizzled tile row: the pair of 8-byte units (2 cq, 2 cq + 1) stays one 16-byte slot under the
// XOR permutation; its halves trade places when the permutation's low bit is set
__device__ __forceinline__ int slot16_off(int row, int cq) { return row * kPitch + 8 * (cq ^ (swz(row) >> 1)); }
__device__ __forceinline__ uint4 swap_halves(uint4 v, int row) {
  const bool s = swz(row) & 1;
  return make_uint4(s ? v.z : v.x, s ? v.w : v.y, s ? v.x : v.z, s ? v.y : v.w);
}
__device__ __forceinline__ void tile_put16(unsigned short* __restrict__ tile, int row, int cq, uint4 v) {
  *reinterpret_cast<uint4*>(tile + slot16_off(row, cq)) = swap_halves(v, row);
}
__device__ __forceinline__ uint4 tile_get16(const unsigned short* __restrict__ tile, int row, int cq) {
  return swap_halves(*reinterpret_cast<const uint4*>(tile + slot16_off(row, cq)), row);
}
__device__ __forceinline__ uint4 and16(uint4 v, unsigned m) { return make_uint4(v.x & m, v.y & m, v.z & m, v.w & m); }
// sum of squares of the 8 bf16 values of a chunk
__device__ __forceinline__ float ssq16(uint4 v) {
  float ss = 0.f;
  const bf16x2 a = __builtin_bit_cast(bf16x2, v.x), b = __builtin_bit_cast(bf16x2, v.y), c = __builtin_bit_cast(bf16x2, v.z),
               d = __builtin_bit_cast(bf16x2, v.w);
  ss = __builtin_amdgcn_fdot2_f32_bf16(a, a, ss, false);
  ss = __builtin_amdgcn_fdot2_f32_bf16(b, b, ss, false);
  ss = __builtin_amdgcn_fdot2_f32_bf16(c, c, ss, false);
  ss = __builtin_amdgcn_fdot2_f32_bf16(d, d, ss, false);
  return ss;
}
// dot product of the 8 bf16 pairs of two chunks
__device__ __forceinline__ float dot16(uint4 a, uint4 b) {
  float ss = 0.f;
  ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.x), __builtin_bit_cast(bf16x2, b.x), ss, false);
  ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.y), __builtin_bit_cast(bf16x2, b.y), ss, false);
  ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.z), __builtin_bit_cast(bf16x2, b.z), ss, false);
  ss = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.w), __builtin_bit_cast(bf16x2, b.w), ss, false);
  return ss;
}
// 1 / max(|row|, 1e-12) from the per-chunk sums of squares: the CPH (2 or 4) chunks of a head are adjacent lanes
template <int CPH>
__device__ __forceinline__ float inv_norm_chunks(float ss) {
  ss += gd_dpp_mov<0xB1>(ss);                       // lane ^ 1
  if (CPH >= 4) ss += gd_dpp_mov<0x4E>(ss);         // lane ^ 2
  return fminf(__builtin_amdgcn_rsqf(ss), kInvEpsNorm);
}
}  // namespace attn

// ---------------------------------------------------------------------------------------------------------------------
// 16 x 16 tiles (the sparse occupancy level): lane l = (c, g) = (l & 15, l >> 4) holds of token row c the 8-byte pieces
// dh = 16 p + 4 g + {0..3}, p < DH / 16 (unit 4 p + g of the row); same row-major tiles, their own bank permutation (attention_t16.hip)
// ---------------------------------------------------------------------------------------------------------------------
namespace attn16 {
using attn::kPitch;
constexpr int kTile = 16 * kPitch;         // one [token][dh] tile, in elements
template <int NP>
struct Row {
  uint2 p[NP];
};
__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1) | (((row >> 2) & 1) << 2); }
__device__ __forceinline__ int unit_off(int row, int unit) { return row * kPitch + 4 * (unit ^ swz(row)); }
template <int NP>
__device__ __forceinline__ Row<NP> lds_row(const unsigned short* __restrict__ tile, int c, int g) {
  Row<NP> r;
#pragma unroll
  for (int p = 0; p < NP; ++p) r.p[p] = *reinterpret_cast<const uint2*>(tile + unit_off(c, 4 * p + g));
  return r;
}
template <int NP>
__device__ __forceinline__ void store_tile(unsigned short* __restrict__ tile, int c, int g, const Row<NP>& r) {
#pragma unroll
  for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(tile + unit_off(c, 4 * p + g)) = r.p[p];
}
// dh-contracted product of two token tiles: D[row of a][row of b]
template <int NP>
__device__ __forceinline__ f32x4 mma_rows(const Row<NP>& a, const Row<NP>& b) {
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  if constexpr (NP == 2) {
    const uint4 ua = make_uint4(a.p[0].x, a.p[0].y, a.p[1].x, a.p[1].y), ub = make_uint4(b.p[0].x, b.p[0].y, b.p[1].x, b.p[1].y);
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ua), __builtin_bit_cast(bf16x8, ub), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a.p[0]), __builtin_bit_cast(s16x4, b.p[0]), c, 0, 0, 0);
  }
}
// token-contracted product: tile^T[dh 16 p + (column of the lane)][token] . b[token][column c]; A read transposed from LDS
__device__ __forceinline__ f32x4 mma_tokens(const unsigned short* __restrict__ tile, int p, int c, int g, f32x4 b) {
  const unsigned short* src = tile + unit_off(4 * g + (c >> 2), 4 * p + (c & 3));
  const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)src);
  const bf16x4 bb = __builtin_convertvector(b, bf16x4);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(s16x4, bb), acc, 0, 0, 0);
}
// 16-byte chunk cq (elements 8 cq ...) of a row = units (2 cq, 2 cq + 1)
__device__ __forceinline__ int slot16_off(int row, int cq) { return row * kPitch + 8 * (cq ^ (swz(row) >> 1)); }
__device__ __forceinline__ uint4 swap_halves(uint4 v, int row) {
  const bool s = swz(row) & 1;
  return make_uint4(s ? v.z : v.x, s ? v.w : v.y, s ? v.x : v.z, s ? v.y : v.w);
}
__device__ __forceinline__ void tile_put16(unsigned short* __restrict__ tile, int row, int cq, uint4 v) {
  *reinterpret_cast<uint4*>(tile + slot16_off(row, cq)) = swap_halves(v, row);
}
__device__ __forceinline__ uint4 tile_get16(const unsigned short* __restrict__ tile, int row, int cq) {
  return swap_halves(*reinterpret_cast<const uint4*>(tile + slot16_off(row, cq)), row);
}
}  // namespace attn16
