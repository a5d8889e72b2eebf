// Weight gradients of one encoder layer as ONE launch (the backward of the nn.Linear layers of
// sst_basic_block.py:57-84 and of the cosine_msa.py in- / out-projections):
//
//     dW_g (M_g, N_g) = G_g^T (M_g, rows) X_g (rows, N_g)        g = W2, W1, Wo, Wqk, Wv      rows = 20-50 k tokens
//     db_g (M_g)      = column sums of G_g                        for the layers whose bias gradient is not a by-product
//                                                                 of a LayerNorm backward (W1, Wqk, Wv)
//
// G / X are bf16 row-major, so the contraction runs over the slow axis of both operands: an MFMA fragment needs 8
// consecutive rows of one column.  Chunks of 64 rows are staged row-major in LDS with 16-byte stores and the fragments are
// read transposed (ds_read_b64_tr_b16); the 64-byte segments of a staged row are XOR-permuted with (row & 3), which makes
// both the 16-byte stores and the transposed reads bank-conflict free at a 256-byte pitch.
//
// Work decomposition: the rows are cut into S slices; a workgroup (4 wavefronts = 2 x 2 blocks of 64 x 64) owns one
// 128 x 128 output tile of one matrix for one slice, loops over the slice's chunks (next chunk prefetched into registers
// behind the MFMAs) and writes its fp32 partial tile; partial tiles are summed in a fixed order by the caller's reduce
// kernel (deterministic).  All tiles of a slice - every matrix of the layer - are placed on ONE XCD (blockIdx % 8), so
// the 2-4x re-reads of a slice's activations by the tiles that share them are served by that XCD's L2.
//
// Load discipline (round 4, what the ISA said - DESIGN 9): the kernel is templated on the X addressing mode and every load is
// unconditional (clamped indices, masks applied when a chunk is staged), so that no control-flow path joins another with loads in
// flight: the compiler's wait insertion answered every such join with `s_waitcnt vmcnt(0)`, i.e. it drained the loads of chunk c + 2
// before the products of chunk c (132 -> 92 us for the five gradients of a d = 256 layer at 49 k rows).
#include "common.h"
#include "dw_grouped.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int kTile = 128;
constexpr int kChunk = 64;                       // rows per LDS stage
constexpr int kRowBytes = kTile * 2;             // 256
constexpr int kStage = kChunk * kRowBytes;       // one operand chunk: 16 KB

// byte offset of the 16-byte unit `u16` (0 ... 15) of staged row `row`
__device__ __forceinline__ int stage_off(int row, int u16) { return row * kRowBytes + (((u16 >> 2) ^ (row & 3)) << 6) + ((u16 & 3) << 4); }

struct Chunk {          // this thread's four 16-byte pieces of a staged chunk (rows r, r + 16, r + 32, r + 48)
  uint4 q0, q1, q2, q3;
};
// p: the operand's base + this thread's column; rows r, r + 16, r + 32, r + 48, clamped to `last` (the highest row that may be read;
// rows past it repeat it and are cleared by mask_chunk before they are staged).  Branch-free: a branch around loads makes the
// compiler drain the load counter where the paths join, which serialises the loads of chunk c + 2 with the products of chunk c.
__device__ __forceinline__ Chunk load_chunk(const unsigned short* __restrict__ p, int ld, unsigned r, unsigned last) {
  Chunk k;
  k.q0 = *reinterpret_cast<const uint4*>(p + (unsigned long long)(r < last ? r : last) * (unsigned)ld);
  k.q1 = *reinterpret_cast<const uint4*>(p + (unsigned long long)(r + 16 < last ? r + 16 : last) * (unsigned)ld);
  k.q2 = *reinterpret_cast<const uint4*>(p + (unsigned long long)(r + 32 < last ? r + 32 : last) * (unsigned)ld);
  k.q3 = *reinterpret_cast<const uint4*>(p + (unsigned long long)(r + 48 < last ? r + 48 : last) * (unsigned)ld);
  return k;
}
__device__ __forceinline__ void zero_chunk_unless(Chunk& k, unsigned keep_mask) {      // keep_mask: ~0u or 0 (an AND, not a select of structs)
  k.q0.x &= keep_mask; k.q0.y &= keep_mask; k.q0.z &= keep_mask; k.q0.w &= keep_mask;
  k.q1.x &= keep_mask; k.q1.y &= keep_mask; k.q1.z &= keep_mask; k.q1.w &= keep_mask;
  k.q2.x &= keep_mask; k.q2.y &= keep_mask; k.q2.z &= keep_mask; k.q2.w &= keep_mask;
  k.q3.x &= keep_mask; k.q3.y &= keep_mask; k.q3.z &= keep_mask; k.q3.w &= keep_mask;
}
// X rows through an index (rulebook column of a sparse convolution); rows >= n_valid are not looked up.  Branch-free: the four
// indices of a thread's pieces are loaded first, then the four rows (row 0 for a missing tap, cleared afterwards), so a chunk costs
// two round trips instead of eight serialised ones.
__device__ __forceinline__ uint4 cvt_f32x8(const float4& a, const float4& b) {
  uint4 q;
  q.x = gd_pack_bf16(a.x, a.y); q.y = gd_pack_bf16(a.z, a.w);
  q.z = gd_pack_bf16(b.x, b.y); q.w = gd_pack_bf16(b.z, b.w);
  return q;
}
struct ChunkIdx {
  int j0, j1, j2, j3;
};
__device__ __forceinline__ ChunkIdx load_chunk_idx(const int* __restrict__ xidx, int stride, unsigned row0, unsigned n_valid, int tid) {
  const unsigned r = row0 + (tid >> 4), lastv = n_valid - 1;
  auto at = [&](unsigned rr) { return xidx[(unsigned long long)(rr < lastv ? rr : lastv) * (unsigned)stride]; };
  // rows >= n_valid read the index of row n_valid - 1 (a valid address); their data is cleared by mask_chunk when the chunk is
  // staged - no select here, which would wait for the index loads where they are issued
  ChunkIdx k;
  k.j0 = at(r); k.j1 = at(r + 16); k.j2 = at(r + 32); k.j3 = at(r + 48);
  return k;
}
template <bool F32>
__device__ __forceinline__ Chunk load_chunk_rows(const void* __restrict__ base, int ld, const ChunkIdx& I, int col0, int tid) {
  const int col = col0 + (tid & 15) * 8;
  const long long o0 = (long long)(I.j0 < 0 ? 0 : I.j0) * ld + col, o1 = (long long)(I.j1 < 0 ? 0 : I.j1) * ld + col;
  const long long o2 = (long long)(I.j2 < 0 ? 0 : I.j2) * ld + col, o3 = (long long)(I.j3 < 0 ? 0 : I.j3) * ld + col;
  Chunk k;
  if (F32) {
    const float* f = (const float*)base;
    const float4 a0 = *reinterpret_cast<const float4*>(f + o0), b0 = *reinterpret_cast<const float4*>(f + o0 + 4);
    const float4 a1 = *reinterpret_cast<const float4*>(f + o1), b1 = *reinterpret_cast<const float4*>(f + o1 + 4);
    const float4 a2 = *reinterpret_cast<const float4*>(f + o2), b2 = *reinterpret_cast<const float4*>(f + o2 + 4);
    const float4 a3 = *reinterpret_cast<const float4*>(f + o3), b3 = *reinterpret_cast<const float4*>(f + o3 + 4);
    k.q0 = cvt_f32x8(a0, b0); k.q1 = cvt_f32x8(a1, b1); k.q2 = cvt_f32x8(a2, b2); k.q3 = cvt_f32x8(a3, b3);
  } else {
    const unsigned short* h = (const unsigned short*)base;
    k.q0 = *reinterpret_cast<const uint4*>(h + o0);
    k.q1 = *reinterpret_cast<const uint4*>(h + o1);
    k.q2 = *reinterpret_cast<const uint4*>(h + o2);
    k.q3 = *reinterpret_cast<const uint4*>(h + o3);
  }
  return k;       // the pieces of missing taps (index < 0) hold row 0: cleared by zero_missing when the chunk is staged, not here
                  // (a select on the loaded registers would wait for the loads right where they are issued)
}
// bit k set: piece k of the chunk belongs to a missing tap
__device__ __forceinline__ unsigned missing_bits(const ChunkIdx& I) {
  return (I.j0 < 0 ? 1u : 0u) | (I.j1 < 0 ? 2u : 0u) | (I.j2 < 0 ? 4u : 0u) | (I.j3 < 0 ? 8u : 0u);
}
__device__ __forceinline__ void zero_missing(Chunk& k, unsigned bits) {
  const unsigned m0 = (bits & 1u) ? 0u : ~0u, m1 = (bits & 2u) ? 0u : ~0u, m2 = (bits & 4u) ? 0u : ~0u, m3 = (bits & 8u) ? 0u : ~0u;
  k.q0.x &= m0; k.q0.y &= m0; k.q0.z &= m0; k.q0.w &= m0;
  k.q1.x &= m1; k.q1.y &= m1; k.q1.z &= m1; k.q1.w &= m1;
  k.q2.x &= m2; k.q2.y &= m2; k.q2.z &= m2; k.q2.w &= m2;
  k.q3.x &= m3; k.q3.y &= m3; k.q3.z &= m3; k.q3.w &= m3;
}
// rows >= n_valid (the padding of the row count to the slice grid) contribute nothing, whatever the buffers hold there:
// the chunk(s) that reach past n_valid are cleared in registers before they are staged (uniform branch per chunk)
__device__ __forceinline__ void mask_chunk(Chunk& k, long long row0, int tid, long long n_valid) {
  if (row0 + kChunk <= n_valid) return;
  const int r = tid >> 4;
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  if (row0 + r >= n_valid) k.q0 = z;
  if (row0 + r + 16 >= n_valid) k.q1 = z;
  if (row0 + r + 32 >= n_valid) k.q2 = z;
  if (row0 + r + 48 >= n_valid) k.q3 = z;
}
__device__ __forceinline__ void store_chunk(unsigned char* __restrict__ lds, int tid, const Chunk& k) {
  const int c = tid & 15, r = tid >> 4;
  unsigned char* p = lds + stage_off(r, c);            // (row & 3) is the same for r, r + 16, ...
  *reinterpret_cast<uint4*>(p) = k.q0;
  *reinterpret_cast<uint4*>(p + 16 * kRowBytes) = k.q1;
  *reinterpret_cast<uint4*>(p + 32 * kRowBytes) = k.q2;
  *reinterpret_cast<uint4*>(p + 48 * kRowBytes) = k.q3;
}
__device__ __forceinline__ void add_bf16x8(const uint4& q, float (&cs)[8]) {
  cs[0] += __uint_as_float(q.x << 16); cs[1] += __uint_as_float(q.x & 0xFFFF0000u);
  cs[2] += __uint_as_float(q.y << 16); cs[3] += __uint_as_float(q.y & 0xFFFF0000u);
  cs[4] += __uint_as_float(q.z << 16); cs[5] += __uint_as_float(q.z & 0xFFFF0000u);
  cs[6] += __uint_as_float(q.w << 16); cs[7] += __uint_as_float(q.w & 0xFFFF0000u);
}
// operand fragment: column (col32 + (lane & 31)) of the staged chunk, rows k0 + 8 h + {0..7}
__device__ __forceinline__ bf16x8 frag(const unsigned char* __restrict__ lds, int k0, int col32, int lane) {
  const int i = lane & 15, grp = (lane >> 4) & 1, h = lane >> 5;
  const int row = k0 + 8 * h + (i >> 2);
  const int seg = col32 >> 5;                                      // 64-byte segment of the 32 columns
  const unsigned char* p0 = lds + row * kRowBytes + ((seg ^ (row & 3)) << 6) + 32 * grp + 8 * (i & 3);
  const int row1 = row + 4;
  const unsigned char* p1 = lds + row1 * kRowBytes + ((seg ^ (row1 & 3)) << 6) + 32 * grp + 8 * (i & 3);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p1);
  const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
  return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
}

// XM: 0 = X rows in place, 1 = X rows through an index (bf16), 2 = through an index, fp32 rows rounded on load
template <int XM>
__global__ __launch_bounds__(256, 2) void k_dw_grouped(GdDwGroup A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];        // [buffer][G | X][kStage]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // slice s lives on XCD s % 8: blockIdx = xcd + 8 * (tile + tiles_total * (s / 8))
  const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
  const int sg = pos / A.tiles_total, t = pos - sg * A.tiles_total;
  const int s = sg * 8 + xcd;
  // uniform selects of the job fields (no dynamic indexing of the kernel argument, no struct temporaries)
  const unsigned short* G = (const unsigned short*)A.job[0].G;
  const unsigned short* X = (const unsigned short*)A.job[0].X;
  float* part = A.job[0].part;
  float* colpart = A.job[0].colpart;
  int M = A.job[0].M, N = A.job[0].N, tile0 = 0;
  const int* xidx = A.job[0].xidx;
  int xstride = A.job[0].xidx_stride, x_f32 = A.job[0].x_f32;
  int g_ld = A.job[0].g_ld, g_cols = A.job[0].g_cols;
#pragma unroll
  for (int j = 1; j < GD_DW_MAX_JOBS; ++j)
    if (j < A.n_jobs && t >= A.job[j].tile0) {
      G = (const unsigned short*)A.job[j].G;
      X = (const unsigned short*)A.job[j].X;
      part = A.job[j].part;
      colpart = A.job[j].colpart;
      M = A.job[j].M;
      N = A.job[j].N;
      tile0 = A.job[j].tile0;
      xidx = A.job[j].xidx;
      xstride = A.job[j].xidx_stride;
      x_f32 = A.job[j].x_f32;
      g_ld = A.job[j].g_ld;
      g_cols = A.job[j].g_cols;
    }
  if (g_ld == 0) g_ld = M;
  if (g_cols == 0) g_cols = M;
  const int tn_count = N / kTile;
  const int tm = (t - tile0) / tn_count, tn = (t - tile0) - tm * tn_count;
  const long long r0 = (long long)s * A.rows_per_slice;
  const int nchunk = (int)(A.rows_per_slice / kChunk);
  const int wm = wv >> 1, wn = wv & 1;                 // this wavefront's 64 x 64 block
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;
  const bool want_cs = colpart != nullptr && tn == 0;

  // two chunks in flight in registers (the loads of chunk c + 2 are issued behind the barrier of chunk c: ~2 compute phases of
  // latency cover per load), two LDS buffers
  const int gcol = tm * kTile, xcol = tn * kTile;
  const unsigned last = A.guard_rows ? (unsigned)(A.n_valid - 1) : 0xFFFFFFFFu;
  const unsigned nv = (unsigned)A.n_valid, rb = (unsigned)r0, rt = (unsigned)r0 + (tid >> 4);
  const int c8 = (tid & 15) * 8;
  const bool gkeep = gcol + c8 < g_cols;                  // columns past the operand's width (narrow G) are staged as zeros
  const unsigned short* gp = G + (gkeep ? gcol + c8 : 0);
  const unsigned gmask = gkeep ? 0xFFFFFFFFu : 0u;
  const unsigned short* xp = X + xcol + c8;
  // gathered X: the indices of chunk c + 4 are requested when the rows of chunk c + 2 are (one more stage of look-ahead for the
  // dependent load); i0 / i1 hold the indices of the next even / odd chunk to fetch
  ChunkIdx i0 = {-1, -1, -1, -1}, i1 = {-1, -1, -1, -1};
  unsigned miss0 = 0u, miss1 = 0u;                         // missing-tap bits of the chunks held in x0 / x1
  auto load_x = [&](unsigned chunk, ChunkIdx& I, unsigned& miss) {
    if constexpr (XM == 0) {
      return load_chunk(xp, N, rt + chunk * kChunk, last);
    } else {
      const Chunk k = load_chunk_rows<XM == 2>(X, N, I, xcol, tid);
      miss = missing_bits(I);
      I = load_chunk_idx(xidx, xstride, rb + (chunk + 2) * kChunk, nv, tid);      // for this buffer's next chunk
      return k;
    }
  };
  auto load_g = [&](unsigned chunk) { return load_chunk(gp, g_ld, rt + chunk * kChunk, last); };
  if constexpr (XM != 0) {
    i0 = load_chunk_idx(xidx, xstride, rb, nv, tid);
    i1 = load_chunk_idx(xidx, xstride, rb + kChunk, nv, tid);
  }
  auto compute = [&](const unsigned char* bg, const unsigned char* bx) {
#pragma unroll
    for (int ks = 0; ks < kChunk / 16; ++ks) {
      const bf16x8 a0 = frag(bg, 16 * ks, wm * 64, lane), a1 = frag(bg, 16 * ks, wm * 64 + 32, lane);
      const bf16x8 b0 = frag(bx, 16 * ks, wn * 64, lane), b1 = frag(bx, 16 * ks, wn * 64 + 32, lane);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
  };
  unsigned char* bufg0 = lds;
  unsigned char* bufx0 = lds + kStage;
  unsigned char* bufg1 = lds + 2 * kStage;
  unsigned char* bufx1 = lds + 3 * kStage;
  const long long r_end = r0 + A.rows_per_slice < A.n_valid ? r0 + A.rows_per_slice : A.n_valid;
  // issue order pinned (chunk 0 before chunk 1): the wait for a chunk is counted in loads issued after it, and the loop's first
  // phase may only wait for all but the 8 newest if that holds on the way in as well as around the loop
  Chunk g0 = load_g(0), x0 = load_x(0, i0, miss0);
  __builtin_amdgcn_sched_barrier(0);
  const unsigned c1 = nchunk > 1 ? 1u : 0u;
  Chunk g1 = load_g(c1), x1 = load_x(c1, i1, miss1);
  __builtin_amdgcn_sched_barrier(0);
  // Both phases of an iteration run unconditionally and every load is issued unconditionally (chunk index clamped to the slice's
  // last chunk): a path that skips loads would join one that issued them, and the compiler then drains the load counter at the
  // join - which serialises the loads of chunk c + 2 with the products of chunk c (49 k rows, d = 256: 132 us with the drains, 92 without).
  // An odd chunk count ends with one all-zero phase (rows >= the slice's end are cleared like the rows >= n_valid).
  for (int c = 0; c < nchunk; c += 2) {
    mask_chunk(g0, r0 + (long long)c * kChunk, tid, r_end);
    mask_chunk(x0, r0 + (long long)c * kChunk, tid, r_end);
    zero_chunk_unless(g0, gmask);
    if constexpr (XM != 0) zero_missing(x0, miss0);
    store_chunk(bufg0, tid, g0);
    store_chunk(bufx0, tid, x0);
    if (want_cs) {
      add_bf16x8(g0.q0, cs); add_bf16x8(g0.q1, cs); add_bf16x8(g0.q2, cs); add_bf16x8(g0.q3, cs);
    }
    __syncthreads();          // chunk c staged; buffer 0 was last read two phases ago, before the previous barrier
    {
      const unsigned cn = c + 2 < nchunk ? c + 2 : nchunk - 1;
      g0 = load_g(cn);
      x0 = load_x(cn, i0, miss0);
    }
    __builtin_amdgcn_sched_barrier(0);      // the loads go out BEFORE the products (two phases of cover)
    compute(bufg0, bufx0);
    mask_chunk(g1, r0 + (long long)(c + 1) * kChunk, tid, r_end);
    mask_chunk(x1, r0 + (long long)(c + 1) * kChunk, tid, r_end);
    zero_chunk_unless(g1, gmask);
    if constexpr (XM != 0) zero_missing(x1, miss1);
    store_chunk(bufg1, tid, g1);
    store_chunk(bufx1, tid, x1);
    if (want_cs) {
      add_bf16x8(g1.q0, cs); add_bf16x8(g1.q1, cs); add_bf16x8(g1.q2, cs); add_bf16x8(g1.q3, cs);
    }
    __syncthreads();
    {
      const unsigned cn = c + 3 < nchunk ? c + 3 : nchunk - 1;
      g1 = load_g(cn);
      x1 = load_x(cn, i1, miss1);
    }
    __builtin_amdgcn_sched_barrier(0);
    compute(bufg1, bufx1);
  }
  // ---- partial tile: D[row = m by register, column = n by lane]
  float* out = part + ((long long)s * M + tm * kTile + wm * 64) * N + tn * kTile + wn * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        out[(long long)m * N + j * 32 + (lane & 31)] = acc[i][j][e];
      }
  if (want_cs) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);       // (16 row groups, 128 columns)
    const int c = tid & 15, r = tid >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) red[r * kTile + c * 8 + j] = cs[j];
    __syncthreads();
    if (tid < kTile) {
      float tsum = 0.f;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) tsum += red[rr * kTile + tid];
      colpart[(long long)s * M + tm * kTile + tid] = tsum;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// 128 x 256 pair tiles (round 6; opt-in, GDMAE_DW_PAIR=1 - measured slower, see gd_dw_grouped_s).  k_dw_grouped is bound by its L2 -> CU load stream (probe at 49 k rows, d = 256: loads alone 76 of
// 92 us, products alone 51): a 128 x 128 tile requests 32 KB per 64-row chunk for 16 MFMAs per wavefront.  Here a workgroup of EIGHT
// wavefronts stages ONE G chunk for TWO X sub-tiles - adjacent column blocks of the job (pair_mode 1) or the same column block of
// two taps of a gathered launch (pair_mode 2: the nine tap jobs read the same G) - 48 KB per chunk for twice the products, i.e. 0.75
// of the bytes per product; wavefronts 0 - 3 / 4 - 7 own the 2 x 2 blocks of sub-tile 0 / 1 exactly as the four wavefronts of
// k_dw_grouped own theirs (same fragments, same k order: bit-identical partial tiles; the optional column sums of G are taken by 32
// instead of 16 row groups, another fixed order).  512 threads move a chunk (two 16-byte
// pieces of G and of each X sub-tile per thread), two chunks in flight in registers, two LDS buffers of 48 KB: one workgroup per
// CU at the two wavefronts per SIMD of the four-wavefront kernel.
// ---------------------------------------------------------------------------------------------------------------------
struct Chunk2 {          // rows r, r + 32 of a staged chunk (r = tid >> 4 of 512 threads)
  uint4 q0, q1;
};
__device__ __forceinline__ Chunk2 load_chunk2(const unsigned short* __restrict__ p, int ld, unsigned r, unsigned last) {
  Chunk2 k;
  k.q0 = *reinterpret_cast<const uint4*>(p + (unsigned long long)(r < last ? r : last) * (unsigned)ld);
  k.q1 = *reinterpret_cast<const uint4*>(p + (unsigned long long)(r + 32 < last ? r + 32 : last) * (unsigned)ld);
  return k;
}
struct ChunkIdx2 {
  int j0, j1;
};
__device__ __forceinline__ ChunkIdx2 load_chunk_idx2(const int* __restrict__ xidx, int stride, unsigned row0, unsigned n_valid, int tid) {
  const unsigned r = row0 + (tid >> 4), lastv = n_valid - 1;
  ChunkIdx2 k;
  k.j0 = xidx[(unsigned long long)(r < lastv ? r : lastv) * (unsigned)stride];
  k.j1 = xidx[(unsigned long long)(r + 32 < lastv ? r + 32 : lastv) * (unsigned)stride];
  return k;
}
template <bool F32>
__device__ __forceinline__ Chunk2 load_chunk_rows2(const void* __restrict__ base, int ld, const ChunkIdx2& I, int col0, int tid) {
  const int col = col0 + (tid & 15) * 8;
  const long long o0 = (long long)(I.j0 < 0 ? 0 : I.j0) * ld + col, o1 = (long long)(I.j1 < 0 ? 0 : I.j1) * ld + col;
  Chunk2 k;
  if (F32) {
    const float* f = (const float*)base;
    const float4 a0 = *reinterpret_cast<const float4*>(f + o0), b0 = *reinterpret_cast<const float4*>(f + o0 + 4);
    const float4 a1 = *reinterpret_cast<const float4*>(f + o1), b1 = *reinterpret_cast<const float4*>(f + o1 + 4);
    k.q0 = cvt_f32x8(a0, b0); k.q1 = cvt_f32x8(a1, b1);
  } else {
    const unsigned short* h = (const unsigned short*)base;
    k.q0 = *reinterpret_cast<const uint4*>(h + o0);
    k.q1 = *reinterpret_cast<const uint4*>(h + o1);
  }
  return k;
}
__device__ __forceinline__ void and_u4(uint4& q, unsigned m) { q.x &= m; q.y &= m; q.z &= m; q.w &= m; }
// rows >= n_valid / >= the slice's end and missing taps (bits) are cleared when the chunk is staged
__device__ __forceinline__ void clear_chunk2(Chunk2& k, long long row0, int tid, long long n_valid, unsigned miss_bits) {
  const int r = tid >> 4;
  and_u4(k.q0, (row0 + r < n_valid && !(miss_bits & 1u)) ? ~0u : 0u);
  and_u4(k.q1, (row0 + r + 32 < n_valid && !(miss_bits & 2u)) ? ~0u : 0u);
}
__device__ __forceinline__ void store_chunk2(unsigned char* __restrict__ lds, int tid, const Chunk2& k) {
  const int c = tid & 15, r = tid >> 4;
  unsigned char* p = lds + stage_off(r, c);            // (row & 3) is the same for r and r + 32
  *reinterpret_cast<uint4*>(p) = k.q0;
  *reinterpret_cast<uint4*>(p + 32 * kRowBytes) = k.q1;
}

template <int XM>
__global__ __launch_bounds__(512, 1) void k_dw_grouped2(GdDwGroup A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];        // [buffer][G | X sub 0 | X sub 1][kStage]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int xcd = blockIdx.x & 7, pos = blockIdx.x >> 3;
  const int sg = pos / A.tiles_total, t = pos - sg * A.tiles_total;
  const int s = sg * 8 + xcd;
  const bool taps = A.pair_mode == 2;
  const unsigned short* G = (const unsigned short*)A.job[0].G;
  const void* X = A.job[0].X;
  float* part = A.job[0].part;
  float* colpart = A.job[0].colpart;
  int M = A.job[0].M, N = A.job[0].N, tile0 = 0;
  const int* xidx = A.job[0].xidx;
  int xstride = A.job[0].xidx_stride;
  // sub-tile 1 of a tap pair: the NEXT job's index column and partial tiles (null: an odd tap without a partner)
  const int* xidx_b = A.n_jobs > 1 ? A.job[1].xidx : nullptr;
  float* part_b = A.n_jobs > 1 ? A.job[1].part : nullptr;
#pragma unroll
  for (int j = 1; j < GD_DW_MAX_JOBS; ++j)
    if (j < A.n_jobs && t >= A.job[j].tile0) {          // (tap pairs: odd jobs carry tile0 = INT_MAX and are never selected)
      G = (const unsigned short*)A.job[j].G;
      X = A.job[j].X;
      part = A.job[j].part;
      colpart = A.job[j].colpart;
      M = A.job[j].M;
      N = A.job[j].N;
      tile0 = A.job[j].tile0;
      xidx = A.job[j].xidx;
      xstride = A.job[j].xidx_stride;
      xidx_b = j + 1 < A.n_jobs ? A.job[j + 1 < GD_DW_MAX_JOBS ? j + 1 : j].xidx : nullptr;
      part_b = j + 1 < A.n_jobs ? A.job[j + 1 < GD_DW_MAX_JOBS ? j + 1 : j].part : nullptr;
    }
  const int tn_count = taps ? N / kTile : N / (2 * kTile);
  const int tm = (t - tile0) / tn_count, tq = (t - tile0) - tm * tn_count;
  const int xcol_a = taps ? tq * kTile : tq * 2 * kTile, xcol_b = taps ? xcol_a : xcol_a + kTile;
  if (!taps) { xidx_b = xidx; part_b = part; }
  const bool have_b = part_b != nullptr;
  if (!have_b) xidx_b = xidx;                          // readable stand-in; sub-tile 1 is computed and dropped
  const long long r0 = (long long)s * A.rows_per_slice;
  const int nchunk = (int)(A.rows_per_slice / kChunk);
  const int sub = wv >> 2, wm = (wv >> 1) & 1, wn = wv & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float cs[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) cs[j] = 0.f;
  const bool want_cs = colpart != nullptr && tq == 0;
  const int gcol = tm * kTile;
  const unsigned last = A.guard_rows ? (unsigned)(A.n_valid - 1) : 0xFFFFFFFFu;
  const unsigned nv = (unsigned)A.n_valid, rb = (unsigned)r0, rt = (unsigned)r0 + (tid >> 4);
  const int c8 = (tid & 15) * 8;
  const unsigned short* gp = G + gcol + c8;
  const unsigned short* xpa = (const unsigned short*)X + xcol_a + c8;
  const unsigned short* xpb = (const unsigned short*)X + xcol_b + c8;
  ChunkIdx2 ia0 = {-1, -1}, ia1 = {-1, -1}, ib0 = {-1, -1}, ib1 = {-1, -1};
  unsigned ma0 = 0u, ma1 = 0u, mb0 = 0u, mb1 = 0u;
  auto load_xa = [&](unsigned chunk, ChunkIdx2& I, unsigned& miss) {
    if constexpr (XM == 0) {
      return load_chunk2(xpa, N, rt + chunk * kChunk, last);
    } else {
      const Chunk2 k = load_chunk_rows2<XM == 2>(X, N, I, xcol_a, tid);
      miss = (I.j0 < 0 ? 1u : 0u) | (I.j1 < 0 ? 2u : 0u);
      I = load_chunk_idx2(xidx, xstride, rb + (chunk + 2) * kChunk, nv, tid);
      return k;
    }
  };
  auto load_xb = [&](unsigned chunk, ChunkIdx2& I, unsigned& miss) {
    if constexpr (XM == 0) {
      return load_chunk2(xpb, N, rt + chunk * kChunk, last);
    } else {
      const Chunk2 k = load_chunk_rows2<XM == 2>(X, N, I, xcol_b, tid);
      miss = (I.j0 < 0 ? 1u : 0u) | (I.j1 < 0 ? 2u : 0u);
      I = load_chunk_idx2(xidx_b, xstride, rb + (chunk + 2) * kChunk, nv, tid);
      return k;
    }
  };
  auto load_g = [&](unsigned chunk) { return load_chunk2(gp, M, rt + chunk * kChunk, last); };
  if constexpr (XM != 0) {
    ia0 = load_chunk_idx2(xidx, xstride, rb, nv, tid);
    ib0 = load_chunk_idx2(xidx_b, xstride, rb, nv, tid);
    ia1 = load_chunk_idx2(xidx, xstride, rb + kChunk, nv, tid);
    ib1 = load_chunk_idx2(xidx_b, xstride, rb + kChunk, nv, tid);
  }
  auto compute = [&](const unsigned char* bg, const unsigned char* bx) {
#pragma unroll
    for (int ks = 0; ks < kChunk / 16; ++ks) {
      const bf16x8 a0 = frag(bg, 16 * ks, wm * 64, lane), a1 = frag(bg, 16 * ks, wm * 64 + 32, lane);
      const bf16x8 b0 = frag(bx, 16 * ks, wn * 64, lane), b1 = frag(bx, 16 * ks, wn * 64 + 32, lane);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
  };
  unsigned char* const buf0 = lds;
  unsigned char* const buf1 = lds + 3 * kStage;
  const long long r_end = r0 + A.rows_per_slice < A.n_valid ? r0 + A.rows_per_slice : A.n_valid;
  Chunk2 g0 = load_g(0), xa0 = load_xa(0, ia0, ma0), xb0 = load_xb(0, ib0, mb0);
  __builtin_amdgcn_sched_barrier(0);
  const unsigned c1 = nchunk > 1 ? 1u : 0u;
  Chunk2 g1 = load_g(c1), xa1 = load_xa(c1, ia1, ma1), xb1 = load_xb(c1, ib1, mb1);
  __builtin_amdgcn_sched_barrier(0);
  for (int c = 0; c < nchunk; c += 2) {
    {
      const long long cr = r0 + (long long)c * kChunk;
      clear_chunk2(g0, cr, tid, r_end, 0u);
      clear_chunk2(xa0, cr, tid, r_end, XM != 0 ? ma0 : 0u);
      clear_chunk2(xb0, cr, tid, r_end, XM != 0 ? mb0 : 0u);
    }
    store_chunk2(buf0, tid, g0);
    store_chunk2(buf0 + kStage, tid, xa0);
    store_chunk2(buf0 + 2 * kStage, tid, xb0);
    if (want_cs) { add_bf16x8(g0.q0, cs); add_bf16x8(g0.q1, cs); }
    __syncthreads();
    {
      const unsigned cn = c + 2 < nchunk ? c + 2 : nchunk - 1;
      g0 = load_g(cn);
      xa0 = load_xa(cn, ia0, ma0);
      xb0 = load_xb(cn, ib0, mb0);
    }
    __builtin_amdgcn_sched_barrier(0);
    compute(buf0, buf0 + (1 + sub) * kStage);
    {
      const long long cr = r0 + (long long)(c + 1) * kChunk;
      clear_chunk2(g1, cr, tid, r_end, 0u);
      clear_chunk2(xa1, cr, tid, r_end, XM != 0 ? ma1 : 0u);
      clear_chunk2(xb1, cr, tid, r_end, XM != 0 ? mb1 : 0u);
    }
    store_chunk2(buf1, tid, g1);
    store_chunk2(buf1 + kStage, tid, xa1);
    store_chunk2(buf1 + 2 * kStage, tid, xb1);
    if (want_cs) { add_bf16x8(g1.q0, cs); add_bf16x8(g1.q1, cs); }
    __syncthreads();
    {
      const unsigned cn = c + 3 < nchunk ? c + 3 : nchunk - 1;
      g1 = load_g(cn);
      xa1 = load_xa(cn, ia1, ma1);
      xb1 = load_xb(cn, ib1, mb1);
    }
    __builtin_amdgcn_sched_barrier(0);
    compute(buf1, buf1 + (1 + sub) * kStage);
  }
  // ---- partial tile of this wavefront's sub-tile
  if (sub == 0 || have_b) {
    float* out = (sub ? part_b : part) + ((long long)s * M + tm * kTile + wm * 64) * N + (sub ? xcol_b : xcol_a) + wn * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          out[(long long)m * N + j * 32 + (lane & 31)] = acc[i][j][e];
        }
  }
  if (want_cs) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);       // (32 row groups, 128 columns)
    const int c = tid & 15, r = tid >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) red[r * kTile + c * 8 + j] = cs[j];
    __syncthreads();
    if (tid < kTile) {
      float tsum = 0.f;
#pragma unroll
      for (int rr = 0; rr < 32; ++rr) tsum += red[rr * kTile + tid];
      colpart[(long long)s * M + tm * kTile + tid] = tsum;
    }
  }
}
}  // namespace

bool gd_dw_group_supported(long long n_pad, int d, int ff) {
  return d % kTile == 0 && ff % kTile == 0 && d <= 512 && ff <= 1024 && n_pad > 0 && n_pad % (8 * kChunk) == 0;
}
// number of row slices: a multiple of 8 (one XCD per slice residue), tiles x slices ~ 2 workgroups per CU
int gd_dw_group_slices(long long n_pad, int tiles_total) {
  // ~2 workgroups per CU for a layer with many rows, ~1 for a short one: every slice writes a full set of fp32 partial tiles (33 MB per
  // d = 256 layer at 16 slices) whatever its row count, so below GDMAE_DW_ROWS rows half the slices are cheaper than the parallelism they
  // buy (config C's 4-frame step, same box: 4.67 -> 4.64 ms with 256 workgroups for all layers; at 8 frames 512 stays best for the
  // long stages).  GDMAE_DW_WGS pins the count (experiment switch).
  static const int pinned = getenv("GDMAE_DW_WGS") ? atoi(getenv("GDMAE_DW_WGS")) : 0;
  static const long long short_rows = getenv("GDMAE_DW_ROWS") ? atoll(getenv("GDMAE_DW_ROWS")) : 32768;
  const int wgs = pinned > 0 ? pinned : (n_pad < short_rows ? 256 : 512);
  return gd_dw_group_slices_for(n_pad, tiles_total, wgs);
}
// Slice count FIRST, then the row padding that makes it fit (gathered launches: the row count is data dependent, and a padding
// granule of 1024 rows capped the slices at 16 - 288 workgroups for the 384 k decoder sites): S doubles while tiles x S stays within
// max_wgs and a slice keeps >= 4 chunks; *n_pad = rows padded to S x 64.
int gd_dw_pick(long long n, int tiles_total, int max_wgs, long long* n_pad) {
  int S = 8;
  while (S < 64 && (long long)tiles_total * S * 2 <= max_wgs && n / (S * 2) >= 4 * kChunk) S *= 2;
  const long long g = (long long)S * kChunk;
  *n_pad = (n + g - 1) / g * g;
  if (*n_pad < g) *n_pad = g;
  return S;
}
// ... with the number of workgroups the launch should not exceed after doubling (the layer launch wants ~1 workgroup per CU:
// its tiles re-read few operands; the gathered sparse-convolution launch is latency-bound and wants both resident slots filled)
int gd_dw_group_slices_for(long long n_pad, int tiles_total, int max_wgs) {
  int S = 8;
  while (S < 64 && (long long)tiles_total * S * 2 <= max_wgs && n_pad % ((long long)S * 2 * kChunk) == 0 && n_pad / (S * 2) >= 2 * kChunk) S *= 2;
  return S;
}

// jobs: G (n_pad, M) / X (n_pad, N) bf16 row-major, M and N multiples of 128; part: (S, M, N) fp32, colpart: (S, M) fp32 or null;
// rows >= n_valid are ignored (n_pad = the row count padded to the slice grid; the buffers must extend to n_pad rows).
// Fills tile0 / tiles_total / S / rows_per_slice and launches.
int gd_dw_grouped(hipStream_t st, GdDwGroup& A, long long n_pad, long long n_valid) { return gd_dw_grouped_s(st, A, n_pad, n_valid, 0); }
// S > 0: that many row slices (a multiple of 8 that divides n_pad / 64) instead of the default choice
int gd_dw_grouped_s(hipStream_t st, GdDwGroup& A, long long n_pad, long long n_valid, int S) {
  GD_REQUIRE(A.n_jobs >= 1 && A.n_jobs <= GD_DW_MAX_JOBS, "dw_grouped: job count");
  int tiles = 0;
  for (int j = 0; j < A.n_jobs; ++j) {
    GD_REQUIRE(A.job[j].M % kTile == 0 && A.job[j].N % kTile == 0, "dw_grouped: M, N must be multiples of 128");
    A.job[j].tile0 = tiles;
    tiles += (A.job[j].M / kTile) * (A.job[j].N / kTile);
  }
  A.tiles_total = tiles;
  GD_REQUIRE(n_pad % (8 * kChunk) == 0, "dw_grouped: rows must be a multiple of 512");
  A.S = S > 0 ? S : gd_dw_group_slices(n_pad, tiles);
  GD_REQUIRE(A.S % 8 == 0 && n_pad % ((long long)A.S * kChunk) == 0, "dw_grouped: slices must be a multiple of 8 that divides the row chunks");
  A.rows_per_slice = n_pad / A.S;
  A.n_valid = n_valid;
  // Algorithmic (HBM) bytes: every DISTINCT operand matrix of the launch once - the nine tap jobs of a gathered launch share one G
  // and one X, the q/k and v gradients of a layer share nothing - + the index columns + the fp32 partial tiles written.  What the
  // workgroups request from L2 (every job reads its G and X rows: the 9-fold re-read of G by the tap jobs, the 2 - 4-fold re-reads by
  // the tiles of a slice not even counted) is reported separately as the side figure (bench.py: l2_stream_bytes_per_launch).
  double by = 0.0, fl = 0.0, l2 = 0.0;
  for (int j = 0; j < A.n_jobs; ++j) {
    bool g_new = true, x_new = true;
    for (int i = 0; i < j; ++i) {
      g_new = g_new && A.job[i].G != A.job[j].G;
      x_new = x_new && A.job[i].X != A.job[j].X;
    }
    const double gcols = A.job[j].g_cols ? A.job[j].g_cols : A.job[j].M, xb = A.job[j].x_f32 ? 4.0 : 2.0;
    const double part = 4.0 * A.S * (double)A.job[j].M * A.job[j].N;
    by += (g_new ? 2.0 * n_valid * gcols : 0.0) + (x_new ? xb * n_valid * A.job[j].N : 0.0) + (A.job[j].xidx ? 4.0 * n_valid : 0.0) + part;
    l2 += 2.0 * n_valid * gcols + xb * n_valid * A.job[j].N + (A.job[j].xidx ? 4.0 * n_valid : 0.0) + part;
    fl += 2.0 * n_valid * (double)A.job[j].M * A.job[j].N;
  }
  GdTimed timed(GD_T_DW_GROUPED, st, by, fl, l2);
  GD_REQUIRE(n_valid >= 0 && n_pad < (1ll << 31), "dw_grouped: row count");
  if (n_valid == 0) {                    // no rows (an empty stage): the partial tiles are zero; the kernel's clamped loads need a row 0
    for (int j = 0; j < A.n_jobs; ++j) {
      GD_CHECK(hipMemsetAsync(A.job[j].part, 0, (size_t)A.S * A.job[j].M * A.job[j].N * sizeof(float), st));
      if (A.job[j].colpart) GD_CHECK(hipMemsetAsync(A.job[j].colpart, 0, (size_t)A.S * A.job[j].M * sizeof(float), st));
    }
    return 0;
  }
  const int xm = A.job[0].xidx ? (A.job[0].x_f32 ? 2 : 1) : 0;
  for (int j = 1; j < A.n_jobs; ++j)
    GD_REQUIRE((A.job[j].xidx ? (A.job[j].x_f32 ? 2 : 1) : 0) == xm, "dw_grouped: the jobs of a launch share the X addressing mode");
  // ---- 128 x 256 pair tiles where the launch's jobs allow it: OPT-IN (GDMAE_DW_PAIR=1).  Built for VERDICT r5 task 4 and measured
  // same-box (tools/ab_env.sh): 7.72 -> 7.82 ms per 8-frame step, 4.82 -> 4.86 at 4 frames - 0.75 of the L2 stream per product, but
  // eight wavefronts behind one barrier leave ONE workgroup per CU (194 registers: two wavefronts per SIMD either way), and a CU whose
  // only workgroup waits at a barrier idles, where two independent four-wavefront workgroups cover each other (the finding of
  // TL_HALVES for the fused layer kernels again).  Bit-identical partial tiles; the bias column sums of a layer launch in another fixed order (tests/test_ride_along.py runs a step with it).
  static const bool pair_ok = getenv("GDMAE_DW_PAIR") && atoi(getenv("GDMAE_DW_PAIR")) != 0;
  int mode = 0;
  if (pair_ok) {
    bool plain = true, cols = true, same = true;
    for (int j = 0; j < A.n_jobs; ++j) {
      const GdDwJob& q = A.job[j];
      plain = plain && q.g_ld == 0 && q.g_cols == 0;
      cols = cols && q.N % (2 * kTile) == 0;
      same = same && q.G == A.job[0].G && q.X == A.job[0].X && q.M == A.job[0].M && q.N == A.job[0].N && q.xidx != nullptr &&
             q.xidx_stride == A.job[0].xidx_stride && q.colpart == nullptr;
    }
    if (plain && cols) mode = 1;
    else if (plain && same && A.n_jobs >= 2) mode = 2;
  }
  if (mode != 0) {
    A.pair_mode = mode;
    int pt = 0;
    for (int j = 0; j < A.n_jobs; ++j) {
      if (mode == 1) {
        A.job[j].tile0 = pt;
        pt += (A.job[j].M / kTile) * (A.job[j].N / (2 * kTile));
      } else if ((j & 1) == 0) {
        A.job[j].tile0 = pt;
        pt += (A.job[j].M / kTile) * (A.job[j].N / kTile);
      } else {
        A.job[j].tile0 = 0x7FFFFFFF;                   // the second tap of a pair: reached through its partner
      }
    }
    A.tiles_total = pt;
    static bool once[3] = {false, false, false};
    if (!once[xm]) {
      const void* fn = xm == 0 ? (const void*)k_dw_grouped2<0> : (xm == 1 ? (const void*)k_dw_grouped2<1> : (const void*)k_dw_grouped2<2>);
      GD_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * kStage));
      once[xm] = true;
    }
    const dim3 grid2((unsigned)(pt * A.S));
    if (xm == 0) hipLaunchKernelGGL(k_dw_grouped2<0>, grid2, dim3(512), 6 * kStage, st, A);
    else if (xm == 1) hipLaunchKernelGGL(k_dw_grouped2<1>, grid2, dim3(512), 6 * kStage, st, A);
    else hipLaunchKernelGGL(k_dw_grouped2<2>, grid2, dim3(512), 6 * kStage, st, A);
    GD_LAUNCH_CHECK();
    return 0;
  }
  const dim3 grid((unsigned)(tiles * A.S));
  if (xm == 0) hipLaunchKernelGGL(k_dw_grouped<0>, grid, dim3(256), 4 * kStage, st, A);
  else if (xm == 1) hipLaunchKernelGGL(k_dw_grouped<1>, grid, dim3(256), 4 * kStage, st, A);
  else hipLaunchKernelGGL(k_dw_grouped<2>, grid, dim3(256), 4 * kStage, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// C ABI (tests, stand-alone use): one matrix.  dW (M, N) fp32 = G^T X, dbias (M) fp32 = column sums of G (optional)
// ---------------------------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void k_dw_reduce(const float* __restrict__ part, int S, long long P, float* __restrict__ dst) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += part[(long long)s * P + i];
    dst[i] = a;
  }
}
}  // namespace
extern "C" size_t gdmae_dw_gemm_workspace_bytes(long long rows, int M, int N) {
  const int S = gd_dw_group_slices(rows, (M / kTile) * (N / kTile));
  return gd_align((size_t)S * M * N * sizeof(float)) + gd_align((size_t)S * M * sizeof(float));
}
extern "C" int gdmae_dw_gemm(const void* G, const void* X, long long rows, int M, int N, float* dW, float* dbias, void* workspace,
                             void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GD_REQUIRE(M % kTile == 0 && N % kTile == 0 && rows > 0 && rows % (8 * kChunk) == 0, "dw_gemm: M, N multiples of 128, rows a multiple of 512");
  GdDwGroup A;
  A.n_jobs = 1;
  const int S = gd_dw_group_slices(rows, (M / kTile) * (N / kTile));
  float* part = (float*)workspace;
  float* colpart = (float*)((char*)workspace + gd_align((size_t)S * M * N * sizeof(float)));
  A.job[0] = GdDwJob{G, X, M, N, part, dbias ? colpart : nullptr, 0};
  {
    const int rc = gd_dw_grouped(st, A, rows, rows);
    if (rc != 0) return rc;
  }
  hipLaunchKernelGGL(k_dw_reduce, dim3(256), dim3(256), 0, st, (const float*)part, A.S, (long long)M * N, dW);
  GD_LAUNCH_CHECK();
  if (dbias) {
    hipLaunchKernelGGL(k_dw_reduce, dim3(4), dim3(256), 0, st, (const float*)colpart, A.S, (long long)M, dbias);
    GD_LAUNCH_CHECK();
  }
  return 0;
}
