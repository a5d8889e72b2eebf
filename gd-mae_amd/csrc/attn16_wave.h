// (bodies; kernels in attention_t16.hip and - merged with the other levels of a layer - attention_coop.hip)
// bf16-MFMA windowed cosine attention for the sparse occupancy level (T = 16 padded tokens), bf16 token I/O.
//
// Same contract as the lane-per-query kernels of attention.hip (reference cosine_msa.py:114-176 through
// sst_basic_block.py:22-54): one wavefront = a 16-row tile (1 ... 4 consecutive windows of the level, packed) x one head.  The VALU
// kernels spend ~4 k fp32 FMAs and ~1 k 16-byte LDS reads per wavefront on a 16 x 16 x DH problem; here every product is one
// v_mfma_f32_16x16x{32,16}_bf16 per head:
//
//  * lane l = (c, g) = (l & 15, l >> 4) loads the 4-element pieces dh = 16 p + 4 g + {0..3}
//    (p < DH / 16) of token row c - as A operand that is "row c, k-slots of group g", as B operand "column c", and it is
//    exactly the piece of row c that the token-contracted outputs (dQ^T, dK^T, dV^T, O^T tiles: column c, rows 4 g + j of
//    tile p) hand back to the lane, so epilogues are lane-local;
//  * logits use the RAW bf16 rows: (q . k) is exact in the fp32 accumulator (products of bf16 values), and the cosine
//    normalisation is applied afterwards as  s = acc * (1 / |q| tau) * (1 / |k|)  - no split-bf16 operands, one MFMA;
//    the same factors are folded into dS before it is rounded to bf16 for the dQ / dK products, whose A operands are again
//    the raw rows (read transposed from LDS with ds_read_b64_tr_b16);
//  * softmax statistics: 4 accumulator registers per lane + a cross-group reduction with v_permlane16/32_swap;
//  * the backward evaluates the scores in both orientations, like attention_t32.hip (S^T with the query on the lane for dQ, S with the
//    key on the lane for dK / dV), so there are no atomics and gradients are deterministic.
#pragma once
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#ifndef ATTN16_UPFRONT
#define ATTN16_UPFRONT 0      // 1: experiment switch (first two passes of a quad planned up front; measured: 145 vs 143 us per step, not taken)
#endif

namespace t16w {
constexpr float kInvEpsNorm = 1e12f;       // 1 / 1e-12 (F.normalize eps)
constexpr int kPitch = 32;                 // LDS tile row pitch in bf16 elements (64 B: conflict-free transposed reads)
constexpr int kTile = 16 * kPitch;         // one [token][dh] tile

struct A16Args {
  const unsigned short* qk;
  const unsigned short* v;
  unsigned short* out;
  const int* csr_tok;
  const int* win_start;
  const int* win_len;
  int n_win, d, H;
  const float* tau;
  float tau_min;
};
struct A16BwdArgs {
  const unsigned short* qk;
  const unsigned short* v;
  const unsigned short* dout;
  unsigned short* dqk;
  unsigned short* dv;
  float* dtau_part;
  const int* csr_tok;
  const int* win_start;
  const int* win_len;
  int n_win, d, H;
  const float* tau;
  float tau_min;
};

template <int NP>
struct Row {            // this lane's pieces of one token row: NP x 4 bf16
  uint2 p[NP];
};

__device__ __forceinline__ float lo_f(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_f(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }

// sum / max over the 4 lane groups that share a column c
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

// LDS tiles are row-major [token][32 bf16]; the eight 8-byte units of a row are XOR-permuted by a function of the row so that
// the 16 rows of a ds_write_b64 lane group (32 banks) and the rows r, r + 4 of a transposed-read group (64 banks) do not meet
// on a bank (rows are 64 B apart).
__device__ __forceinline__ int unit_off(int row, int unit) {
  const int x = ((row >> 1) & 1) | (((row >> 3) & 1) << 1) | (((row >> 2) & 1) << 2);
  return row * kPitch + 4 * (unit ^ x);
}
// Rows are loaded unconditionally (a padding row reads token 0) and cleared afterwards by keep_row: a load under `act ? ... : 0`
// is a branch, and where it joins the compiler drains the load counter - the q / k / v / dO rows arrived one round trip after the other.
template <int NP>
__device__ __forceinline__ Row<NP> load_row(const unsigned short* __restrict__ base) {
  Row<NP> r;
#pragma unroll
  for (int p = 0; p < NP; ++p) r.p[p] = *reinterpret_cast<const uint2*>(base + 16 * p);
  return r;
}
template <int NP>
__device__ __forceinline__ void keep_row(Row<NP>& r, bool act) {
  const unsigned m = act ? 0xFFFFFFFFu : 0u;
#pragma unroll
  for (int p = 0; p < NP; ++p) { r.p[p].x &= m; r.p[p].y &= m; }
}
template <int NP>
__device__ __forceinline__ void store_tile(unsigned short* __restrict__ tile, int c, int g, const Row<NP>& r) {
#pragma unroll
  for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(tile + unit_off(c, 4 * p + g)) = r.p[p];
}
// 1 / max(|row|, 1e-12) of the full row (all 4 groups of column c contribute their pieces)
template <int NP>
__device__ __forceinline__ float inv_norm(const Row<NP>& r) {
  float ss = 0.f;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const bf16x2 a = __builtin_bit_cast(bf16x2, r.p[p].x), b = __builtin_bit_cast(bf16x2, r.p[p].y);
    ss = __builtin_amdgcn_fdot2_f32_bf16(a, a, ss, false);
    ss = __builtin_amdgcn_fdot2_f32_bf16(b, b, ss, false);
  }
  ss = grp_sum(ss);
  return fminf(__builtin_amdgcn_rsqf(ss), kInvEpsNorm);
}
// dh-contracted product of two token tiles: D[row of a][row of b] (a: A operand, b: B operand)
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
__device__ __forceinline__ void store_piece(unsigned short* __restrict__ dst, float a, float b, float c, float d) {
  f32x4 t = {a, b, c, d};
  const bf16x4 o = __builtin_convertvector(t, bf16x4);
  *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, o);
}
__device__ __forceinline__ void piece_f32(const uint2& w, float (&x)[4]) {
  x[0] = lo_f(w.x); x[1] = hi_f(w.x); x[2] = lo_f(w.y); x[3] = hi_f(w.y);
}

// LDS per wavefront (= one head): {tile A, tile B} + 4 x 16 floats + 16 window ids
constexpr int kWaveLds = 2 * kTile * 2 + 64 * 4 + 64;              // bytes
constexpr int kWinPerWave = 4;       // consecutive windows of the level handled by one wavefront
constexpr int kPadWin = 31;          // window id of a padding row

// The level's windows hold 1 ... 16 tokens but ~5 on average, so a wavefront takes kWinPerWave consecutive windows and
// packs them greedily (in order) into 16-row tiles: one pass = the windows [a, b) whose tokens fit one tile.  Rows of
// different windows never attend to each other (their logits are masked like padded keys), so the results are those of
// the one-window-per-tile kernel; 16 / 4.7 tokens would allow 3.4 windows per tile, four consecutive windows give ~2.4.
struct Pass {
  int a, b;        // windows [a, b) of the wave's kWinPerWave
  int wid;         // this lane's row c: window (0 ... 3) or kPadWin
  int tok;         // token index of row c (0 for padding rows)
};
__device__ __forceinline__ bool next_pass(Pass& ps, const int (&len)[kWinPerWave], const int (&start)[kWinPerWave], const int* __restrict__ csr_tok,
                                          int c) {
  int a = ps.b;
  // skip windows past the end of the level (length 0)
#pragma unroll
  for (int i = 0; i < kWinPerWave; ++i)
    if (i == a && len[i] == 0) ++a;
  // (no early return when a >= kWinPerWave: the pass then has no windows, every row is padding and the index load below reads entry 0 -
  //  the forward plans its second pass without a branch so that both passes' rows are in flight together)
  int b = a, fill = 0;
#pragma unroll
  for (int i = 0; i < kWinPerWave; ++i)
    if (i == b && i >= a && len[i] > 0 && fill + len[i] <= 16) {
      fill += len[i];
      b = i + 1;
    }
  int off = 0, idx = -1, wid = kPadWin;
#pragma unroll
  for (int i = 0; i < kWinPerWave; ++i) {
    const int li = (i >= a && i < b) ? len[i] : 0;
    if (c >= off && c < off + li) {
      wid = i;
      idx = start[i] + (c - off);
    }
    off += li;
  }
  ps.a = a;
  ps.b = b;
  ps.wid = wid;
  ps.tok = csr_tok ? csr_tok[idx >= 0 ? idx : 0] : (idx >= 0 ? idx : 0);           // padding rows read entry 0: their rows are cleared (keep_row), no branch around the load
  return a < kWinPerWave;
}

template <int DH>
__device__ __forceinline__ void t16_fwd_body(const A16Args& A, const unsigned blk, unsigned char* __restrict__ smem_t16) {
  constexpr int NP = DH / 16;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  unsigned short* tV = reinterpret_cast<unsigned short*>(smem_t16 + wib * kWaveLds);
  float* sK = reinterpret_cast<float*>(tV + 2 * kTile);
  int* sWid = reinterpret_cast<int*>(sK + 64);
  // workgroup = (window quad, head quad): its 4 wavefronts take one head each (the quad's q / k / v pieces share cache lines)
  const int groups = A.H >> 2;
  const int wq = blk / groups, hd = 4 * (blk % groups) + wib;
  int len[kWinPerWave], start[kWinPerWave];
#pragma unroll
  for (int i = 0; i < kWinPerWave; ++i) {
    const int w = kWinPerWave * wq + i;
    const int wc = w < A.n_win ? w : A.n_win - 1;         // unconditional loads, the length cleared afterwards
    const int l_ = A.win_len[wc];
    start[i] = A.win_start[wc];
    len[i] = w < A.n_win ? l_ : 0;
  }
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);
  const int d = A.d;
  const int col = hd * DH + 4 * g;
  // one pass = the packed windows [a, b) of the quad; rows q / k / v of its 16 slots are already in registers
  auto run_pass = [&](const Pass& ps, Row<NP> q, Row<NP> k, Row<NP> v) {
    const bool act = ps.wid != kPadWin;
    const int tok = ps.tok;
    keep_row<NP>(q, act);
    keep_row<NP>(k, act);
    keep_row<NP>(v, act);
    store_tile<NP>(tV, c, g, v);
    const float qa = inv_norm<NP>(q) * inv_tau;
    const float kin = inv_norm<NP>(k);
    if (g == 0) {
      sWid[c] = ps.wid;
      sK[c] = kin;
    }
    __builtin_amdgcn_wave_barrier();
    const int4 kw4 = *reinterpret_cast<const int4*>(sWid + 4 * g);
    const float4 kk = *reinterpret_cast<const float4*>(sK + 4 * g);
    const int kw[4] = {kw4.x, kw4.y, kw4.z, kw4.w};
    const float kj[4] = {kk.x, kk.y, kk.z, kk.w};
    f32x4 s = mma_rows<NP>(k, q);                                  // S^T[key 4 g + j][query c]
    float m = -1e30f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = (kw[j] == ps.wid) ? s[j] * qa * kj[j] : -1e30f;       // keys of other windows and padding rows: masked
      m = fmaxf(m, s[j]);
    }
    m = grp_max(m);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s[j] = __expf(s[j] - m);
      l += s[j];
    }
    l = grp_sum(l);
    const float il = __builtin_amdgcn_rcpf(l);
    f32x4 o[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) o[p] = mma_tokens(tV, p, c, g, s);      // O^T[dh 16 p + 4 g + j][query c]
    if (act) {
      unsigned short* dst = A.out + (long long)tok * d + col;
#pragma unroll
      for (int p = 0; p < NP; ++p) store_piece(dst + 16 * p, o[p][0] * il, o[p][1] * il, o[p][2] * il, o[p][3] * il);
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto rows = [&](const Pass& ps, Row<NP>& q, Row<NP>& k, Row<NP>& v) {
    q = load_row<NP>(A.qk + (long long)ps.tok * 2 * d + col);
    k = load_row<NP>(A.qk + (long long)ps.tok * 2 * d + d + col);
    v = load_row<NP>(A.v + (long long)ps.tok * d + col);
  };
#if ATTN16_UPFRONT
  // The quad's first TWO passes (1.7 on average) are planned from the scalar descriptors at once: both token-index loads go out
  // together, then both passes' rows - three dependent round trips for the wavefront instead of five (a second pass without windows
  // reads entry / row 0 and is skipped).
  Pass pa{0, 0, kPadWin, 0};
  next_pass(pa, len, start, A.csr_tok, c);
  Pass pb = pa;
  const bool vb = next_pass(pb, len, start, A.csr_tok, c);
  Row<NP> qa, ka, va, qb, kb, vb_;
  rows(pa, qa, ka, va);
  rows(pb, qb, kb, vb_);
  run_pass(pa, qa, ka, va);
  if (vb) {
    run_pass(pb, qb, kb, vb_);
    Pass ps = pb;
    while (next_pass(ps, len, start, A.csr_tok, c)) {
      Row<NP> q, k, v;
      rows(ps, q, k, v);
      run_pass(ps, q, k, v);
    }
  }
#else
  Pass ps{0, 0, kPadWin, 0};
  while (next_pass(ps, len, start, A.csr_tok, c)) {
    Row<NP> q, k, v;
    rows(ps, q, k, v);
    run_pass(ps, q, k, v);
  }
#endif
}

template <int DH>
__device__ __forceinline__ void t16_bwd_body(const A16BwdArgs& A, const unsigned blk, const unsigned nblk, unsigned char* __restrict__ smem_t16) {
  constexpr int NP = DH / 16;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  unsigned short* tA = reinterpret_cast<unsigned short*>(smem_t16 + wib * kWaveLds);     // K (phase 1), then Q (phase 2)
  unsigned short* tB = tA + kTile;                                                        // dO (phase 2)
  float* sKin = reinterpret_cast<float*>(tB + kTile);
  float* sQa = sKin + 16;
  float* sLse = sKin + 32;
  float* sD = sKin + 48;
  int* sWid = reinterpret_cast<int*>(sKin + 64);
  const int groups = A.H >> 2;
  const int wq = blk / groups, hd = 4 * (blk % groups) + wib;
  int len[kWinPerWave], start[kWinPerWave];
#pragma unroll
  for (int i = 0; i < kWinPerWave; ++i) {
    const int w = kWinPerWave * wq + i;
    const int wc = w < A.n_win ? w : A.n_win - 1;         // unconditional loads, the length cleared afterwards
    const int l_ = A.win_len[wc];
    start[i] = A.win_start[wc];
    len[i] = w < A.n_win ? l_ : 0;
  }
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);
  const int d = A.d;
  const int col = hd * DH + 4 * g;
  float dtau = 0.f;
  Pass ps{0, 0, kPadWin, 0};
  while (next_pass(ps, len, start, A.csr_tok, c)) {
    const bool act = ps.wid != kPadWin;
    const int tok = ps.tok;
    Row<NP> q = load_row<NP>(A.qk + (long long)tok * 2 * d + col);
    Row<NP> k = load_row<NP>(A.qk + (long long)tok * 2 * d + d + col);
    Row<NP> v = load_row<NP>(A.v + (long long)tok * d + col);
    Row<NP> dO = load_row<NP>(A.dout + (long long)tok * d + col);
    keep_row<NP>(q, act);
    keep_row<NP>(k, act);
    keep_row<NP>(v, act);
    keep_row<NP>(dO, act);
    const float qin = inv_norm<NP>(q);
    const float kin = inv_norm<NP>(k);
    const float qa = qin * inv_tau;
    store_tile<NP>(tA, c, g, k);
    store_tile<NP>(tB, c, g, dO);
    if (g == 0) {
      sWid[c] = ps.wid;
      sKin[c] = kin;
      sQa[c] = qa;
    }
    __builtin_amdgcn_wave_barrier();
    const int4 kw4 = *reinterpret_cast<const int4*>(sWid + 4 * g);
    const float4 kk = *reinterpret_cast<const float4*>(sKin + 4 * g);
    const int kw[4] = {kw4.x, kw4.y, kw4.z, kw4.w};               // window of rows 4 g + j (keys in phase 1, queries in phase 2)
    const float kj[4] = {kk.x, kk.y, kk.z, kk.w};
    // ---------------- phase 1: query on the lane -> dQ ----------------
    {
      f32x4 s = mma_rows<NP>(k, q);                                // S^T[key 4 g + j][query c] (raw dot products)
      const f32x4 dP = mma_rows<NP>(v, dO);                        // dP^T[key][query]
      float m = -1e30f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s[j] = (kw[j] == ps.wid) ? s[j] * qa * kj[j] : -1e30f;     // other windows' keys and padding rows: masked
        m = fmaxf(m, s[j]);
      }
      m = grp_max(m);
      float e[4], l = 0.f, Dn = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        e[j] = __expf(s[j] - m);                                   // exactly 0 for masked keys
        l += e[j];
        Dn = fmaf(e[j], dP[j], Dn);
      }
      l = grp_sum(l);
      Dn = grp_sum(Dn);
      const float il = __builtin_amdgcn_rcpf(l);
      const float D = Dn * il;
      f32x4 dS;
      float dt = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float ds = e[j] * il * (dP[j] - D);
        dt = fmaf(ds, s[j], dt);                                   // masked key: 0 * -1e30 = -0
        dS[j] = ds * kj[j];                                        // 1 / |k| of the key folded in: the A operand is the raw K row
      }
      if (act) dtau = fmaf(-dt, inv_tau, dtau);                    // d a / d tau = -a / tau
      if (g == 0) {
        sLse[c] = act ? m + __logf(l) : 1e30f;                     // padded queries: exp(a - 1e30) = 0 in phase 2
        sD[c] = D;
      }
      float qh[NP][4], pr = 0.f;
      f32x4 dq[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        dq[p] = mma_tokens(tA, p, c, g, dS);                       // dQ^^T[dh 16 p + 4 g + j][query c], without 1 / tau
        piece_f32(q.p[p], qh[p]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          qh[p][j] *= qin;
          pr = fmaf(qh[p][j], dq[p][j], pr);
        }
      }
      pr = grp_sum(pr);
      if (act) {
        unsigned short* dst = A.dqk + (long long)tok * 2 * d + col;
#pragma unroll
        for (int p = 0; p < NP; ++p)
          store_piece(dst + 16 * p, (dq[p][0] - qh[p][0] * pr) * qa, (dq[p][1] - qh[p][1] * pr) * qa, (dq[p][2] - qh[p][2] * pr) * qa,
                      (dq[p][3] - qh[p][3] * pr) * qa);
      }
    }
    // ---------------- phase 2: key on the lane -> dK, dV ----------------
    __builtin_amdgcn_wave_barrier();
    store_tile<NP>(tA, c, g, q);
    __builtin_amdgcn_wave_barrier();
    {
      const float4 qq = *reinterpret_cast<const float4*>(sQa + 4 * g);
      const float4 ll = *reinterpret_cast<const float4*>(sLse + 4 * g);
      const float4 dd = *reinterpret_cast<const float4*>(sD + 4 * g);
      const float qj[4] = {qq.x, qq.y, qq.z, qq.w}, lj[4] = {ll.x, ll.y, ll.z, ll.w}, dj[4] = {dd.x, dd.y, dd.z, dd.w};
      const f32x4 s = mma_rows<NP>(q, k);                          // S[query 4 g + j][key c]
      const f32x4 dP = mma_rows<NP>(dO, v);                        // dP[query][key]
      f32x4 P, dS;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float pj = (kw[j] == ps.wid && act) ? __expf(s[j] * qj[j] * kin - lj[j]) : 0.f;    // same window only
        P[j] = pj;
        dS[j] = pj * (dP[j] - dj[j]) * qj[j];                      // 1 / (|q| tau) of the query folded in
      }
      float kh[NP][4], pr = 0.f;
      f32x4 dk[NP], dvv[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        dk[p] = mma_tokens(tA, p, c, g, dS);                       // dK^^T[dh][key c]
        dvv[p] = mma_tokens(tB, p, c, g, P);                       // dV^T[dh][key c]
        piece_f32(k.p[p], kh[p]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          kh[p][j] *= kin;
          pr = fmaf(kh[p][j], dk[p][j], pr);
        }
      }
      pr = grp_sum(pr);
      if (act) {
        unsigned short* dkp = A.dqk + (long long)tok * 2 * d + d + col;
        unsigned short* dvp = A.dv + (long long)tok * d + col;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
          store_piece(dkp + 16 * p, (dk[p][0] - kh[p][0] * pr) * kin, (dk[p][1] - kh[p][1] * pr) * kin, (dk[p][2] - kh[p][2] * pr) * kin,
                      (dk[p][3] - kh[p][3] * pr) * kin);
          store_piece(dvp + 16 * p, dvv[p][0], dvv[p][1], dvv[p][2], dvv[p][3]);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  // the 4 lane groups of a column hold the same reduced statistics and split the keys between them (dt sums this lane's 4
  // keys), so the plain wave sum counts every (query, key) pair once; one partial per (window quad, head)
  dtau = gd_wave_sum(dtau);
  if (lane == 0) {
    // the level owns n_win * H partial slots (one per window and head); this grid fills gridDim * 4 <= n_win * H of them and
    // zeroes the rest, so the consumer can sum the whole range without a separate clear
    const long long mine = (long long)blk * 4 + wib, used = (long long)nblk * 4, all = (long long)A.n_win * A.H;
    A.dtau_part[mine] = dtau;
    for (long long i = used + mine; i < all; i += used) A.dtau_part[i] = 0.f;
  }
}
}  // namespace t16w
