// bf16-MFMA windowed cosine attention for the T = 32 / 64 occupancy levels, bf16 token I/O (throughput mode).
//
// Contract: reference cosine_msa.py:114-176 / sst_basic_block.py:22-54 through the window CSR, as attention.hip.
// One wavefront = one (window, head); 32 x 32 x 16 bf16 MFMA tiles with the S^T layout (query on the lane) for the forward
// and for dQ, the S layout (key on the lane) for dK / dV - no atomics, deterministic.  Same operand scheme as
// attention_t16.hip:
//  * lane l = (rho, h) = (l & 31, l >> 5) holds, of token row rho of each 32-token tile, the DH / 8 pieces
//    dh = 8 t + 4 h + {0..3}: k-slots of the dh-contracted products (two pieces per 16-wide step) AND exactly the rows that
//    the token-contracted outputs (C layout: column rho, rows (r & 3) + 8 (r >> 2) + 4 h) hand back, so epilogues are
//    lane-local and nothing is re-read from memory;
//  * logits from the RAW bf16 rows (exact products, fp32 accumulation); the cosine normalisation 1 / |q| tau, 1 / |k| is
//    applied to the accumulator (the per-lane factor rides in the exponent's scale), and folded into dS before it is
//    rounded to bf16 for the dQ / dK products, whose A operands are the raw rows read transposed from row-major LDS tiles
//    (ds_read_b64_tr_b16) - no split-bf16 operands (3 -> 1 MFMA per step), no transposed 2-byte LDS scatter;
//  * padded keys get the normalised dot product -1e15 (logit -1e15 / (|q| tau)): exp() is exactly 0 and every product with
//    it stays finite; padded queries carry lse = +1e30 into phase 2, padded-key lanes start their accumulators at -1e30.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr float kInvEpsNorm = 1e12f;       // 1 / 1e-12 (F.normalize eps)
constexpr float kPadKey = -1e15f;          // normalised dot product of a padded key
constexpr float kLog2e = 1.4426950408889634f;
constexpr int kPitch = 32;                 // LDS tile row pitch in bf16 elements (64 B)

struct T32Args {
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
struct T32BwdArgs {
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

template <int NPC>
struct Row {
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
// LDS tiles are row-major [token][32 bf16] with the eight 8-byte units of a row XOR-permuted by a function of the row:
// ds_write_b64 is serviced in groups of 16 contiguous lanes over 32 banks, ds_read_b64 / ds_read_b64_tr_b16 in groups of 32
// lanes over 64 banks; with rows 64 B apart, lanes rho and rho + 2 (writes) / rho + 4 (reads) would share banks.  The
// permutation below makes the 16 rows of a write group and the 8 same-(row & 3) rows of a read group land on distinct
// units; transposed reads touch whole rows (4 rows x 64 B) and are conflict-free under any per-row permutation.
__device__ __forceinline__ int unit_off(int row, int unit) {      // element offset of logical 4-element unit `unit` of `row`
  const int x = (((row >> 1) ^ (row >> 4)) & 1) | (((row >> 2) & 3) << 1);
  return row * kPitch + 4 * (unit ^ x);
}
// Rows are loaded UNCONDITIONALLY (a padded slot reads the window's last token) and cleared afterwards by keep_row: a load under
// `act ? ... : 0` is a branch, and where it joins the compiler drains the load counter - the q / k / v / dO rows of a window used to
// arrive one round trip after the other.
template <int NPC>
__device__ __forceinline__ Row<NPC> load_row(const unsigned short* __restrict__ base) {
  Row<NPC> r;
#pragma unroll
  for (int t = 0; t < NPC; ++t) r.p[t] = *reinterpret_cast<const uint2*>(base + 8 * t);
  return r;
}
template <int NPC>
__device__ __forceinline__ void keep_row(Row<NPC>& r, bool act) {
  const unsigned m = act ? 0xFFFFFFFFu : 0u;
#pragma unroll
  for (int t = 0; t < NPC; ++t) { r.p[t].x &= m; r.p[t].y &= m; }
}
// token of window slot r (slots past the window's length repeat its last token; their rows are cleared by keep_row)
__device__ __forceinline__ int slot_token(const int* __restrict__ csr_tok, int start, int r, int n) {
  const int rr = r < n ? r : (n > 0 ? n - 1 : 0);
  return csr_tok[start + rr];
}
template <int NPC>
__device__ __forceinline__ void store_tile(unsigned short* __restrict__ tile, int row, int h, const Row<NPC>& r) {
#pragma unroll
  for (int t = 0; t < NPC; ++t) *reinterpret_cast<uint2*>(tile + unit_off(row, 2 * t + h)) = r.p[t];
}
template <int NPC>
__device__ __forceinline__ float inv_norm(const Row<NPC>& r) {
  float ss = 0.f;
#pragma unroll
  for (int t = 0; t < NPC; ++t) {
    const bf16x2 a = __builtin_bit_cast(bf16x2, r.p[t].x), b = __builtin_bit_cast(bf16x2, r.p[t].y);
    ss = __builtin_amdgcn_fdot2_f32_bf16(a, a, ss, false);
    ss = __builtin_amdgcn_fdot2_f32_bf16(b, b, ss, false);
  }
  ss = half_sum(ss);
  return fminf(__builtin_amdgcn_rsqf(ss), kInvEpsNorm);
}
template <int NPC>
__device__ __forceinline__ bf16x8 step_frag(const Row<NPC>& r, int s) {
  const uint4 u = make_uint4(r.p[2 * s].x, r.p[2 * s].y, r.p[2 * s + 1].x, r.p[2 * s + 1].y);
  return __builtin_bit_cast(bf16x8, u);
}
// acc += a-rows . b-rows^T over the head dim
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
// 8 accumulator registers [r0, r0 + 8) -> bf16 B fragment (round to nearest even)
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
__device__ __forceinline__ void store_piece(unsigned short* __restrict__ dst, float a, float b, float c, float d) {
  f32x4 t = {a, b, c, d};
  const bf16x4 o = __builtin_convertvector(t, bf16x4);
  *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, o);
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

template <int NT>
constexpr int fwd_wave_lds() { return 32 * NT * kPitch * 2 + 32 * NT * 4; }            // V tile + 1 / |k|
template <int NT>
constexpr int bwd_wave_lds() { return 2 * 32 * NT * kPitch * 2 + 4 * 32 * NT * 4; }    // 2 tiles + 4 scalar rows

template <int NT, int DH>
__device__ __forceinline__ void t32_fwd_body(const T32Args& A, const unsigned blk, unsigned char* __restrict__ smem_t32) {
  constexpr int NPC = DH / 8;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int rho = lane & 31, h = lane >> 5;
  unsigned short* tV = reinterpret_cast<unsigned short*>(smem_t32 + wib * fwd_wave_lds<NT>());
  float* sKin = reinterpret_cast<float*>(tV + 32 * NT * kPitch);
  const long long item = (long long)blk * 4 + wib;
  if (item >= (long long)A.n_win * A.H) return;
  const int w = (int)(item / A.H), hd = (int)(item % A.H);
  const int n = A.win_len[w], start = A.win_start[w];
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);
  const int d = A.d;
  Row<NPC> q[NT], k[NT];
  int tok[NT];
  float qc[NT];                                     // (1 / |q| tau) log2(e): exponent scale of this lane's query column
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int r = 32 * ti + rho;
    const bool act = r < n;
    tok[ti] = slot_token(A.csr_tok, start, r, n);
    const int col = hd * DH + 4 * h;
    q[ti] = load_row<NPC>(A.qk + (long long)tok[ti] * 2 * d + col);
    k[ti] = load_row<NPC>(A.qk + (long long)tok[ti] * 2 * d + d + col);
    Row<NPC> v = load_row<NPC>(A.v + (long long)tok[ti] * d + col);
    keep_row<NPC>(q[ti], act);
    keep_row<NPC>(k[ti], act);
    keep_row<NPC>(v, act);
    store_tile<NPC>(tV, r, h, v);
    qc[ti] = inv_norm<NPC>(q[ti]) * inv_tau * kLog2e;
    const float kin = inv_norm<NPC>(k[ti]);
    if (h == 0) sKin[r] = kin;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int qi = 0; qi < NT; ++qi) {
    f32x16 aS[NT];
    float m = -INFINITY;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj) {
      aS[kj] = mma_rows<NPC>(k[kj], q[qi], splat(0.f));          // S^T[key][query], raw dot products
      float kr[16];
      row_scalars(sKin + 32 * kj, h, kr);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        aS[kj][r] = (32 * kj + c_row(r, h) < n) ? aS[kj][r] * kr[r] : kPadKey;
        m = fmaxf(m, aS[kj][r]);
      }
    }
    m = half_max(m);
    float l = 0.f;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f((aS[kj][r] - m) * qc[qi]);
        aS[kj][r] = e;
        l += e;
      }
    l = half_sum(l);
    const float il = __builtin_amdgcn_rcpf(l);
    f32x16 o = splat(0.f);
#pragma unroll
    for (int kj = 0; kj < NT; ++kj) o = mma_tokens(tV + 32 * kj * kPitch, aS[kj], o, lane);   // O^T[dh][query]
    if (32 * qi + rho < n) {
      unsigned short* dst = A.out + (long long)tok[qi] * d + hd * DH + 4 * h;
#pragma unroll
      for (int t = 0; t < NPC; ++t) store_piece(dst + 8 * t, o[4 * t] * il, o[4 * t + 1] * il, o[4 * t + 2] * il, o[4 * t + 3] * il);
    }
  }
}

template <int NT, int DH>
__device__ __forceinline__ void t32_bwd_body(const T32BwdArgs& A, const unsigned blk, unsigned char* __restrict__ smem_t32) {
  constexpr int NPC = DH / 8;
  constexpr int TN = 32 * NT;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int rho = lane & 31, h = lane >> 5;
  unsigned short* tA = reinterpret_cast<unsigned short*>(smem_t32 + wib * bwd_wave_lds<NT>());    // K (phase 1), Q (phase 2)
  unsigned short* tB = tA + TN * kPitch;                                                           // dO (phase 2)
  float* sKin = reinterpret_cast<float*>(tB + TN * kPitch);
  float* sQa = sKin + TN;
  float* sLse = sQa + TN;          // log2 units
  float* sD = sLse + TN;
  const long long item = (long long)blk * 4 + wib;
  if (item >= (long long)A.n_win * A.H) return;
  const int w = (int)(item / A.H), hd = (int)(item % A.H);
  const int n = A.win_len[w], start = A.win_start[w];
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);
  const int d = A.d;
  const int col = hd * DH + 4 * h;
  Row<NPC> q[NT], k[NT], v[NT], dO[NT];
  int tok[NT];
  float qin[NT], kin[NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int r = 32 * ti + rho;
    const bool act = r < n;
    tok[ti] = slot_token(A.csr_tok, start, r, n);
    q[ti] = load_row<NPC>(A.qk + (long long)tok[ti] * 2 * d + col);
    k[ti] = load_row<NPC>(A.qk + (long long)tok[ti] * 2 * d + d + col);
    v[ti] = load_row<NPC>(A.v + (long long)tok[ti] * d + col);
    dO[ti] = load_row<NPC>(A.dout + (long long)tok[ti] * d + col);
  }
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const bool act = 32 * ti + rho < n;
    keep_row<NPC>(q[ti], act);
    keep_row<NPC>(k[ti], act);
    keep_row<NPC>(v[ti], act);
    keep_row<NPC>(dO[ti], act);
  }
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int r = 32 * ti + rho;
    store_tile<NPC>(tA, r, h, k[ti]);
    store_tile<NPC>(tB, r, h, dO[ti]);
    qin[ti] = inv_norm<NPC>(q[ti]);
    kin[ti] = inv_norm<NPC>(k[ti]);
    if (h == 0) {
      sKin[r] = kin[ti];
      sQa[r] = qin[ti] * inv_tau;
    }
  }
  float dtau = 0.f;
  __builtin_amdgcn_wave_barrier();
  // ================= phase 1: query on the lane (S^T, dP^T) -> dQ =================
#pragma unroll
  for (int qi = 0; qi < NT; ++qi) {
    const float qa = qin[qi] * inv_tau;
    const float qc = qa * kLog2e;
    f32x16 aS[NT], aP[NT];
    float m = -INFINITY;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj) {
      aS[kj] = mma_rows<NPC>(k[kj], q[qi], splat(0.f));
      aP[kj] = mma_rows<NPC>(v[kj], dO[qi], splat(0.f));
      float kr[16];
      row_scalars(sKin + 32 * kj, h, kr);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        aS[kj][r] = (32 * kj + c_row(r, h) < n) ? aS[kj][r] * kr[r] : kPadKey;      // t = (q . k) / |k|
        m = fmaxf(m, aS[kj][r]);
      }
    }
    m = half_max(m);
    float l = 0.f, Dn = 0.f, E1 = 0.f, E2 = 0.f;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = aS[kj][r];
        const float e = __builtin_amdgcn_exp2f((t - m) * qc);     // exactly 0 for padded keys
        const float ep = e * aP[kj][r];
        l += e;
        Dn += ep;
        E1 = fmaf(ep, t, E1);
        E2 = fmaf(e, t, E2);
        aS[kj][r] = e;
      }
    l = half_sum(l);
    Dn = half_sum(Dn);
    const float il = __builtin_amdgcn_rcpf(l);
    const float D = Dn * il;
    const bool qact = 32 * qi + rho < n;
    // sum_keys dS a = qa (E1 - D E2) / l over this lane's keys; d a / d tau = -a / tau
    dtau = fmaf(-(E1 - D * E2) * il, qa * inv_tau, dtau);
    if (h == 0) {
      sLse[32 * qi + rho] = qact ? fmaf(m, qc, __builtin_amdgcn_logf(l)) : 1e30f;   // log2 units; padded query: p = 0 in phase 2
      sD[32 * qi + rho] = D;
    }
    f32x16 oq = splat(0.f);
#pragma unroll
    for (int kj = 0; kj < NT; ++kj) {
      float kr[16];
      row_scalars(sKin + 32 * kj, h, kr);
#pragma unroll
      for (int r = 0; r < 16; ++r) aS[kj][r] = aS[kj][r] * (aP[kj][r] - D) * (kr[r] * il);      // dS / |k|
      oq = mma_tokens(tA + 32 * kj * kPitch, aS[kj], oq, lane);                                 // dQ^^T[dh][query] (without 1 / tau)
    }
    float qh[NPC][4], pr = 0.f;
#pragma unroll
    for (int t = 0; t < NPC; ++t) {
      piece_f32(q[qi].p[t], qh[t]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        qh[t][j] *= qin[qi];
        pr = fmaf(qh[t][j], oq[4 * t + j], pr);
      }
    }
    pr = half_sum(pr);
    if (qact) {
      unsigned short* dst = A.dqk + (long long)tok[qi] * 2 * d + col;
#pragma unroll
      for (int t = 0; t < NPC; ++t)
        store_piece(dst + 8 * t, (oq[4 * t] - qh[t][0] * pr) * qa, (oq[4 * t + 1] - qh[t][1] * pr) * qa, (oq[4 * t + 2] - qh[t][2] * pr) * qa,
                    (oq[4 * t + 3] - qh[t][3] * pr) * qa);
    }
  }
  // ================= phase 2: key on the lane (S, dP) -> dK, dV =================
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) store_tile<NPC>(tA, 32 * ti + rho, h, q[ti]);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int kj = 0; kj < NT; ++kj) {
    const bool kact = 32 * kj + rho < n;
    const float kc = kin[kj] * kLog2e;
    f32x16 okk = splat(0.f), ov = splat(0.f);
#pragma unroll
    for (int qi = 0; qi < NT; ++qi) {
      f32x16 aS = mma_rows<NPC>(q[qi], k[kj], splat(kact ? 0.f : -1e30f));     // S[query][key]; padded key (lane): p = 0
      f32x16 aP = mma_rows<NPC>(dO[qi], v[kj], splat(0.f));                    // dP[query][key]
      float qr[16], lr[16], dr[16];
      row_scalars(sQa + 32 * qi, h, qr);
      row_scalars(sLse + 32 * qi, h, lr);
      row_scalars(sD + 32 * qi, h, dr);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(aS[r], qr[r] * kc, -lr[r]));   // 0 for padded keys and padded queries
        aS[r] = p * (aP[r] - dr[r]) * qr[r];                                        // dS / (|q| tau)
        aP[r] = p;
      }
      okk = mma_tokens(tA + 32 * qi * kPitch, aS, okk, lane);      // dK^^T[dh][key]
      ov = mma_tokens(tB + 32 * qi * kPitch, aP, ov, lane);        // dV^T[dh][key]
    }
    float kh[NPC][4], pr = 0.f;
#pragma unroll
    for (int t = 0; t < NPC; ++t) {
      piece_f32(k[kj].p[t], kh[t]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kh[t][j] *= kin[kj];
        pr = fmaf(kh[t][j], okk[4 * t + j], pr);
      }
    }
    pr = half_sum(pr);
    if (kact) {
      unsigned short* dkp = A.dqk + (long long)tok[kj] * 2 * d + d + col;
      unsigned short* dvp = A.dv + (long long)tok[kj] * d + col;
      const float ks = kin[kj];
#pragma unroll
      for (int t = 0; t < NPC; ++t) {
        store_piece(dkp + 8 * t, (okk[4 * t] - kh[t][0] * pr) * ks, (okk[4 * t + 1] - kh[t][1] * pr) * ks, (okk[4 * t + 2] - kh[t][2] * pr) * ks,
                    (okk[4 * t + 3] - kh[t][3] * pr) * ks);
        store_piece(dvp + 8 * t, ov[4 * t], ov[4 * t + 1], ov[4 * t + 2], ov[4 * t + 3]);
      }
    }
  }
  dtau = gd_wave_sum(dtau);
  if (lane == 0) A.dtau_part[item] = dtau;
}

// ---------------------------------------------------------------------------------------------------------------------
// T = 64: TWO wavefronts per (window, head).  Wave `sub` loads only the rows of token tile `sub`, publishes them (row-major
// tiles + per-token scalars) in LDS, then owns query tile `sub` in phase 1 and key tile `sub` in phase 2; the other tile's
// rows come back from LDS as 8-byte pieces.  Half the registers of the one-wave form (no accumulator spills, 2+ waves per
// SIMD) and half the dependent work per wave.
// ---------------------------------------------------------------------------------------------------------------------
template <int NPC>
__device__ __forceinline__ Row<NPC> lds_row(const unsigned short* __restrict__ tile, int row, int h) {
  Row<NPC> r;
#pragma unroll
  for (int t = 0; t < NPC; ++t) r.p[t] = *reinterpret_cast<const uint2*>(tile + unit_off(row, 2 * t + h));
  return r;
}
constexpr int kPairFwdLds = 2 * 64 * kPitch * 2 + 64 * 4;             // K, V tiles + 1 / |k|          (per item)
constexpr int kPairBwdLds = 4 * 64 * kPitch * 2 + 4 * 64 * 4 + 16;    // K, V, Q, dO tiles + 4 scalar rows + the pair's dtau slot

template <int DH>
__device__ __forceinline__ void t64_fwd_body(const T32Args& A, const unsigned blk, unsigned char* __restrict__ smem_t32) {
  constexpr int NPC = DH / 8;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int rho = lane & 31, h = lane >> 5, sub = wib & 1;
  unsigned short* tK = reinterpret_cast<unsigned short*>(smem_t32 + (wib >> 1) * kPairFwdLds);
  unsigned short* tV = tK + 64 * kPitch;
  float* sKin = reinterpret_cast<float*>(tV + 64 * kPitch);
  const long long n_items = (long long)A.n_win * A.H;
  const long long item = min((long long)blk * 2 + (wib >> 1), n_items - 1);      // odd tail: the pair recomputes the last item
  const int w = (int)(item / A.H), hd = (int)(item % A.H);
  const int n = A.win_len[w], start = A.win_start[w];
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);
  const int d = A.d;
  const int r = 32 * sub + rho;
  const bool act = r < n;
  const int tok = slot_token(A.csr_tok, start, r, n);
  const int col = hd * DH + 4 * h;
  Row<NPC> q = load_row<NPC>(A.qk + (long long)tok * 2 * d + col);
  Row<NPC> k = load_row<NPC>(A.qk + (long long)tok * 2 * d + d + col);
  Row<NPC> v = load_row<NPC>(A.v + (long long)tok * d + col);
  keep_row<NPC>(q, act);
  keep_row<NPC>(k, act);
  keep_row<NPC>(v, act);
  store_tile<NPC>(tK, r, h, k);
  store_tile<NPC>(tV, r, h, v);
  const float qc = inv_norm<NPC>(q) * inv_tau * kLog2e;
  const float kin = inv_norm<NPC>(k);
  if (h == 0) sKin[r] = kin;
  __syncthreads();
  f32x16 aS[2];
  float m = -INFINITY;
#pragma unroll
  for (int kj = 0; kj < 2; ++kj) {
    const Row<NPC> kk = lds_row<NPC>(tK, 32 * kj + rho, h);
    aS[kj] = mma_rows<NPC>(kk, q, splat(0.f));
    float kr[16];
    row_scalars(sKin + 32 * kj, h, kr);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      aS[kj][e] = (32 * kj + c_row(e, h) < n) ? aS[kj][e] * kr[e] : kPadKey;
      m = fmaxf(m, aS[kj][e]);
    }
  }
  m = half_max(m);
  float l = 0.f;
#pragma unroll
  for (int kj = 0; kj < 2; ++kj)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p = __builtin_amdgcn_exp2f((aS[kj][e] - m) * qc);
      aS[kj][e] = p;
      l += p;
    }
  l = half_sum(l);
  const float il = __builtin_amdgcn_rcpf(l);
  f32x16 o = splat(0.f);
#pragma unroll
  for (int kj = 0; kj < 2; ++kj) o = mma_tokens(tV + 32 * kj * kPitch, aS[kj], o, lane);
  if (act) {
    unsigned short* dst = A.out + (long long)tok * d + col;
#pragma unroll
    for (int t = 0; t < NPC; ++t) store_piece(dst + 8 * t, o[4 * t] * il, o[4 * t + 1] * il, o[4 * t + 2] * il, o[4 * t + 3] * il);
  }
}

template <int DH>
__device__ __forceinline__ void t64_bwd_body(const T32BwdArgs& A, const unsigned blk, unsigned char* __restrict__ smem_t32) {
  constexpr int NPC = DH / 8;
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int rho = lane & 31, h = lane >> 5, sub = wib & 1;
  unsigned short* tK = reinterpret_cast<unsigned short*>(smem_t32 + (wib >> 1) * kPairBwdLds);
  unsigned short* tV = tK + 64 * kPitch;
  unsigned short* tQ = tV + 64 * kPitch;
  unsigned short* tO = tQ + 64 * kPitch;
  float* sKin = reinterpret_cast<float*>(tO + 64 * kPitch);
  float* sQa = sKin + 64;
  float* sLse = sQa + 64;
  float* sD = sLse + 64;
  const long long n_items = (long long)A.n_win * A.H;
  const long long item_raw = (long long)blk * 2 + (wib >> 1);
  const bool live = item_raw < n_items;                 // odd tail: the pair recomputes the last item, stores nothing
  const long long item = live ? item_raw : n_items - 1;
  const int w = (int)(item / A.H), hd = (int)(item % A.H);
  const int n = A.win_len[w], start = A.win_start[w];
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);
  const int d = A.d;
  const int r = 32 * sub + rho;
  const bool act = r < n;
  const int tok = slot_token(A.csr_tok, start, r, n);
  const int col = hd * DH + 4 * h;
  Row<NPC> q = load_row<NPC>(A.qk + (long long)tok * 2 * d + col);
  Row<NPC> k = load_row<NPC>(A.qk + (long long)tok * 2 * d + d + col);
  Row<NPC> v = load_row<NPC>(A.v + (long long)tok * d + col);
  Row<NPC> dO = load_row<NPC>(A.dout + (long long)tok * d + col);
  keep_row<NPC>(q, act);
  keep_row<NPC>(k, act);
  keep_row<NPC>(v, act);
  keep_row<NPC>(dO, act);
  store_tile<NPC>(tK, r, h, k);
  store_tile<NPC>(tV, r, h, v);
  store_tile<NPC>(tQ, r, h, q);
  store_tile<NPC>(tO, r, h, dO);
  const float qin = inv_norm<NPC>(q), kin = inv_norm<NPC>(k);
  const float qa = qin * inv_tau, qc = qa * kLog2e;
  if (h == 0) {
    sKin[r] = kin;
    sQa[r] = qa;
  }
  __syncthreads();
  float dtau;
  // ================= phase 1: this wave's query tile -> dQ =================
  {
    f32x16 aS[2], aP[2];
    float m = -INFINITY;
#pragma unroll
    for (int kj = 0; kj < 2; ++kj) {
      const Row<NPC> kk = lds_row<NPC>(tK, 32 * kj + rho, h), vv = lds_row<NPC>(tV, 32 * kj + rho, h);
      aS[kj] = mma_rows<NPC>(kk, q, splat(0.f));
      aP[kj] = mma_rows<NPC>(vv, dO, splat(0.f));
      float kr[16];
      row_scalars(sKin + 32 * kj, h, kr);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        aS[kj][e] = (32 * kj + c_row(e, h) < n) ? aS[kj][e] * kr[e] : kPadKey;
        m = fmaxf(m, aS[kj][e]);
      }
    }
    m = half_max(m);
    float l = 0.f, Dn = 0.f, E1 = 0.f, E2 = 0.f;
#pragma unroll
    for (int kj = 0; kj < 2; ++kj)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float t = aS[kj][e];
        const float p = __builtin_amdgcn_exp2f((t - m) * qc);
        const float pp = p * aP[kj][e];
        l += p;
        Dn += pp;
        E1 = fmaf(pp, t, E1);
        E2 = fmaf(p, t, E2);
        aS[kj][e] = p;
      }
    l = half_sum(l);
    Dn = half_sum(Dn);
    const float il = __builtin_amdgcn_rcpf(l);
    const float D = Dn * il;
    dtau = -(E1 - D * E2) * il * (qa * inv_tau);
    if (h == 0) {
      sLse[r] = act ? fmaf(m, qc, __builtin_amdgcn_logf(l)) : 1e30f;
      sD[r] = D;
    }
    f32x16 oq = splat(0.f);
#pragma unroll
    for (int kj = 0; kj < 2; ++kj) {
      float kr[16];
      row_scalars(sKin + 32 * kj, h, kr);
#pragma unroll
      for (int e = 0; e < 16; ++e) aS[kj][e] = aS[kj][e] * (aP[kj][e] - D) * (kr[e] * il);
      oq = mma_tokens(tK + 32 * kj * kPitch, aS[kj], oq, lane);
    }
    float qh[NPC][4], pr = 0.f;
#pragma unroll
    for (int t = 0; t < NPC; ++t) {
      piece_f32(q.p[t], qh[t]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        qh[t][j] *= qin;
        pr = fmaf(qh[t][j], oq[4 * t + j], pr);
      }
    }
    pr = half_sum(pr);
    if (act && live) {
      unsigned short* dst = A.dqk + (long long)tok * 2 * d + col;
#pragma unroll
      for (int t = 0; t < NPC; ++t)
        store_piece(dst + 8 * t, (oq[4 * t] - qh[t][0] * pr) * qa, (oq[4 * t + 1] - qh[t][1] * pr) * qa, (oq[4 * t + 2] - qh[t][2] * pr) * qa,
                    (oq[4 * t + 3] - qh[t][3] * pr) * qa);
    }
  }
  // one partial per item: the pair's two query-tile sums meet in LDS (slot sD + 64 is past the 64 D values)
  dtau = gd_wave_sum(dtau);
  float* sPair = sD + 64;
  if (sub == 1 && lane == 0) *sPair = dtau;
  __syncthreads();
  if (sub == 0 && lane == 0 && live) A.dtau_part[item] = dtau + *sPair;
  // ================= phase 2: this wave's key tile -> dK, dV =================
  {
    const float kc = kin * kLog2e;
    f32x16 okk = splat(0.f), ov = splat(0.f);
#pragma unroll
    for (int qi = 0; qi < 2; ++qi) {
      const Row<NPC> qq = lds_row<NPC>(tQ, 32 * qi + rho, h), oo = lds_row<NPC>(tO, 32 * qi + rho, h);
      f32x16 aS = mma_rows<NPC>(qq, k, splat(act ? 0.f : -1e30f));
      f32x16 aP = mma_rows<NPC>(oo, v, splat(0.f));
      float qr[16], lr[16], dr[16];
      row_scalars(sQa + 32 * qi, h, qr);
      row_scalars(sLse + 32 * qi, h, lr);
      row_scalars(sD + 32 * qi, h, dr);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(fmaf(aS[e], qr[e] * kc, -lr[e]));
        aS[e] = p * (aP[e] - dr[e]) * qr[e];
        aP[e] = p;
      }
      okk = mma_tokens(tQ + 32 * qi * kPitch, aS, okk, lane);
      ov = mma_tokens(tO + 32 * qi * kPitch, aP, ov, lane);
    }
    float kh[NPC][4], pr = 0.f;
#pragma unroll
    for (int t = 0; t < NPC; ++t) {
      piece_f32(k.p[t], kh[t]);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kh[t][j] *= kin;
        pr = fmaf(kh[t][j], okk[4 * t + j], pr);
      }
    }
    pr = half_sum(pr);
    if (act && live) {
      unsigned short* dkp = A.dqk + (long long)tok * 2 * d + d + col;
      unsigned short* dvp = A.dv + (long long)tok * d + col;
#pragma unroll
      for (int t = 0; t < NPC; ++t) {
        store_piece(dkp + 8 * t, (okk[4 * t] - kh[t][0] * pr) * kin, (okk[4 * t + 1] - kh[t][1] * pr) * kin, (okk[4 * t + 2] - kh[t][2] * pr) * kin,
                    (okk[4 * t + 3] - kh[t][3] * pr) * kin);
        store_piece(dvp + 8 * t, ov[4 * t], ov[4 * t + 1], ov[4 * t + 2], ov[4 * t + 3]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// kernels: one level alone, or the T = 64 and T = 32 levels of a layer in ONE launch (the T = 64 pairs first - they run
// longest - then the T = 32 items fill in behind them; a level that is alone in its launch costs a full launch floor)
// ---------------------------------------------------------------------------------------------------------------------
template <int NT, int DH>
__global__ __launch_bounds__(256) void k_attn_t32_fwd(T32Args A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
  t32_fwd_body<NT, DH>(A, blockIdx.x, smem_attn);
}
template <int NT, int DH>
__global__ __launch_bounds__(256) void k_attn_t32_bwd(T32BwdArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
  t32_bwd_body<NT, DH>(A, blockIdx.x, smem_attn);
}
template <int DH>
__global__ __launch_bounds__(256) void k_attn_t64_fwd(T32Args A64, T32Args A32, unsigned nb64) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
  if (blockIdx.x < nb64) t64_fwd_body<DH>(A64, blockIdx.x, smem_attn);
  else t32_fwd_body<1, DH>(A32, blockIdx.x - nb64, smem_attn);
}
template <int DH>
__global__ __launch_bounds__(256) void k_attn_t64_bwd(T32BwdArgs A64, T32BwdArgs A32, unsigned nb64) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_attn[];
  if (blockIdx.x < nb64) t64_bwd_body<DH>(A64, blockIdx.x, smem_attn);
  else t32_bwd_body<1, DH>(A32, blockIdx.x - nb64, smem_attn);
}

constexpr int imax(int a, int b) { return a > b ? a : b; }
}  // namespace

// bf16 I/O only; called from attention.hip's entry points.  n32 / n64: windows of the T = 32 / T = 64 levels (either may be 0),
// (win_start, win_len) of each level; both levels of a layer go out as one launch.
int gd_attn_t3264_fwd(const void* qk, const void* v, void* out, const int* csr_tok, const int* ws32, const int* wl32, int n32, const int* ws64,
                      const int* wl64, int n64, int d, int H, const float* tau, float tau_min, hipStream_t st) {
  const T32Args A32{(const unsigned short*)qk, (const unsigned short*)v, (unsigned short*)out, csr_tok, ws32, wl32, n32, d, H, tau, tau_min};
  const T32Args A64{(const unsigned short*)qk, (const unsigned short*)v, (unsigned short*)out, csr_tok, ws64, wl64, n64, d, H, tau, tau_min};
  const int DH = d / H;
  const unsigned nb32 = (unsigned)gd_div_up((long long)n32 * H, 4), nb64 = (unsigned)gd_div_up((long long)n64 * H, 2);
  if (nb32 + nb64 == 0) return 0;
  if (nb64 == 0) {
    if (DH == 16) hipLaunchKernelGGL((k_attn_t32_fwd<1, 16>), dim3(nb32), dim3(256), 4 * fwd_wave_lds<1>(), st, A32);
    else hipLaunchKernelGGL((k_attn_t32_fwd<1, 32>), dim3(nb32), dim3(256), 4 * fwd_wave_lds<1>(), st, A32);
  } else {
    constexpr int lds = imax(2 * kPairFwdLds, 4 * fwd_wave_lds<1>());
    if (DH == 16) hipLaunchKernelGGL((k_attn_t64_fwd<16>), dim3(nb64 + nb32), dim3(256), lds, st, A64, A32, nb64);
    else hipLaunchKernelGGL((k_attn_t64_fwd<32>), dim3(nb64 + nb32), dim3(256), lds, st, A64, A32, nb64);
  }
  GD_LAUNCH_CHECK();
  return 0;
}

// part32 / part64: n32 * H / n64 * H partial slots of d loss / d tau
int gd_attn_t3264_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, const int* csr_tok, const int* ws32,
                      const int* wl32, int n32, float* part32, const int* ws64, const int* wl64, int n64, float* part64, int d, int H,
                      const float* tau, float tau_min, hipStream_t st) {
  const T32BwdArgs A32{(const unsigned short*)qk, (const unsigned short*)v, (const unsigned short*)dout, (unsigned short*)dqk, (unsigned short*)dv,
                       part32, csr_tok, ws32, wl32, n32, d, H, tau, tau_min};
  const T32BwdArgs A64{(const unsigned short*)qk, (const unsigned short*)v, (const unsigned short*)dout, (unsigned short*)dqk, (unsigned short*)dv,
                       part64, csr_tok, ws64, wl64, n64, d, H, tau, tau_min};
  const int DH = d / H;
  const unsigned nb32 = (unsigned)gd_div_up((long long)n32 * H, 4), nb64 = (unsigned)gd_div_up((long long)n64 * H, 2);
  if (nb32 + nb64 == 0) return 0;
  if (nb64 == 0) {
    if (DH == 16) hipLaunchKernelGGL((k_attn_t32_bwd<1, 16>), dim3(nb32), dim3(256), 4 * bwd_wave_lds<1>(), st, A32);
    else hipLaunchKernelGGL((k_attn_t32_bwd<1, 32>), dim3(nb32), dim3(256), 4 * bwd_wave_lds<1>(), st, A32);
  } else {
    constexpr int lds = imax(2 * kPairBwdLds, 4 * bwd_wave_lds<1>());
    if (DH == 16) hipLaunchKernelGGL((k_attn_t64_bwd<16>), dim3(nb64 + nb32), dim3(256), lds, st, A64, A32, nb64);
    else hipLaunchKernelGGL((k_attn_t64_bwd<32>), dim3(nb64 + nb32), dim3(256), lds, st, A64, A32, nb64);
  }
  GD_LAUNCH_CHECK();
  return 0;
}

// one level (the per-level C entry points of attention.hip)
int gd_attn_t32_fwd(const void* qk, const void* v, void* out, const int* csr_tok, const int* win_start, const int* win_len, int n_win,
                    int T, int d, int H, const float* tau, float tau_min, hipStream_t st) {
  if (T == 32) return gd_attn_t3264_fwd(qk, v, out, csr_tok, win_start, win_len, n_win, nullptr, nullptr, 0, d, H, tau, tau_min, st);
  return gd_attn_t3264_fwd(qk, v, out, csr_tok, nullptr, nullptr, 0, win_start, win_len, n_win, d, H, tau, tau_min, st);
}
int gd_attn_t32_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, float* dtau_part, const int* csr_tok,
                    const int* win_start, const int* win_len, int n_win, int T, int d, int H, const float* tau, float tau_min,
                    hipStream_t st) {
  if (T == 32)
    return gd_attn_t3264_bwd(qk, v, dout, dqk, dv, csr_tok, win_start, win_len, n_win, dtau_part, nullptr, nullptr, 0, nullptr, d, H, tau,
                             tau_min, st);
  return gd_attn_t3264_bwd(qk, v, dout, dqk, dv, csr_tok, nullptr, nullptr, 0, nullptr, win_start, win_len, n_win, dtau_part, d, H, tau,
                           tau_min, st);
}
