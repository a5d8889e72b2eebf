// bf16-MFMA variant of the windowed cosine attention for bf16 token I/O (throughput mode), T = 32 / 64 levels.
//
// Same contract and the same wavefront mapping as attention_mfma.hip (one wavefront = one (window, head), S^T layout
// with the query on the lane, accumulators fed back as B operands, S evaluated both ways in the backward), but on
// v_mfma_f32_32x32x16_bf16 (16 k-values per instruction instead of 2):
//  * logits  S = K^ Q^T  keep fp32-grade accuracy by splitting the normalised operands into two bf16 terms,
//    x^ = hi + lo, and accumulating hi.hi + hi.lo + lo.hi in fp32 (error ~2^-16 |S|; the temperature 1/tau <= 100
//    amplifies a plain bf16 product's 2^-9 to O(0.4) in the logits, see attention_mfma.hip) - 3 MFMAs per 16
//    head-dim columns instead of 8 fp32 ones;
//  * dP = dO V^T uses the bf16 rows exactly as they are stored (no conversion, no loss: v and dO ARE bf16);
//  * P V, dS K^, dS^T Q^, P^T dO contract over tokens: the fp32 accumulator registers of P / dS are rounded to bf16
//    and fed back as the B operand (8 registers = one k-group of 16 keys per half-wave pair), the A operand is
//    read from a transposed bf16 LDS tile [head-dim][token] as two 8-byte words that hold exactly the 8 tokens the
//    B registers of that lane-half map to (am_row: rows 16t + 4 half + {0..3} and + 8).
// Per (window, head) backward at T = 64, DH = 32: 88 bf16 MFMAs (2.8 k cycles) instead of 448 fp32 ones (28.7 k).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define AH_EPS 1e-12f
__device__ __forceinline__ float ah_exp(float x) { return __expf(x); }
__device__ __forceinline__ int ah_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

union AhFrag {
  uint4 u;
  uint2 u2[2];
  bf16x8 v;
  unsigned short s[8];
};

__device__ __forceinline__ float ah_bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short ah_f2bf(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7F800000u) == 0x7F800000u) return (unsigned short)(u >> 16);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
// 8 accumulator registers [r0, r0 + 8) -> bf16 B fragment
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 ah_pack(const f32x16& a, int r0) {
  f32x8 t;
#pragma unroll
  for (int j = 0; j < 8; ++j) t[j] = a[r0 + j];
  return __builtin_convertvector(t, bf16x8);     // v_cvt_pk_bf16_f32 (round to nearest even)
}
__device__ __forceinline__ f32x16 ah_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// A fragment of a token-contracted product: row `row` (head-dim index) of the transposed tile, the 8 tokens
// base + 4 half + {0..3} and base + 8 + 4 half + {0..3}
__device__ __forceinline__ bf16x8 ah_lds_a(const unsigned short* __restrict__ sT, int row, int ldt, int base, int half, bool valid) {
  AhFrag f;
  if (valid) {
    f.u2[0] = *reinterpret_cast<const uint2*>(sT + row * ldt + base + 4 * half);
    f.u2[1] = *reinterpret_cast<const uint2*>(sT + row * ldt + base + 8 + 4 * half);
  } else {
    f.u = make_uint4(0, 0, 0, 0);
  }
  return f.v;
}

struct AhArgs {
  const void* qk;
  const void* v;
  void* out;
  const int* csr_tok;
  const int* win_start;
  const int* win_len;
  int n_win, d, H;
  const float* tau;
  float tau_min;
};

// normalise a row held as KS fragments (this lane's 8 * KS columns; the partner half-wave holds the rest), split hi/lo
// `post`: extra factor folded into the split operands (1 / tau for the queries: the MFMA then yields the logits
// directly and no per-element scaling is needed); the returned 1/|x| is the plain normalisation factor
template <int KS>
__device__ __forceinline__ float ah_normalize(const AhFrag (&raw)[KS], AhFrag (&hi)[KS], AhFrag (&lo)[KS], float post = 1.f) {
  float ss = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = ah_bf2f(raw[s].s[j]);
      ss = fmaf(x, x, ss);
    }
  ss += __shfl_xor(ss, 32, 64);
  const float inv = 1.f / fmaxf(sqrtf(ss), AH_EPS);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    f32x8 x, r;
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = ah_bf2f(raw[s].s[j]) * (inv * post);
    hi[s].v = __builtin_convertvector(x, bf16x8);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = x[j] - ah_bf2f(hi[s].s[j]);
    lo[s].v = __builtin_convertvector(r, bf16x8);
  }
  return inv;
}

// transposed store of this lane's fragments: sT[dh][token r], dh = 16 s + 8 half + j
template <int KS>
__device__ __forceinline__ void ah_store_t(unsigned short* __restrict__ sT, int ldt, int r, int half, const AhFrag (&f)[KS]) {
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) sT[(16 * s + 8 * half + j) * ldt + r] = f[s].s[j];
}

template <int NT, int DH>
__global__ __launch_bounds__(256) void k_attn_mfma16_fwd(AhArgs A) {
  constexpr int KS = DH / 16;
  constexpr int LDT = 32 * NT + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int rho = lane & 31, half = lane >> 5;
  unsigned short* sV = smem16 + wib * (DH * LDT);
  const long long item = (long long)blockIdx.x * 4 + wib;
  if (item >= (long long)A.n_win * A.H) return;
  const int w = (int)(item / A.H), h = (int)(item % A.H);
  const int n = A.win_len[w], start = A.win_start[w];
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);
  const unsigned short* gqk = (const unsigned short*)A.qk;
  const unsigned short* gv = (const unsigned short*)A.v;
  unsigned short* gout = (unsigned short*)A.out;
  const int d = A.d;

  AhFrag qhi[NT][KS], qlo[NT][KS], khi[NT][KS], klo[NT][KS];
  int tok[NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int r = 32 * ti + rho;
    const bool act = r < n;
    tok[ti] = act ? A.csr_tok[start + r] : 0;
    AhFrag qr[KS], kr[KS], vr[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (act) {
        qr[s].u = *reinterpret_cast<const uint4*>(gqk + (long long)tok[ti] * 2 * d + h * DH + 16 * s + 8 * half);
        kr[s].u = *reinterpret_cast<const uint4*>(gqk + (long long)tok[ti] * 2 * d + d + h * DH + 16 * s + 8 * half);
        vr[s].u = *reinterpret_cast<const uint4*>(gv + (long long)tok[ti] * d + h * DH + 16 * s + 8 * half);
      } else {
        qr[s].u = kr[s].u = vr[s].u = make_uint4(0, 0, 0, 0);
      }
    }
    ah_normalize<KS>(qr, qhi[ti], qlo[ti], inv_tau);       // logits = (q^ / tau) . k^
    ah_normalize<KS>(kr, khi[ti], klo[ti]);
    ah_store_t<KS>(sV, LDT, r, half, vr);
  }
  // S^T[key][query] tiles
  // padded keys start at -1e30 instead of 0: exp() of their logits is exactly 0, no per-element masking afterwards
  f32x16 acc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = (32 * a + ah_row(r, half) < n) ? 0.f : -1e30f;
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
      for (int qi = 0; qi < NT; ++qi) {
        acc[kj][qi] = ah_mfma(khi[kj][s].v, qhi[qi][s].v, acc[kj][qi]);
        acc[kj][qi] = ah_mfma(khi[kj][s].v, qlo[qi][s].v, acc[kj][qi]);
        acc[kj][qi] = ah_mfma(klo[kj][s].v, qhi[qi][s].v, acc[kj][qi]);
      }
  float linv[NT];
#pragma unroll
  for (int qi = 0; qi < NT; ++qi) {
    float m = -INFINITY;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[kj][qi][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = ah_exp(acc[kj][qi][r] - m);
        acc[kj][qi][r] = p;
        l += p;
      }
    l += __shfl_xor(l, 32, 64);
    linv[qi] = 1.f / l;
  }
  __builtin_amdgcn_wave_barrier();
  // O^T[dh][query] = sum_keys V^T[dh][key] P^T[key][query]
  f32x16 o[NT];
#pragma unroll
  for (int qi = 0; qi < NT; ++qi)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[qi][r] = 0.f;
#pragma unroll
  for (int kj = 0; kj < NT; ++kj)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bf16x8 a = ah_lds_a(sV, rho, LDT, 32 * kj + 16 * t, half, rho < DH);
#pragma unroll
      for (int qi = 0; qi < NT; ++qi) o[qi] = ah_mfma(a, ah_pack(acc[kj][qi], 8 * t), o[qi]);
    }
#pragma unroll
  for (int qi = 0; qi < NT; ++qi) {
    if (32 * qi + rho < n) {
      unsigned short* dst = gout + (long long)tok[qi] * d + h * DH;
#pragma unroll
      for (int g = 0; g < DH / 8; ++g) {
        uint2 v;
        v.x = ah_f2bf(o[qi][4 * g] * linv[qi]) | ((unsigned)ah_f2bf(o[qi][4 * g + 1] * linv[qi]) << 16);
        v.y = ah_f2bf(o[qi][4 * g + 2] * linv[qi]) | ((unsigned)ah_f2bf(o[qi][4 * g + 3] * linv[qi]) << 16);
        *reinterpret_cast<uint2*>(dst + 8 * g + 4 * half) = v;
      }
    }
  }
}

struct AhBwdArgs {
  const void* qk;
  const void* v;
  const void* dout;
  void* dqk;
  void* dv;
  float* dtau_part;
  const int* csr_tok;
  const int* win_start;
  const int* win_len;
  int n_win, d, H;
  const float* tau;
  float tau_min;
};

__device__ __forceinline__ void ah_load4(const unsigned short* __restrict__ p, float (&r)[4]) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  r[0] = __uint_as_float(v.x << 16);
  r[1] = __uint_as_float(v.x & 0xFFFF0000u);
  r[2] = __uint_as_float(v.y << 16);
  r[3] = __uint_as_float(v.y & 0xFFFF0000u);
}
__device__ __forceinline__ void ah_store4(unsigned short* __restrict__ p, float a, float b, float c, float d) {
  uint2 v;
  v.x = ah_f2bf(a) | ((unsigned)ah_f2bf(b) << 16);
  v.y = ah_f2bf(c) | ((unsigned)ah_f2bf(d) << 16);
  *reinterpret_cast<uint2*>(p) = v;
}

template <int NT, int DH>
__global__ __launch_bounds__(256) void k_attn_mfma16_bwd(AhBwdArgs A) {
  constexpr int KS = DH / 16;
  constexpr int LDT = 32 * NT + 4;
  constexpr int TILE = DH * LDT;                 // bf16 elements
  extern __shared__ __attribute__((aligned(16))) unsigned short smem16[];
  const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
  const int rho = lane & 31, half = lane >> 5;
  unsigned short* sA = smem16 + wib * (2 * TILE + 4 * 64);   // K^^T (phase 1) / Q^^T (phase 2)
  unsigned short* sB = sA + TILE;                              // dO^T (phase 2)
  float* sLse = reinterpret_cast<float*>(sB + TILE);
  float* sD = sLse + 64;
  const long long item = (long long)blockIdx.x * 4 + wib;
  if (item >= (long long)A.n_win * A.H) return;
  const int w = (int)(item / A.H), h = (int)(item % A.H);
  const int n = A.win_len[w], start = A.win_start[w];
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);
  const unsigned short* gqk = (const unsigned short*)A.qk;
  const unsigned short* gv = (const unsigned short*)A.v;
  const unsigned short* gdo = (const unsigned short*)A.dout;
  unsigned short* gdqk = (unsigned short*)A.dqk;
  unsigned short* gdv = (unsigned short*)A.dv;
  const int d = A.d;

  AhFrag qhi[NT][KS], qlo[NT][KS], khi[NT][KS], klo[NT][KS], vf[NT][KS], dof[NT][KS];
  float qin[NT], kin[NT];
  int tok[NT];
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int r = 32 * ti + rho;
    const bool act = r < n;
    tok[ti] = act ? A.csr_tok[start + r] : 0;
    AhFrag qr[KS], kr[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (act) {
        qr[s].u = *reinterpret_cast<const uint4*>(gqk + (long long)tok[ti] * 2 * d + h * DH + 16 * s + 8 * half);
        kr[s].u = *reinterpret_cast<const uint4*>(gqk + (long long)tok[ti] * 2 * d + d + h * DH + 16 * s + 8 * half);
        vf[ti][s].u = *reinterpret_cast<const uint4*>(gv + (long long)tok[ti] * d + h * DH + 16 * s + 8 * half);
        dof[ti][s].u = *reinterpret_cast<const uint4*>(gdo + (long long)tok[ti] * d + h * DH + 16 * s + 8 * half);
      } else {
        qr[s].u = kr[s].u = vf[ti][s].u = dof[ti][s].u = make_uint4(0, 0, 0, 0);
      }
    }
    qin[ti] = ah_normalize<KS>(qr, qhi[ti], qlo[ti], inv_tau);     // logits = (q^ / tau) . k^
    kin[ti] = ah_normalize<KS>(kr, khi[ti], klo[ti]);
    ah_store_t<KS>(sA, LDT, r, half, khi[ti]);     // K^^T tile: A operand of dQ^T
  }

  float dtau = 0.f;
  __builtin_amdgcn_wave_barrier();
  // ================= phase 1: query on the lane (S^T, dP^T) -> dQ, one query tile at a time =================
#pragma unroll
  for (int qi = 0; qi < NT; ++qi) {
    f32x16 aS[NT], aP[NT];
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        aS[kj][r] = (32 * kj + ah_row(r, half) < n) ? 0.f : -1e30f;    // padded keys: exp() = 0 without masking
        aP[kj][r] = 0.f;
      }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int kj = 0; kj < NT; ++kj) {
        aS[kj] = ah_mfma(khi[kj][s].v, qhi[qi][s].v, aS[kj]);
        aS[kj] = ah_mfma(khi[kj][s].v, qlo[qi][s].v, aS[kj]);
        aS[kj] = ah_mfma(klo[kj][s].v, qhi[qi][s].v, aS[kj]);
        aP[kj] = ah_mfma(vf[kj][s].v, dof[qi][s].v, aP[kj]);
      }
    float m = -INFINITY;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, aS[kj][r]);
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f, Dn = 0.f, E1 = 0.f, E2 = 0.f;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a = aS[kj][r];
        const float e = ah_exp(a - m);                   // exactly 0 for padded keys (a = -1e30: finite, 0 * a = -0)
        l += e;
        Dn = fmaf(e, aP[kj][r], Dn);
        E1 = fmaf(e * aP[kj][r], a, E1);
        E2 = fmaf(e, a, E2);
        aS[kj][r] = e;
      }
    l += __shfl_xor(l, 32, 64);
    Dn += __shfl_xor(Dn, 32, 64);
    const float il = 1.f / l;
    const float lse = m + logf(l);
    const float D = Dn * il;
    const bool qact = 32 * qi + rho < n;
    if (qact) dtau -= (E1 - D * E2) * il * inv_tau;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
      for (int r = 0; r < 16; ++r) aS[kj][r] = aS[kj][r] * il * (aP[kj][r] - D);   // dS (the 1/tau factor is applied to dq / dk)
    if (half == 0) {
      sLse[32 * qi + rho] = qact ? lse : 1e30f;       // padded queries: exp(a - 1e30) = 0 in phase 2 without masking
      sD[32 * qi + rho] = D;
    }
    f32x16 oq;
#pragma unroll
    for (int r = 0; r < 16; ++r) oq[r] = 0.f;
#pragma unroll
    for (int kj = 0; kj < NT; ++kj)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        oq = ah_mfma(ah_lds_a(sA, rho, LDT, 32 * kj + 16 * t, half, rho < DH), ah_pack(aS[kj], 8 * t), oq);
    float qv[DH / 8][4];
    float pr = 0.f;
#pragma unroll
    for (int g = 0; g < DH / 8; ++g) {
      if (qact) ah_load4(gqk + (long long)tok[qi] * 2 * d + h * DH + 8 * g + 4 * half, qv[g]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qv[g][e] = qact ? qv[g][e] * qin[qi] : 0.f;
        pr = fmaf(qv[g][e], oq[4 * g + e], pr);
      }
    }
    pr += __shfl_xor(pr, 32, 64);
    if (qact) {
      unsigned short* dst = gdqk + (long long)tok[qi] * 2 * d + h * DH;
      const float sc = qin[qi] * inv_tau;
#pragma unroll
      for (int g = 0; g < DH / 8; ++g)
        ah_store4(dst + 8 * g + 4 * half, (oq[4 * g] - qv[g][0] * pr) * sc, (oq[4 * g + 1] - qv[g][1] * pr) * sc,
                  (oq[4 * g + 2] - qv[g][2] * pr) * sc, (oq[4 * g + 3] - qv[g][3] * pr) * sc);
    }
  }
  // ================= phase 2: key on the lane (S, dP) -> dK, dV, one key tile at a time =================
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int ti = 0; ti < NT; ++ti) {
    const int r = 32 * ti + rho;
    ah_store_t<KS>(sA, LDT, r, half, qhi[ti]);    // Q^^T
    ah_store_t<KS>(sB, LDT, r, half, dof[ti]);    // dO^T
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int kj = 0; kj < NT; ++kj) {
    const bool kact = 32 * kj + rho < n;
    const float ini = kact ? 0.f : -1e30f;              // padded key (lane): probabilities exactly 0
    f32x16 aS[NT], aP[NT];
#pragma unroll
    for (int qi = 0; qi < NT; ++qi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        aS[qi][r] = ini;
        aP[qi][r] = 0.f;
      }
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int qi = 0; qi < NT; ++qi) {
        aS[qi] = ah_mfma(qhi[qi][s].v, khi[kj][s].v, aS[qi]);    // S[q][key]
        aS[qi] = ah_mfma(qhi[qi][s].v, klo[kj][s].v, aS[qi]);
        aS[qi] = ah_mfma(qlo[qi][s].v, khi[kj][s].v, aS[qi]);
        aP[qi] = ah_mfma(dof[qi][s].v, vf[kj][s].v, aP[qi]);     // dP[q][key]
      }
#pragma unroll
    for (int qi = 0; qi < NT; ++qi) {
      // row statistics of this lane-half's 16 query rows: read unconditionally (all 32 NT rows were written in phase 1;
      // a conditional read costs an exec-mask branch per element)
      float lq[16], dq[16];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 a = *reinterpret_cast<const float4*>(sLse + 32 * qi + 8 * g + 4 * half);
        const float4 b = *reinterpret_cast<const float4*>(sD + 32 * qi + 8 * g + 4 * half);
        lq[4 * g] = a.x; lq[4 * g + 1] = a.y; lq[4 * g + 2] = a.z; lq[4 * g + 3] = a.w;
        dq[4 * g] = b.x; dq[4 * g + 1] = b.y; dq[4 * g + 2] = b.z; dq[4 * g + 3] = b.w;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = ah_exp(aS[qi][r] - lq[r]);       // 0 for padded keys (logit -1e30) and padded queries (lse +1e30)
        aS[qi][r] = p * (aP[qi][r] - dq[r]);             // dS
        aP[qi][r] = p;
      }
    }
    f32x16 okk, ov;
#pragma unroll
    for (int r = 0; r < 16; ++r) okk[r] = ov[r] = 0.f;
#pragma unroll
    for (int qi = 0; qi < NT; ++qi)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        okk = ah_mfma(ah_lds_a(sA, rho, LDT, 32 * qi + 16 * t, half, rho < DH), ah_pack(aS[qi], 8 * t), okk);
        ov = ah_mfma(ah_lds_a(sB, rho, LDT, 32 * qi + 16 * t, half, rho < DH), ah_pack(aP[qi], 8 * t), ov);
      }
    float kv[DH / 8][4];
    float pr = 0.f;
#pragma unroll
    for (int g = 0; g < DH / 8; ++g) {
      if (kact) ah_load4(gqk + (long long)tok[kj] * 2 * d + d + h * DH + 8 * g + 4 * half, kv[g]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        kv[g][e] = kact ? kv[g][e] * kin[kj] : 0.f;
        pr = fmaf(kv[g][e], okk[4 * g + e], pr);
      }
    }
    pr += __shfl_xor(pr, 32, 64);
    if (kact) {
      unsigned short* dk = gdqk + (long long)tok[kj] * 2 * d + d + h * DH;
      unsigned short* dvp = gdv + (long long)tok[kj] * d + h * DH;
      const float ksc = kin[kj];                  // the 1/tau factor of dk is already in the Q^^T operand (q^ / tau)
#pragma unroll
      for (int g = 0; g < DH / 8; ++g) {
        ah_store4(dk + 8 * g + 4 * half, (okk[4 * g] - kv[g][0] * pr) * ksc, (okk[4 * g + 1] - kv[g][1] * pr) * ksc,
                  (okk[4 * g + 2] - kv[g][2] * pr) * ksc, (okk[4 * g + 3] - kv[g][3] * pr) * ksc);
        ah_store4(dvp + 8 * g + 4 * half, ov[4 * g], ov[4 * g + 1], ov[4 * g + 2], ov[4 * g + 3]);
      }
    }
  }
  dtau = gd_wave_sum(dtau);
  if (lane == 0) A.dtau_part[item] = dtau;
}

template <int NT, int DH>
static int ah_launch_fwd(const AhArgs& A, hipStream_t st) {
  const long long items = (long long)A.n_win * A.H;
  const size_t lds = (size_t)4 * DH * (32 * NT + 4) * sizeof(unsigned short);
  hipLaunchKernelGGL((k_attn_mfma16_fwd<NT, DH>), dim3(gd_div_up(items, 4)), dim3(256), lds, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}
template <int NT, int DH>
static int ah_launch_bwd(const AhBwdArgs& A, hipStream_t st) {
  const long long items = (long long)A.n_win * A.H;
  const size_t lds = (size_t)4 * (2 * DH * (32 * NT + 4) + 4 * 64) * sizeof(unsigned short);
  hipLaunchKernelGGL((k_attn_mfma16_bwd<NT, DH>), dim3(gd_div_up(items, 4)), dim3(256), lds, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}

// bf16 I/O only; called from attention.hip's dispatchers for T = 32 (NT = 1) and T = 64 (NT = 2)
int gd_attn_mfma16_fwd(const void* qk, const void* v, void* out, const int* csr_tok, const int* win_start, const int* win_len,
                       int n_win, int T, int d, int H, const float* tau, float tau_min, hipStream_t st) {
  AhArgs A{qk, v, out, csr_tok, win_start, win_len, n_win, d, H, tau, tau_min};
  const int DH = d / H;
  if (T == 32) return DH == 16 ? ah_launch_fwd<1, 16>(A, st) : ah_launch_fwd<1, 32>(A, st);
  return DH == 16 ? ah_launch_fwd<2, 16>(A, st) : ah_launch_fwd<2, 32>(A, st);
}

int gd_attn_mfma16_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, float* dtau_part, const int* csr_tok,
                       const int* win_start, const int* win_len, int n_win, int T, int d, int H, const float* tau, float tau_min,
                       hipStream_t st) {
  AhBwdArgs A{qk, v, dout, dqk, dv, dtau_part, csr_tok, win_start, win_len, n_win, d, H, tau, tau_min};
  const int DH = d / H;
  if (T == 32) return DH == 16 ? ah_launch_bwd<1, 16>(A, st) : ah_launch_bwd<1, 32>(A, st);
  return DH == 16 ? ah_launch_bwd<2, 16>(A, st) : ah_launch_bwd<2, 32>(A, st);
}
