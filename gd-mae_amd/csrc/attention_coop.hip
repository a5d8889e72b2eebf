// Workgroup-cooperative bf16-MFMA windowed cosine attention, T = 32 / 64 occupancy levels (throughput mode; round 5).
//
// Contract and arithmetic: reference cosine_msa.py:114-176 / sst_basic_block.py:22-54 through the window CSR, exactly as
// attention_t32.hip (raw-operand logits normalised on the accumulator, S^T layout for the forward and dQ, S layout for dK / dV, no
// atomics).  What is new is how rows travel (attn_tiles.h): a workgroup = one window x HW heads whose row segments are contiguous
// (T = 64: 2 heads x 2 wavefronts, T = 32: 4 heads x 1 wavefront); all 256 threads load the q / k / v (/ dO) segments of the window's rows
// 16 bytes per lane - whole cache lines - into the swizzled LDS tiles, the wavefronts take their MFMA operand pieces from there, and the
// results go back through the same tiles as 16-byte row segments.  The row norms are taken in the cooperative pass (adjacent lanes
// hold a head's chunks), the window descriptor is a scalar load (the window is uniform per workgroup).
#include "attn_tiles.h"
#include "attn16_wave.h"
#include <stdlib.h>

using namespace attn;

#ifndef ATTN_K64
#define ATTN_K64 1      // head groups (2 heads each) of a T = 64 window per workgroup
#endif
#ifndef ATTN_K32
#define ATTN_K32 1      // head groups (4 heads each) of a T = 32 window per workgroup
#endif

namespace {
struct CArgs {
  const unsigned short* qk;
  const unsigned short* v;
  unsigned short* out;
  float* lse;                   // (n_tok, H) log2 sum exp of a row's logits (for the backward), or null
  const int* csr_tok;
  const int* win_start;
  const int* win_len;
  int n_win, d, H;
  const float* tau;
  float tau_min;
};
struct CBwdArgs {
  const unsigned short* qk;
  const unsigned short* v;
  const unsigned short* o;      // the forward's output rows
  const float* lse;             // the forward's lse
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

// LDS per head: tiles of ROWS x 64 B + per-row scalars
template <int ROWS>
constexpr int fwd_head_lds() { return 3 * ROWS * kPitch * 2 + 3 * ROWS * 4; }           // Q, K, V | 1 / |k|, 1 / |q|, lse
template <int ROWS>
constexpr int bwd_head_lds() { return 4 * ROWS * kPitch * 2 + 5 * ROWS * 4 + 16; }      // K, V, Q, dO | kin, qa, lse, D, qin | dtau slot

// ---------------------------------------------------------------------------------------------------------------------
// forward.  NW = wavefronts per head (T = 64: 2, each owns a 32-query tile; T = 32: 1), HW = heads per workgroup = 4 / NW
// ---------------------------------------------------------------------------------------------------------------------
// K = head groups of the window a workgroup takes, one after the other: the row segments of group g + 1 (the neighbouring bytes of the
// same token rows) are requested when group g's tiles are complete and stay in flight under its products and stores (registers: the
// loads cross the barriers, s_barrier does not drain vmcnt on gfx950).
template <int DH, int NW, int K>
__device__ __forceinline__ void coop_fwd(const CArgs& A, const unsigned blk, unsigned char* __restrict__ smem) {
  constexpr int NPC = DH / 8, HW = 4 / NW, ROWS = 32 * NW;
  using G = Coop<DH, HW, ROWS>;
  constexpr int HL = fwd_head_lds<ROWS>();
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  const int rho = lane & 31, h = lane >> 5, sub = wib % NW, hw = wib / NW;
  const int HG = A.H / (HW * K);                // workgroups per window
  const int w = blk / HG, hg0 = (blk - w * HG) * K;
  const int n = A.win_len[w], start = A.win_start[w];
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);
  const int d = A.d;
  // ---- cooperative load: q / k / v segments of the window's rows -> tiles, norms -> scalar rows
  const int lr = tid / G::LPR, ch = tid % G::LPR, chl = ch / G::CPH, cq = ch % G::CPH;
  unsigned short* cQ = reinterpret_cast<unsigned short*>(smem + chl * HL);
  unsigned short* cK = cQ + ROWS * kPitch;
  unsigned short* cV = cK + ROWS * kPitch;
  float* cKin = reinterpret_cast<float*>(cV + ROWS * kPitch);
  float* cQn = cKin + ROWS;
  int col = hg0 * HW * DH + ch * 8;
  int tok[G::P];
#pragma unroll
  for (int p = 0; p < G::P; ++p) {
    const int r = p * G::RPI + lr;
    const int rc = r < n ? r : n - 1;
    tok[p] = A.csr_tok ? A.csr_tok[start + (rc > 0 ? rc : 0)] : start + (rc > 0 ? rc : 0);       // slots past the window repeat its last token; cleared below (null: rows in window-major order)
  }
  uint4 q16[G::P], k16[G::P], v16[G::P];
#pragma unroll
  for (int p = 0; p < G::P; ++p) {
    const unsigned short* qp = A.qk + (long long)tok[p] * 2 * d + col;
    q16[p] = *reinterpret_cast<const uint4*>(qp);
    k16[p] = *reinterpret_cast<const uint4*>(qp + d);
    v16[p] = *reinterpret_cast<const uint4*>(A.v + (long long)tok[p] * d + col);
  }
#pragma unroll
 for (int it = 0; it < K; ++it) {
  const int hg = hg0 + it;
  if (it > 0) __syncthreads();              // the previous group's stores have read the tiles
#pragma unroll
  for (int p = 0; p < G::P; ++p) {
    const int r = p * G::RPI + lr;
    if (G::RPI * G::P > ROWS && r >= ROWS) continue;
    const unsigned m = r < n ? 0xFFFFFFFFu : 0u;
    q16[p] = and16(q16[p], m);
    k16[p] = and16(k16[p], m);
    v16[p] = and16(v16[p], m);
    const float qn = inv_norm_chunks<G::CPH>(ssq16(q16[p]));
    const float kn = inv_norm_chunks<G::CPH>(ssq16(k16[p]));
    if (cq == 0) {
      cQn[r] = qn;
      cKin[r] = kn;
    }
    tile_put16(cQ, r, cq, q16[p]);
    tile_put16(cK, r, cq, k16[p]);
    tile_put16(cV, r, cq, v16[p]);
  }
  __syncthreads();
  const int col_st = col;
  if (it + 1 < K) {                         // the next head group's segments: in flight until the top of the next iteration
    col += HW * DH;
#pragma unroll
    for (int p = 0; p < G::P; ++p) {
      const unsigned short* qp = A.qk + (long long)tok[p] * 2 * d + col;
      q16[p] = *reinterpret_cast<const uint4*>(qp);
      k16[p] = *reinterpret_cast<const uint4*>(qp + d);
      v16[p] = *reinterpret_cast<const uint4*>(A.v + (long long)tok[p] * d + col);
    }
  }
  // ---- this wavefront: head hw, query tile sub
  unsigned short* tQ = reinterpret_cast<unsigned short*>(smem + hw * HL);
  unsigned short* tK = tQ + ROWS * kPitch;
  unsigned short* tV = tK + ROWS * kPitch;
  const float* sKin = reinterpret_cast<const float*>(tV + ROWS * kPitch);
  const float* sQn = sKin + ROWS;
  float* sLse = reinterpret_cast<float*>(tV + ROWS * kPitch) + 2 * ROWS;
  const int r = 32 * sub + rho;
#ifndef ATTN_EXP_NOCOMP      // experiment: loads -> tiles -> stores only (what the arithmetic costs on top of the row traffic)
  const Row<NPC> q = lds_row<NPC>(tQ, r, h);
  const float qc = sQn[r] * inv_tau * kLog2e;              // (1 / |q| tau) log2(e): exponent scale of this lane's query column
  f32x16 aS[NW];
  float m = -INFINITY;
#pragma unroll
  for (int kj = 0; kj < NW; ++kj) {
    const Row<NPC> kk = lds_row<NPC>(tK, 32 * kj + rho, h);
    aS[kj] = mma_rows<NPC>(kk, q, splat(0.f));            // S^T[key][query], raw dot products
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
  for (int kj = 0; kj < NW; ++kj)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float p = __builtin_amdgcn_exp2f((aS[kj][e] - m) * qc);
      aS[kj][e] = p;
      l += p;
    }
  l = half_sum(l);
  const float il = __builtin_amdgcn_rcpf(l);
  if (h == 0) sLse[r] = fmaf(m, qc, __builtin_amdgcn_logf(l));      // log2 units: p = exp2(t qc - lse)
  f32x16 o = splat(0.f);
#pragma unroll
  for (int kj = 0; kj < NW; ++kj) o = mma_tokens(tV + 32 * kj * kPitch, aS[kj], o, lane);      // O^T[dh][query]
  // ---- results -> own rows of the Q tile (this wavefront is its only reader) -> cooperative store
  {
    Row<NPC> ob;
#pragma unroll
    for (int t = 0; t < NPC; ++t) ob.p[t] = pack_piece(o[4 * t] * il, o[4 * t + 1] * il, o[4 * t + 2] * il, o[4 * t + 3] * il);
    store_tile<NPC>(tQ, r, h, ob);
  }
#else
  (void)r; (void)sKin; (void)sQn; (void)sLse; (void)tK; (void)inv_tau;
#endif
  __syncthreads();
#pragma unroll
  for (int p = 0; p < G::P; ++p) {
    const int r2 = p * G::RPI + lr;
    if (r2 < n && r2 < ROWS) {
      *reinterpret_cast<uint4*>(A.out + (long long)tok[p] * d + col_st) = tile_get16(cQ, r2, cq);
      if (cq == 0 && A.lse) A.lse[(long long)tok[p] * A.H + hg * HW + chl] = (cKin + 2 * ROWS)[r2];
    }
  }
 }
}

// ---------------------------------------------------------------------------------------------------------------------
// backward.  The forward left lse[token][head] = log2 sum_k exp(logit) (coop_fwd), and D[query] = sum_k P dP = dO[query] . O[query] is a
// row dot product of the saved attention output - so nothing about a query has to be reduced over the keys here: a probability is
// p = exp2(logit log2(e) - lse) wherever it is needed.  Phase 1 (this wave's 32 queries, S^T layout) walks the key tiles one at a time
// with 32 instead of 64 accumulator registers live; phase 2 (this wave's 32 keys, S layout) needs nothing from phase 1.  <= 128
// registers: four wavefronts per SIMD (the two-pass form needed 197), and the sparse level can share the launch.
// ---------------------------------------------------------------------------------------------------------------------
template <int DH, int NW>
__device__ __forceinline__ void coop_bwd(const CBwdArgs& A, const unsigned blk, unsigned char* __restrict__ smem) {
  constexpr int NPC = DH / 8, HW = 4 / NW, ROWS = 32 * NW;
  using G = Coop<DH, HW, ROWS>;
  constexpr int HL = bwd_head_lds<ROWS>();
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  const int rho = lane & 31, h = lane >> 5, sub = wib % NW, hw = wib / NW;
  const int HG = A.H / HW;
  const int w = blk / HG, hg = blk - w * HG;
  const int n = A.win_len[w], start = A.win_start[w];
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);
  const int d = A.d;
  // ---- cooperative load
  const int lr = tid / G::LPR, ch = tid % G::LPR, chl = ch / G::CPH, cq = ch % G::CPH;
  unsigned short* cK = reinterpret_cast<unsigned short*>(smem + chl * HL);
  unsigned short* cV = cK + ROWS * kPitch;
  unsigned short* cQ = cV + ROWS * kPitch;
  unsigned short* cO = cQ + ROWS * kPitch;
  float* cKin = reinterpret_cast<float*>(cO + ROWS * kPitch);
  float* cQa = cKin + ROWS;
  float* cLse = cKin + 2 * ROWS;
  float* cD = cKin + 3 * ROWS;
  float* cQn = cKin + 4 * ROWS;
  const int col = hg * HW * DH + ch * 8;
  int tok[G::P];
#pragma unroll
  for (int p = 0; p < G::P; ++p) {
    const int r = p * G::RPI + lr;
    const int rc = r < n ? r : n - 1;
    tok[p] = A.csr_tok ? A.csr_tok[start + (rc > 0 ? rc : 0)] : start + (rc > 0 ? rc : 0);
  }
  uint4 q16[G::P], k16[G::P], v16[G::P], o16[G::P], x16[G::P];
  float lse_[G::P];
#pragma unroll
  for (int p = 0; p < G::P; ++p) {
    const unsigned short* qp = A.qk + (long long)tok[p] * 2 * d + col;
    q16[p] = *reinterpret_cast<const uint4*>(qp);
    k16[p] = *reinterpret_cast<const uint4*>(qp + d);
    v16[p] = *reinterpret_cast<const uint4*>(A.v + (long long)tok[p] * d + col);
    o16[p] = *reinterpret_cast<const uint4*>(A.dout + (long long)tok[p] * d + col);
    x16[p] = *reinterpret_cast<const uint4*>(A.o + (long long)tok[p] * d + col);
    lse_[p] = A.lse[(long long)tok[p] * A.H + hg * HW + chl];
  }
#pragma unroll
  for (int p = 0; p < G::P; ++p) {
    const int r = p * G::RPI + lr;
    if (G::RPI * G::P > ROWS && r >= ROWS) continue;
    const unsigned m = r < n ? 0xFFFFFFFFu : 0u;
    q16[p] = and16(q16[p], m);
    k16[p] = and16(k16[p], m);
    v16[p] = and16(v16[p], m);
    o16[p] = and16(o16[p], m);
    const float qn = inv_norm_chunks<G::CPH>(ssq16(q16[p]));
    const float kn = inv_norm_chunks<G::CPH>(ssq16(k16[p]));
    float dd = dot16(o16[p], x16[p]);                     // D = dO . O over the head's channels (0 for padded rows: dO cleared)
    dd += gd_dpp_mov<0xB1>(dd);
    if (G::CPH >= 4) dd += gd_dpp_mov<0x4E>(dd);
    if (cq == 0) {
      cQn[r] = qn;
      cQa[r] = qn * inv_tau;
      cKin[r] = kn;
      cLse[r] = r < n ? lse_[p] : 1e30f;                  // padded query: p = exp2(. - 1e30) = 0 everywhere
      cD[r] = dd;                                         // (enters the dP accumulators negated)
    }
    tile_put16(cQ, r, cq, q16[p]);
    tile_put16(cK, r, cq, k16[p]);
    tile_put16(cV, r, cq, v16[p]);
    tile_put16(cO, r, cq, o16[p]);
  }
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
  // ---- this wavefront: head hw, query tile sub (phase 1), key tile sub (phase 2)
  unsigned short* tK = reinterpret_cast<unsigned short*>(smem + hw * HL);
  unsigned short* tV = tK + ROWS * kPitch;
  unsigned short* tQ = tV + ROWS * kPitch;
  unsigned short* tO = tQ + ROWS * kPitch;
  float* sKin = reinterpret_cast<float*>(tO + ROWS * kPitch);
  float* sQa = sKin + ROWS;
  float* sLse = sQa + ROWS;          // log2 units
  float* sD = sLse + ROWS;
  const float* sQn = sD + ROWS;
  float* sPair = sKin + 5 * ROWS;
  const int r = 32 * sub + rho;
  const bool act = r < n;
  float dtau;
  Row<NPC> dqr;
#ifdef ATTN_EXP_NOCOMP
  Row<NPC> dkr, dvr;
  dtau = 0.f;
  dqr = lds_row<NPC>(tQ, r, h); dkr = lds_row<NPC>(tK, r, h); dvr = lds_row<NPC>(tV, r, h);
  (void)act; (void)sLse; (void)sD; (void)sQa; (void)sPair; (void)tO; (void)sQn; (void)sKin; (void)inv_tau;
#else
  // ================= phase 1: this wave's query tile -> dQ =================
  {
    const Row<NPC> q = lds_row<NPC>(tQ, r, h), dO = lds_row<NPC>(tO, r, h);
    const float qin = sQn[r];
    const float qa = qin * inv_tau, qc = qa * kLog2e;
    const float lse = sLse[r], D = sD[r];
    f32x16 oq = splat(0.f);
    float dt = 0.f;
#pragma unroll
    for (int kj = 0; kj < NW; ++kj) {
      const Row<NPC> kk = lds_row<NPC>(tK, 32 * kj + rho, h), vv = lds_row<NPC>(tV, 32 * kj + rho, h);
      f32x16 aS = mma_rows<NPC>(kk, q, splat(0.f));
      const f32x16 aP = mma_rows<NPC>(vv, dO, splat(-D));                          // dP^T - D: the accumulator starts at -D[query]
      float kr[16];
      row_scalars(sKin + 32 * kj, h, kr);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float t = (32 * kj + c_row(e, h) < n) ? aS[e] * kr[e] : kPadKey;      // t = (q . k) / |k|; padded key: p = 0
        const float p = __builtin_amdgcn_exp2f(fmaf(t, qc, -lse));
        const float ds = p * aP[e];
        dt = fmaf(ds, t, dt);
        aS[e] = ds * kr[e];                                                          // dS / |k|: the A operand is the raw K row
      }
      oq = mma_tokens(tK + 32 * kj * kPitch, aS, oq, lane);                          // dQ^^T[dh][query] (without 1 / tau)
      __builtin_amdgcn_sched_barrier(0);
    }
    // sum_keys dS a = qa sum dS t over this lane's keys; d a / d tau = -a / tau
    dtau = -dt * (qa * inv_tau);
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
#pragma unroll
    for (int t = 0; t < NPC; ++t)
      dqr.p[t] = pack_piece((oq[4 * t] - qh[t][0] * pr) * qa, (oq[4 * t + 1] - qh[t][1] * pr) * qa, (oq[4 * t + 2] - qh[t][2] * pr) * qa,
                            (oq[4 * t + 3] - qh[t][3] * pr) * qa);
  }
  dtau = gd_wave_sum(dtau);
  if (NW == 2 && sub == 1 && lane == 0) *sPair = dtau;        // read after the barrier below
  __builtin_amdgcn_sched_barrier(0);                           // the phases are independent: keep the scheduler from overlapping their registers
  // ================= phase 2: this wave's key tile -> dK, dV =================
  Row<NPC> dkr, dvr;
  {
    const Row<NPC> k = lds_row<NPC>(tK, r, h), v = lds_row<NPC>(tV, r, h);
    const float kin = sKin[r];
    const float kc = kin * kLog2e;
    f32x16 okk = splat(0.f), ov = splat(0.f);
#pragma unroll
    for (int qi = 0; qi < NW; ++qi) {
      const Row<NPC> qq = lds_row<NPC>(tQ, 32 * qi + rho, h), oo = lds_row<NPC>(tO, 32 * qi + rho, h);
      f32x16 aS = mma_rows<NPC>(qq, k, splat(act ? 0.f : -1e30f));     // S[query][key]; padded key (lane): p = 0
      f32x16 aP;                                                       // dP[query][key] - D[query]: the accumulator starts at -D
      {
        float dr[16];
        row_scalars(sD + 32 * qi, h, dr);
#pragma unroll
        for (int e = 0; e < 16; ++e) aP[e] = -dr[e];
      }
      aP = mma_rows<NPC>(oo, v, aP);
      float qr[16], lr_[16];
      row_scalars(sQa + 32 * qi, h, qr);
      row_scalars(sLse + 32 * qi, h, lr_);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f(fmaf(aS[e], qr[e] * kc, -lr_[e]));   // 0 for padded keys and padded queries
        aS[e] = p * aP[e] * qr[e];                                                   // dS / (|q| tau)
        aP[e] = p;
      }
      okk = mma_tokens(tQ + 32 * qi * kPitch, aS, okk, lane);      // dK^^T[dh][key]
      ov = mma_tokens(tO + 32 * qi * kPitch, aP, ov, lane);        // dV^T[dh][key]
      __builtin_amdgcn_sched_barrier(0);
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
#pragma unroll
    for (int t = 0; t < NPC; ++t) {
      dkr.p[t] = pack_piece((okk[4 * t] - kh[t][0] * pr) * kin, (okk[4 * t + 1] - kh[t][1] * pr) * kin, (okk[4 * t + 2] - kh[t][2] * pr) * kin,
                            (okk[4 * t + 3] - kh[t][3] * pr) * kin);
      dvr.p[t] = pack_piece(ov[4 * t], ov[4 * t + 1], ov[4 * t + 2], ov[4 * t + 3]);
    }
  }
#endif
  // ---- results -> tiles (every wavefront of the head is done reading them) -> cooperative stores
  __syncthreads();
  if (lane == 0 && sub == 0) A.dtau_part[(long long)w * A.H + hg * HW + hw] = NW == 2 ? dtau + *sPair : dtau;      // one partial per (window, head)
  store_tile<NPC>(tK, r, h, dqr);
  store_tile<NPC>(tV, r, h, dkr);
  store_tile<NPC>(tQ, r, h, dvr);
  __syncthreads();
#pragma unroll
  for (int p = 0; p < G::P; ++p) {
    const int r2 = p * G::RPI + lr;
    if (r2 < n && r2 < ROWS) {
      unsigned short* gp = A.dqk + (long long)tok[p] * 2 * d + col;
      *reinterpret_cast<uint4*>(gp) = tile_get16(cK, r2, cq);
      *reinterpret_cast<uint4*>(gp + d) = tile_get16(cV, r2, cq);
      *reinterpret_cast<uint4*>(A.dv + (long long)tok[p] * d + col) = tile_get16(cQ, r2, cq);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// kernels: ALL occupancy levels of a layer in ONE launch per direction.  The sparse level's wavefronts (attn16_wave.h: one (window quad,
// head) each, a chain of dependent round trips with next to no arithmetic) ride along with the dense levels' workgroups instead of
// paying a launch of their own; first16 = they take the lowest block indices (their chains start at once, the dense workgroups fill
// in behind - measured better than the other order, tools/attn_layer.py).
// ---------------------------------------------------------------------------------------------------------------------
template <int DH, int K64, int K32>
__global__ __launch_bounds__(256, 4) void k_attn_levels_fwd(CArgs A64, CArgs A32, t16w::A16Args A16, unsigned nb64, unsigned nb32, unsigned nb16,
                                                         int first16) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_coop[];
  unsigned b = blockIdx.x;
  if (first16) {
    if (b < nb16) return t16w::t16_fwd_body<DH>(A16, b, smem_coop);
    b -= nb16;
  }
  if (b < nb64) return coop_fwd<DH, 2, K64>(A64, b, smem_coop);
  b -= nb64;
  if (b < nb32) return coop_fwd<DH, 1, K32>(A32, b, smem_coop);
  t16w::t16_fwd_body<DH>(A16, b - nb32, smem_coop);
}
template <int DH>
__global__ __launch_bounds__(256, 3) void k_attn_levels_bwd(CBwdArgs A64, CBwdArgs A32, t16w::A16BwdArgs A16, unsigned nb64, unsigned nb32,
                                                            unsigned nb16, int first16) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_coop[];
  unsigned b = blockIdx.x;
  if (first16) {
    if (b < nb16) return t16w::t16_bwd_body<DH>(A16, b, nb16, smem_coop);
    b -= nb16;
  }
  if (b < nb64) return coop_bwd<DH, 2>(A64, b, smem_coop);
  b -= nb64;
  if (b < nb32) return coop_bwd<DH, 1>(A32, b, smem_coop);
  t16w::t16_bwd_body<DH>(A16, b - nb32, nb16, smem_coop);
}
// a layer without T = 64 windows (the first stage at high mask ratios): the sparse level dominates and wants more wavefronts per SIMD
// than the T = 64 body's registers allow
template <int DH>
__global__ __launch_bounds__(256, 4) void k_attn_lo_bwd(CBwdArgs A32, t16w::A16BwdArgs A16, unsigned nb32, unsigned nb16, int first16) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_coop[];
  unsigned b = blockIdx.x;
  if (first16) {
    if (b < nb16) return t16w::t16_bwd_body<DH>(A16, b, nb16, smem_coop);
    b -= nb16;
  }
  if (b < nb32) return coop_bwd<DH, 1>(A32, b, smem_coop);
  t16w::t16_bwd_body<DH>(A16, b - nb32, nb16, smem_coop);
}
constexpr int imax(int a, int b) { return a > b ? a : b; }
}  // namespace

static int g_first16 = -1;
static int first16() {
  if (g_first16 < 0) g_first16 = getenv("GDMAE_ATTN_FIRST16") ? atoi(getenv("GDMAE_ATTN_FIRST16")) : 1;
  return g_first16;
}

// bf16 I/O, H % 4 == 0, head dim 16 / 32.  (ws, wl, n) per level: windows [0, n) of the level's (win_start, win_len); any may be empty.
// lse (optional): (n_tok, H) fp32, written for the rows of the T = 32 / 64 levels - what gd_attn_levels_bwd reads.
int gd_attn_levels_fwd(const void* qk, const void* v, void* out, float* lse, const int* csr_tok, const int* ws16, const int* wl16, int n16,
                       const int* ws32, const int* wl32, int n32, const int* ws64, const int* wl64, int n64, int d, int H, const float* tau,
                       float tau_min, hipStream_t st) {
  const CArgs A32{(const unsigned short*)qk, (const unsigned short*)v, (unsigned short*)out, lse, csr_tok, ws32, wl32, n32, d, H, tau, tau_min};
  const CArgs A64{(const unsigned short*)qk, (const unsigned short*)v, (unsigned short*)out, lse, csr_tok, ws64, wl64, n64, d, H, tau, tau_min};
  const t16w::A16Args A16{(const unsigned short*)qk, (const unsigned short*)v, (unsigned short*)out, csr_tok, ws16, wl16, n16, d, H, tau, tau_min};
  const int DH = d / H;
  // head groups per workgroup (ATTN_K64 / ATTN_K32, see coop_fwd) where the head count allows it
  const bool multi = (H / 2) % ATTN_K64 == 0 && (H / 4) % ATTN_K32 == 0;
  const int k64 = multi ? ATTN_K64 : 1, k32 = multi ? ATTN_K32 : 1;
  const unsigned nb32 = (unsigned)((long long)n32 * (H / 4 / k32)), nb64 = (unsigned)((long long)n64 * (H / 2 / k64));
  const unsigned nb16 = (unsigned)((long long)gd_div_up(n16, t16w::kWinPerWave) * (H / 4));
  if (nb16 + nb32 + nb64 == 0) return 0;
  constexpr int lds = imax(imax(2 * fwd_head_lds<64>(), 4 * fwd_head_lds<32>()), 4 * t16w::kWaveLds);
  const dim3 grid(nb64 + nb32 + nb16);
  if (multi) {
    if (DH == 16) hipLaunchKernelGGL((k_attn_levels_fwd<16, ATTN_K64, ATTN_K32>), grid, dim3(256), lds, st, A64, A32, A16, nb64, nb32, nb16, first16());
    else hipLaunchKernelGGL((k_attn_levels_fwd<32, ATTN_K64, ATTN_K32>), grid, dim3(256), lds, st, A64, A32, A16, nb64, nb32, nb16, first16());
  } else {
    if (DH == 16) hipLaunchKernelGGL((k_attn_levels_fwd<16, 1, 1>), grid, dim3(256), lds, st, A64, A32, A16, nb64, nb32, nb16, first16());
    else hipLaunchKernelGGL((k_attn_levels_fwd<32, 1, 1>), grid, dim3(256), lds, st, A64, A32, A16, nb64, nb32, nb16, first16());
  }
  GD_LAUNCH_CHECK();
  return 0;
}
// o, lse: the forward's output rows and lse.  part16 / part32 / part64: partial slots of d loss / d tau (n * H per level)
int gd_attn_levels_bwd(const void* qk, const void* v, const void* o, const float* lse, const void* dout, void* dqk, void* dv, const int* csr_tok,
                       const int* ws16, const int* wl16, int n16, float* part16, const int* ws32, const int* wl32, int n32, float* part32,
                       const int* ws64, const int* wl64, int n64, float* part64, int d, int H, const float* tau, float tau_min, hipStream_t st) {
  const int DH = d / H;
  const CBwdArgs A32{(const unsigned short*)qk, (const unsigned short*)v, (const unsigned short*)o, lse, (const unsigned short*)dout,
                     (unsigned short*)dqk, (unsigned short*)dv, part32, csr_tok, ws32, wl32, n32, d, H, tau, tau_min};
  const CBwdArgs A64{(const unsigned short*)qk, (const unsigned short*)v, (const unsigned short*)o, lse, (const unsigned short*)dout,
                     (unsigned short*)dqk, (unsigned short*)dv, part64, csr_tok, ws64, wl64, n64, d, H, tau, tau_min};
  const t16w::A16BwdArgs A16{(const unsigned short*)qk, (const unsigned short*)v, (const unsigned short*)dout, (unsigned short*)dqk,
                             (unsigned short*)dv, part16, csr_tok, ws16, wl16, n16, d, H, tau, tau_min};
  const unsigned nb32 = (unsigned)((long long)n32 * (H / 4)), nb64 = (unsigned)((long long)n64 * (H / 2));
  const unsigned nb16 = (unsigned)((long long)gd_div_up(n16, t16w::kWinPerWave) * (H / 4));
  if (nb16 + nb32 + nb64 == 0) return 0;
  if (nb64 == 0) {
    constexpr int lds_lo = imax(4 * bwd_head_lds<32>(), 4 * t16w::kWaveLds);
    if (DH == 16) hipLaunchKernelGGL((k_attn_lo_bwd<16>), dim3(nb32 + nb16), dim3(256), lds_lo, st, A32, A16, nb32, nb16, first16());
    else hipLaunchKernelGGL((k_attn_lo_bwd<32>), dim3(nb32 + nb16), dim3(256), lds_lo, st, A32, A16, nb32, nb16, first16());
    GD_LAUNCH_CHECK();
    return 0;
  }
  constexpr int lds = imax(imax(2 * bwd_head_lds<64>(), 4 * bwd_head_lds<32>()), 4 * t16w::kWaveLds);
  if (DH == 16) hipLaunchKernelGGL((k_attn_levels_bwd<16>), dim3(nb64 + nb32 + nb16), dim3(256), lds, st, A64, A32, A16, nb64, nb32, nb16, first16());
  else hipLaunchKernelGGL((k_attn_levels_bwd<32>), dim3(nb64 + nb32 + nb16), dim3(256), lds, st, A64, A32, A16, nb64, nb32, nb16, first16());
  GD_LAUNCH_CHECK();
  return 0;
}
