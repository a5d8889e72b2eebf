// Workgroup-cooperative bf16-MFMA windowed cosine attention for the sparse occupancy level (T = 16 padded tokens; round 5).
//
// Contract and arithmetic: reference cosine_msa.py:114-176 / sst_basic_block.py:22-54 through the window CSR, as attention_t16.hip: four
// consecutive windows of the level are packed greedily into 16-row tiles (rows of different windows never attend to each other), every
// product is one v_mfma_f32_16x16x{32,16}_bf16 per head on the RAW bf16 rows, the cosine normalisation is applied to the accumulator,
// both score orientations in the backward (no atomics).
//
// What is new (attn_tiles.h): a workgroup = one window quad x the HW heads whose row segments make 256 contiguous bytes (4 heads of 32
// channels, all 8 heads of 16 channels); the passes of the quad are planned up front from the scalar window descriptors, the token
// indices of ALL passes are requested together, then the q / k / v (/ dO) segments of all passes - 16 bytes per lane, whole cache lines,
// one load instruction per tensor and pass - and only then does the first pass start: three dependent round trips per workgroup instead
// of one plus two per pass and wavefront, and a quarter of the load instructions.  A wavefront takes HW / 4 heads of a pass from the LDS
// tiles; results leave through the same tiles as 16-byte row segments.
#include "attn_tiles.h"

using namespace attn16;
using attn::and16;
using attn::grp_max;
using attn::grp_sum;
using attn::inv_norm_chunks;
using attn::pack_piece;
using attn::piece_f32;
using attn::ssq16;

namespace {
constexpr int kWinPerGroup = 4;      // consecutive windows of the level handled by one workgroup
constexpr int kPadWin = 31;          // window id of a padding row

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

template <int DH>
struct Geo {
  static constexpr int NP = DH / 16;         // 8-byte pieces per lane and row
  static constexpr int HW = 128 / DH;        // heads per workgroup: 256 bytes of a row
  static constexpr int HPW = HW / 4;         // heads per wavefront
  static constexpr int CPH = DH / 8;         // 16-byte chunks per head
};
constexpr int kFwdHead = 3 * kTile * 2 + 2 * 16 * 4;            // Q, K, V tiles | 1 / (|q| tau), 1 / |k|
constexpr int kBwdHead = 4 * kTile * 2 + 5 * 16 * 4;            // K, V, Q, dO tiles | 1 / |k|, 1 / (|q| tau), 1 / |q|, lse, D

// the passes of a window quad: pass p = windows [pa[p], pb[p]) whose tokens fit one 16-row tile (greedy, in order)
struct Passes {
  int pa[kWinPerGroup], pb[kWinPerGroup];
  int n;
};
__device__ __forceinline__ Passes plan_passes(const int (&len)[kWinPerGroup]) {
  Passes P;
  P.n = 0;
  int a = 0;
#pragma unroll
  for (int p = 0; p < kWinPerGroup; ++p) {
#pragma unroll
    for (int i = 0; i < kWinPerGroup; ++i)
      if (i == a && len[i] == 0) ++a;                         // windows past the end of the level
    int b = a, fill = 0;
#pragma unroll
    for (int i = 0; i < kWinPerGroup; ++i)
      if (i == b && i >= a && len[i] > 0 && fill + len[i] <= 16) {
        fill += len[i];
        b = i + 1;
      }
    P.pa[p] = a;
    P.pb[p] = b;
    if (a < kWinPerGroup) P.n = p + 1;
    a = b;
  }
  return P;
}
// row `row` of pass [a, b): its window (or kPadWin) and CSR position (or -1)
__device__ __forceinline__ void pass_row(const int (&len)[kWinPerGroup], const int (&start)[kWinPerGroup], int a, int b, int row, int& wid, int& idx) {
  int off = 0;
  wid = kPadWin;
  idx = -1;
#pragma unroll
  for (int i = 0; i < kWinPerGroup; ++i) {
    const int li = (i >= a && i < b) ? len[i] : 0;
    if (row >= off && row < off + li) {
      wid = i;
      idx = start[i] + (row - off);
    }
    off += li;
  }
}

template <int DH>
__global__ __launch_bounds__(256) void k_attn_coop16_fwd(A16Args A) {
  using G = Geo<DH>;
  constexpr int NP = G::NP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_c16[];
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int HGN = A.H / G::HW;
  const int wq = blockIdx.x / HGN, hg = blockIdx.x - wq * HGN;
  int len[kWinPerGroup], start[kWinPerGroup];
#pragma unroll
  for (int i = 0; i < kWinPerGroup; ++i) {
    const int w = kWinPerGroup * wq + i;
    const int wc = w < A.n_win ? w : A.n_win - 1;
    const int l_ = A.win_len[wc];
    start[i] = A.win_start[wc];
    len[i] = w < A.n_win ? l_ : 0;
  }
  const Passes P = plan_passes(len);
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);
  const int d = A.d;
  // ---- cooperative pass geometry: row lr, 16-byte chunk ch of the 256-byte segment
  const int lr = tid >> 4, ch = tid & 15, chl = ch / G::CPH, cq = ch % G::CPH;
  const int col = hg * G::HW * DH + ch * 8;
  unsigned short* cQ = reinterpret_cast<unsigned short*>(smem_c16 + chl * kFwdHead);
  unsigned short* cK = cQ + kTile;
  unsigned short* cV = cK + kTile;
  float* cQa = reinterpret_cast<float*>(cV + kTile);
  float* cKin = cQa + 16;
  int* sWid = reinterpret_cast<int*>(smem_c16 + G::HW * kFwdHead);
  int tok[kWinPerGroup], wid[kWinPerGroup];
#pragma unroll
  for (int p = 0; p < kWinPerGroup; ++p) {
    int idx;
    pass_row(len, start, P.pa[p], P.pb[p], lr, wid[p], idx);
    tok[p] = 0;
    if (p < P.n) tok[p] = A.csr_tok[idx >= 0 ? idx : start[0]];
  }
  uint4 q16[kWinPerGroup], k16[kWinPerGroup], v16[kWinPerGroup];
#pragma unroll
  for (int p = 0; p < kWinPerGroup; ++p)
    if (p < P.n) {
      const unsigned short* qp = A.qk + (long long)tok[p] * 2 * d + col;
      q16[p] = *reinterpret_cast<const uint4*>(qp);
      k16[p] = *reinterpret_cast<const uint4*>(qp + d);
      v16[p] = *reinterpret_cast<const uint4*>(A.v + (long long)tok[p] * d + col);
    }
#pragma unroll
  for (int p = 0; p < kWinPerGroup; ++p) {
    if (p >= P.n) break;
    const bool valid = wid[p] != kPadWin;
    const unsigned m = valid ? 0xFFFFFFFFu : 0u;
    const uint4 q = and16(q16[p], m), k = and16(k16[p], m), v = and16(v16[p], m);
    const float qn = inv_norm_chunks<G::CPH>(ssq16(q)), kn = inv_norm_chunks<G::CPH>(ssq16(k));
    if (p > 0) __syncthreads();                   // the previous pass's results have left the tiles
    if (cq == 0) {
      cQa[lr] = qn * inv_tau;
      cKin[lr] = kn;
    }
    if (ch == 0) sWid[lr] = wid[p];
    tile_put16(cQ, lr, cq, q);
    tile_put16(cK, lr, cq, k);
    tile_put16(cV, lr, cq, v);
    __syncthreads();
    // ---- this wavefront's heads of the pass
    const int4 kw4 = *reinterpret_cast<const int4*>(sWid + 4 * g);
    const int kw[4] = {kw4.x, kw4.y, kw4.z, kw4.w};
    const int myw = sWid[c];
#pragma unroll
    for (int j = 0; j < G::HPW; ++j) {
      unsigned char* hb = smem_c16 + (wib * G::HPW + j) * kFwdHead;
      unsigned short* tQ = reinterpret_cast<unsigned short*>(hb);
      unsigned short* tK = tQ + kTile;
      unsigned short* tV = tK + kTile;
      const float* sQa = reinterpret_cast<const float*>(tV + kTile);
      const float* sKin = sQa + 16;
      const Row<NP> qr = lds_row<NP>(tQ, c, g), kr = lds_row<NP>(tK, c, g);
      const float qa = sQa[c];
      const float4 kk = *reinterpret_cast<const float4*>(sKin + 4 * g);
      const float kj[4] = {kk.x, kk.y, kk.z, kk.w};
      f32x4 s = mma_rows<NP>(kr, qr);                              // S^T[key 4 g + j][query c]
      float mx = -1e30f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] = (kw[e] == myw) ? s[e] * qa * kj[e] : -1e30f;        // keys of other windows and padding rows: masked
        mx = fmaxf(mx, s[e]);
      }
      mx = grp_max(mx);
      float l = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        s[e] = __expf(s[e] - mx);
        l += s[e];
      }
      l = grp_sum(l);
      const float il = __builtin_amdgcn_rcpf(l);
      Row<NP> ob;
#pragma unroll
      for (int pp = 0; pp < NP; ++pp) {
        const f32x4 o = mma_tokens(tV, pp, c, g, s);               // O^T[dh 16 p + 4 g + j][query c]
        ob.p[pp] = pack_piece(o[0] * il, o[1] * il, o[2] * il, o[3] * il);
      }
      store_tile<NP>(tQ, c, g, ob);                                // this wavefront is the tile's only reader
    }
    __syncthreads();
    if (valid) *reinterpret_cast<uint4*>(A.out + (long long)tok[p] * d + col) = tile_get16(cQ, lr, cq);
  }
}

template <int DH>
__global__ __launch_bounds__(256) void k_attn_coop16_bwd(A16BwdArgs A) {
  using G = Geo<DH>;
  constexpr int NP = G::NP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_c16[];
  const int tid = threadIdx.x, lane = tid & 63, wib = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int HGN = A.H / G::HW;
  const int wq = blockIdx.x / HGN, hg = blockIdx.x - wq * HGN;
  int len[kWinPerGroup], start[kWinPerGroup];
#pragma unroll
  for (int i = 0; i < kWinPerGroup; ++i) {
    const int w = kWinPerGroup * wq + i;
    const int wc = w < A.n_win ? w : A.n_win - 1;
    const int l_ = A.win_len[wc];
    start[i] = A.win_start[wc];
    len[i] = w < A.n_win ? l_ : 0;
  }
  const Passes P = plan_passes(len);
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);
  const int d = A.d;
  const int lr = tid >> 4, ch = tid & 15, chl = ch / G::CPH, cq = ch % G::CPH;
  const int col = hg * G::HW * DH + ch * 8;
  unsigned short* cK = reinterpret_cast<unsigned short*>(smem_c16 + chl * kBwdHead);
  unsigned short* cV = cK + kTile;
  unsigned short* cQ = cV + kTile;
  unsigned short* cO = cQ + kTile;
  float* cKin = reinterpret_cast<float*>(cO + kTile);
  float* cQa = cKin + 16;
  float* cQn = cKin + 32;
  int* sWid = reinterpret_cast<int*>(smem_c16 + G::HW * kBwdHead);
  int tok[kWinPerGroup], wid[kWinPerGroup];
#pragma unroll
  for (int p = 0; p < kWinPerGroup; ++p) {
    int idx;
    pass_row(len, start, P.pa[p], P.pb[p], lr, wid[p], idx);
    tok[p] = 0;
    if (p < P.n) tok[p] = A.csr_tok[idx >= 0 ? idx : start[0]];
  }
  uint4 q16[kWinPerGroup], k16[kWinPerGroup], v16[kWinPerGroup], o16[kWinPerGroup];
#pragma unroll
  for (int p = 0; p < kWinPerGroup; ++p)
    if (p < P.n) {
      const unsigned short* qp = A.qk + (long long)tok[p] * 2 * d + col;
      q16[p] = *reinterpret_cast<const uint4*>(qp);
      k16[p] = *reinterpret_cast<const uint4*>(qp + d);
      v16[p] = *reinterpret_cast<const uint4*>(A.v + (long long)tok[p] * d + col);
      o16[p] = *reinterpret_cast<const uint4*>(A.dout + (long long)tok[p] * d + col);
    }
  float dtau[G::HPW];
#pragma unroll
  for (int j = 0; j < G::HPW; ++j) dtau[j] = 0.f;
#pragma unroll
  for (int p = 0; p < kWinPerGroup; ++p) {
    if (p >= P.n) break;
    const bool valid = wid[p] != kPadWin;
    const unsigned m = valid ? 0xFFFFFFFFu : 0u;
    const uint4 q = and16(q16[p], m), k = and16(k16[p], m), v = and16(v16[p], m), dO = and16(o16[p], m);
    const float qn = inv_norm_chunks<G::CPH>(ssq16(q)), kn = inv_norm_chunks<G::CPH>(ssq16(k));
    if (p > 0) __syncthreads();
    if (cq == 0) {
      cQn[lr] = qn;
      cQa[lr] = qn * inv_tau;
      cKin[lr] = kn;
    }
    if (ch == 0) sWid[lr] = wid[p];
    tile_put16(cQ, lr, cq, q);
    tile_put16(cK, lr, cq, k);
    tile_put16(cV, lr, cq, v);
    tile_put16(cO, lr, cq, dO);
    __syncthreads();
    const int4 kw4 = *reinterpret_cast<const int4*>(sWid + 4 * g);
    const int kw[4] = {kw4.x, kw4.y, kw4.z, kw4.w};                // window of rows 4 g + j (keys in phase 1, queries in phase 2)
    const int myw = sWid[c];
    const bool act = myw != kPadWin;
#pragma unroll
    for (int j = 0; j < G::HPW; ++j) {
      unsigned char* hb = smem_c16 + (wib * G::HPW + j) * kBwdHead;
      unsigned short* tK = reinterpret_cast<unsigned short*>(hb);
      unsigned short* tV = tK + kTile;
      unsigned short* tQ = tV + kTile;
      unsigned short* tO = tQ + kTile;
      float* sKin = reinterpret_cast<float*>(tO + kTile);
      float* sQa = sKin + 16;
      float* sQn = sKin + 32;
      float* sLse = sKin + 48;
      float* sD = sKin + 64;
      const Row<NP> qr = lds_row<NP>(tQ, c, g), kr = lds_row<NP>(tK, c, g), vr = lds_row<NP>(tV, c, g), dor = lds_row<NP>(tO, c, g);
      const float qin = sQn[c], kin = sKin[c], qa = sQa[c];
      const float4 kk = *reinterpret_cast<const float4*>(sKin + 4 * g);
      const float kj[4] = {kk.x, kk.y, kk.z, kk.w};
      Row<NP> dqo, dko, dvo;
      // ---------------- phase 1: query on the lane -> dQ ----------------
      {
        f32x4 s = mma_rows<NP>(kr, qr);                              // S^T[key 4 g + j][query c] (raw dot products)
        const f32x4 dP = mma_rows<NP>(vr, dor);                      // dP^T[key][query]
        float mx = -1e30f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s[e] = (kw[e] == myw) ? s[e] * qa * kj[e] : -1e30f;
          mx = fmaxf(mx, s[e]);
        }
        mx = grp_max(mx);
        float ex[4], l = 0.f, Dn = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ex[e] = __expf(s[e] - mx);                                 // exactly 0 for masked keys
          l += ex[e];
          Dn = fmaf(ex[e], dP[e], Dn);
        }
        l = grp_sum(l);
        Dn = grp_sum(Dn);
        const float il = __builtin_amdgcn_rcpf(l);
        const float D = Dn * il;
        f32x4 dS;
        float dt = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ds = ex[e] * il * (dP[e] - D);
          dt = fmaf(ds, s[e], dt);                                   // masked key: 0 * -1e30 = -0
          dS[e] = ds * kj[e];                                        // 1 / |k| of the key folded in: the A operand is the raw K row
        }
        if (act) dtau[j] = fmaf(-dt, inv_tau, dtau[j]);              // d a / d tau = -a / tau
        if (g == 0) {
          sLse[c] = act ? mx + __logf(l) : 1e30f;                    // padded queries: exp(a - 1e30) = 0 in phase 2
          sD[c] = D;
        }
        float qh[NP][4], pr = 0.f;
        f32x4 dq[NP];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
          dq[pp] = mma_tokens(tK, pp, c, g, dS);                     // dQ^^T[dh 16 p + 4 g + j][query c], without 1 / tau
          piece_f32(qr.p[pp], qh[pp]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            qh[pp][e] *= qin;
            pr = fmaf(qh[pp][e], dq[pp][e], pr);
          }
        }
        pr = grp_sum(pr);
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
          dqo.p[pp] = pack_piece((dq[pp][0] - qh[pp][0] * pr) * qa, (dq[pp][1] - qh[pp][1] * pr) * qa, (dq[pp][2] - qh[pp][2] * pr) * qa,
                                 (dq[pp][3] - qh[pp][3] * pr) * qa);
      }
      // ---------------- phase 2: key on the lane -> dK, dV ----------------
      __builtin_amdgcn_wave_barrier();
      {
        const float4 qq = *reinterpret_cast<const float4*>(sQa + 4 * g);
        const float4 ll = *reinterpret_cast<const float4*>(sLse + 4 * g);
        const float4 dd = *reinterpret_cast<const float4*>(sD + 4 * g);
        const float qj[4] = {qq.x, qq.y, qq.z, qq.w}, lj[4] = {ll.x, ll.y, ll.z, ll.w}, dj[4] = {dd.x, dd.y, dd.z, dd.w};
        const f32x4 s = mma_rows<NP>(qr, kr);                        // S[query 4 g + j][key c]
        const f32x4 dP = mma_rows<NP>(dor, vr);                      // dP[query][key]
        f32x4 Pm, dS;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pj = (kw[e] == myw && act) ? __expf(s[e] * qj[e] * kin - lj[e]) : 0.f;      // same window only
          Pm[e] = pj;
          dS[e] = pj * (dP[e] - dj[e]) * qj[e];                      // 1 / (|q| tau) of the query folded in
        }
        float kh[NP][4], pr = 0.f;
        f32x4 dk[NP], dvv[NP];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
          dk[pp] = mma_tokens(tQ, pp, c, g, dS);                     // dK^^T[dh][key c]
          dvv[pp] = mma_tokens(tO, pp, c, g, Pm);                    // dV^T[dh][key c]
          piece_f32(kr.p[pp], kh[pp]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            kh[pp][e] *= kin;
            pr = fmaf(kh[pp][e], dk[pp][e], pr);
          }
        }
        pr = grp_sum(pr);
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
          dko.p[pp] = pack_piece((dk[pp][0] - kh[pp][0] * pr) * kin, (dk[pp][1] - kh[pp][1] * pr) * kin, (dk[pp][2] - kh[pp][2] * pr) * kin,
                                 (dk[pp][3] - kh[pp][3] * pr) * kin);
          dvo.p[pp] = pack_piece(dvv[pp][0], dvv[pp][1], dvv[pp][2], dvv[pp][3]);
        }
      }
      // results -> the head's tiles (this wavefront is their only reader)
      __builtin_amdgcn_wave_barrier();
      store_tile<NP>(tK, c, g, dqo);
      store_tile<NP>(tV, c, g, dko);
      store_tile<NP>(tQ, c, g, dvo);
    }
    __syncthreads();
    if (valid) {
      unsigned short* gp = A.dqk + (long long)tok[p] * 2 * d + col;
      *reinterpret_cast<uint4*>(gp) = tile_get16(cK, lr, cq);
      *reinterpret_cast<uint4*>(gp + d) = tile_get16(cV, lr, cq);
      *reinterpret_cast<uint4*>(A.dv + (long long)tok[p] * d + col) = tile_get16(cQ, lr, cq);
    }
  }
  // the 4 lane groups of a column hold the same reduced statistics and split the keys between them (dt sums this lane's 4 keys), so the
  // plain wave sum counts every (query, key) pair once; one partial per (window quad, head).  The level owns n_win * H partial slots:
  // this grid fills (quads * H) of them and zeroes the rest, so the consumer can sum the whole range without a separate clear
#pragma unroll
  for (int j = 0; j < G::HPW; ++j) {
    const float t = gd_wave_sum(dtau[j]);
    if (lane == 0) {
      const long long quads = (A.n_win + kWinPerGroup - 1) / kWinPerGroup;
      const long long mine = (long long)wq * A.H + hg * G::HW + wib * G::HPW + j, used = quads * A.H, all = (long long)A.n_win * A.H;
      A.dtau_part[mine] = t;
      for (long long i = used + mine; i < all; i += used) A.dtau_part[i] = 0.f;
    }
  }
}
}  // namespace

// bf16 I/O, T = 16, H % 4 == 0, head dim 16 (then H % 8 == 0) or 32; called from attention.hip's entry points
bool gd_attn_coop16_ok(int d, int H) {
  const int DH = H > 0 ? d / H : 0;
  return (DH == 32 && H % 4 == 0) || (DH == 16 && H % 8 == 0);
}
int gd_attn_coop16_fwd(const void* qk, const void* v, void* out, const int* csr_tok, const int* win_start, const int* win_len, int n_win, int d,
                       int H, const float* tau, float tau_min, hipStream_t st) {
  A16Args A{(const unsigned short*)qk, (const unsigned short*)v, (unsigned short*)out, csr_tok, win_start, win_len, n_win, d, H, tau, tau_min};
  const int quads = gd_div_up(n_win, kWinPerGroup);
  if (d / H == 16) hipLaunchKernelGGL((k_attn_coop16_fwd<16>), dim3((unsigned)(quads * (H / 8))), dim3(256), 8 * kFwdHead + 64, st, A);
  else hipLaunchKernelGGL((k_attn_coop16_fwd<32>), dim3((unsigned)(quads * (H / 4))), dim3(256), 4 * kFwdHead + 64, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}
int gd_attn_coop16_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, float* dtau_part, const int* csr_tok,
                       const int* win_start, const int* win_len, int n_win, int d, int H, const float* tau, float tau_min, hipStream_t st) {
  A16BwdArgs A{(const unsigned short*)qk, (const unsigned short*)v, (const unsigned short*)dout, (unsigned short*)dqk, (unsigned short*)dv,
               dtau_part, csr_tok, win_start, win_len, n_win, d, H, tau, tau_min};
  const int quads = gd_div_up(n_win, kWinPerGroup);
  if (d / H == 16) hipLaunchKernelGGL((k_attn_coop16_bwd<16>), dim3((unsigned)(quads * (H / 8))), dim3(256), 8 * kBwdHead + 64, st, A);
  else hipLaunchKernelGGL((k_attn_coop16_bwd<32>), dim3((unsigned)(quads * (H / 4))), dim3(256), 4 * kBwdHead + 64, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}
