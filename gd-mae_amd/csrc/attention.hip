// Windowed cosine multi-head attention (forward + backward) over variable-length token groups, reading
// the window CSR directly - the padded (nW, T, d) tensors, key-padding masks and the flat2window /
// window2flat copies of the reference are never materialised.
//
// Replaces:
//   flat2window_v2 / window2flat_v2        reference pcdet/models/model_utils/sst_utils.py:107-180
//   WindowAttention.forward                pcdet/models/model_utils/sst_basic_block.py:22-54
//   _scaled_cosine_attention               pcdet/models/model_utils/cosine_msa.py:114-176
//     q^ = q/max(|q|,1e-12), k^ likewise (F.normalize), A = q^ k^T / clamp(tau, tau_min),
//     -inf on padded keys, softmax, P V.  (in/out projections stay token-wise GEMMs outside.)
//
// MI355X mapping.  A window holds n <= 64 tokens (8x8 BEV cells) and a head is 16 or 32 wide, so one
// (window, head) problem is at most 64x64x32 - far too small to amortise MFMA fragment shuffles, and
// the kernel is bound by fetching each token's q/k/v rows once from HBM/L2 (< 1 GFLOP per
// frame).  The layout is therefore "one lane = one query row": a 64-lane wavefront covers the T=64
// occupancy level exactly, and packs 2 (T=32) or 4 (T=16) heads of a window for the sparser levels so
// no lane idles on padding rows that the reference computes and discards.  K^ and V of the wave's
// heads are staged once in LDS (rows padded to DH+4 floats: conflict-free 16-byte stores, broadcast
// reads), scores/softmax/PV are lane-local register loops - no cross-lane reduction anywhere.  The
// backward runs two lane-local phases (lane = query for dQ, then lane = key for dK/dV) re-using the
// same two LDS tiles, so there are no atomics and gradients are deterministic; d(tau) is reduced
// per wave and summed in a fixed order by a second tiny kernel.
#include "common.h"

#define ATT_EPS 1e-12f

// ---- global-memory row IO: fp32 or bf16 (token GEMM outputs under bf16 autocast), fp32 in registers ----
__device__ inline float bf2f(unsigned v) { return __uint_as_float(v << 16); }
__device__ inline unsigned f2bf(float f) { return gd_to_bf16(f); }   // round-to-nearest-even
struct IoF32 {
  typedef float T;
  template <int DH>
  static __device__ inline void load(const T* __restrict__ p, float (&r)[DH]) {
#pragma unroll
    for (int c = 0; c < DH; c += 4) {
      float4 v = *reinterpret_cast<const float4*>(p + c);
      r[c] = v.x; r[c + 1] = v.y; r[c + 2] = v.z; r[c + 3] = v.w;
    }
  }
  template <int DH>
  static __device__ inline void store(T* __restrict__ p, const float (&r)[DH]) {
#pragma unroll
    for (int c = 0; c < DH; c += 4) *reinterpret_cast<float4*>(p + c) = make_float4(r[c], r[c + 1], r[c + 2], r[c + 3]);
  }
};
struct IoBF16 {
  typedef unsigned short T;
  template <int DH>
  static __device__ inline void load(const T* __restrict__ p, float (&r)[DH]) {
#pragma unroll
    for (int c = 0; c < DH; c += 8) {
      uint4 v = *reinterpret_cast<const uint4*>(p + c);
      const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        r[c + 2 * k] = bf2f(w[k] & 0xFFFFu);
        r[c + 2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u);
      }
    }
  }
  template <int DH>
  static __device__ inline void store(T* __restrict__ p, const float (&r)[DH]) {
#pragma unroll
    for (int c = 0; c < DH; c += 8) {
      uint4 v;
      v.x = f2bf(r[c]) | (f2bf(r[c + 1]) << 16);
      v.y = f2bf(r[c + 2]) | (f2bf(r[c + 3]) << 16);
      v.z = f2bf(r[c + 4]) | (f2bf(r[c + 5]) << 16);
      v.w = f2bf(r[c + 6]) | (f2bf(r[c + 7]) << 16);
      *reinterpret_cast<uint4*>(p + c) = v;
    }
  }
};

template <int DH>
__device__ inline void load_row(const float* __restrict__ p, float (&r)[DH]) {
#pragma unroll
  for (int c = 0; c < DH; c += 4) {
    float4 v = *reinterpret_cast<const float4*>(p + c);
    r[c] = v.x;
    r[c + 1] = v.y;
    r[c + 2] = v.z;
    r[c + 3] = v.w;
  }
}
template <int DH>
__device__ inline void store_row(float* __restrict__ p, const float (&r)[DH]) {
#pragma unroll
  for (int c = 0; c < DH; c += 4) *reinterpret_cast<float4*>(p + c) = make_float4(r[c], r[c + 1], r[c + 2], r[c + 3]);
}
template <int DH>
__device__ inline float dot_lds(const float (&a)[DH], const float* __restrict__ l) {
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < DH; c += 4) {
    float4 v = *reinterpret_cast<const float4*>(l + c);
    s = fmaf(a[c], v.x, s);
    s = fmaf(a[c + 1], v.y, s);
    s = fmaf(a[c + 2], v.z, s);
    s = fmaf(a[c + 3], v.w, s);
  }
  return s;
}
template <int DH>
__device__ inline float normalize(float (&r)[DH], float& inv_norm) {
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < DH; ++c) ss = fmaf(r[c], r[c], ss);
  float n = sqrtf(ss);
  inv_norm = 1.f / fmaxf(n, ATT_EPS);
#pragma unroll
  for (int c = 0; c < DH; ++c) r[c] *= inv_norm;
  return n;
}

struct AttnArgs {
  const void* qk;    // (Ms, 2d): q in [0,d), k in [d,2d)   fp32 or bf16 (IO)
  const void* v;     // (Ms, d)
  void* out;         // (Ms, d)
  const int* csr_tok;
  const int* win_start;
  const int* win_len;
  int n_win;
  int d;             // model dim = H * DH
  int H;
  const float* tau;
  float tau_min;
};

template <int T, int DH, typename IO>
__global__ __launch_bounds__(256) void k_win_attn_fwd(AttnArgs A) {
  typedef typename IO::T io_t;
  const io_t* gqk = (const io_t*)A.qk;
  const io_t* gv = (const io_t*)A.v;
  io_t* gout = (io_t*)A.out;
  constexpr int G = GD_WAVE / T;          // heads per wavefront
  constexpr int LD = DH + 4;              // padded LDS row
  constexpr int GS = T * LD + 16;         // head-tile stride: +16 floats so packed heads start on different banks
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  constexpr int TILE = G * GS;
  float* sK = smem + wib * (2 * TILE);
  float* sV = sK + TILE;
  const int groups = A.H / G;
  const long long item = (long long)blockIdx.x * 4 + wib;
  if (item >= (long long)A.n_win * groups) return;
  const int w = (int)(item / groups);
  const int hg = (int)(item % groups);
  const int sub = lane / T, r = lane % T;
  const int h = hg * G + sub;
  const int n = A.win_len[w];
  const int start = A.win_start[w];
  const bool act = r < n;
  const int t = act ? A.csr_tok[start + r] : 0;
  const float inv_tau = 1.f / fmaxf(*A.tau, A.tau_min);

  float q[DH], tmp[DH];
  float dummy;
  if (act) {
    IO::template load<DH>(gqk + (long long)t * 2 * A.d + h * DH, q);
    normalize<DH>(q, dummy);
    IO::template load<DH>(gqk + (long long)t * 2 * A.d + A.d + h * DH, tmp);
    normalize<DH>(tmp, dummy);
    store_row<DH>(sK + sub * GS + r * LD, tmp);
    IO::template load<DH>(gv + (long long)t * A.d + h * DH, tmp);
    store_row<DH>(sV + sub * GS + r * LD, tmp);
  }
  __builtin_amdgcn_wave_barrier();
  if (!act) return;
  const float* kb = sK + sub * GS;
  const float* vb = sV + sub * GS;
  float s[T];
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    float a = -INFINITY;
    if (j < n) a = dot_lds<DH>(q, kb + j * LD) * inv_tau;
    s[j] = a;
    m = fmaxf(m, a);
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    float p = (j < n) ? expf(s[j] - m) : 0.f;
    s[j] = p;
    l += p;
  }
  float o[DH];
#pragma unroll
  for (int c = 0; c < DH; ++c) o[c] = 0.f;
#pragma unroll
  for (int j = 0; j < T; ++j) {
    if (j < n) {
      const float p = s[j];
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        float4 vv = *reinterpret_cast<const float4*>(vb + j * LD + c);
        o[c] = fmaf(p, vv.x, o[c]);
        o[c + 1] = fmaf(p, vv.y, o[c + 1]);
        o[c + 2] = fmaf(p, vv.z, o[c + 2]);
        o[c + 3] = fmaf(p, vv.w, o[c + 3]);
      }
    }
  }
  const float il = 1.f / l;
#pragma unroll
  for (int c = 0; c < DH; ++c) o[c] *= il;
  IO::template store<DH>(gout + (long long)t * A.d + h * DH, o);
}

struct AttnBwdArgs {
  const void* qk;     // fp32 or bf16 (IO), like the forward
  const void* v;
  const void* dout;   // (Ms, d)
  void* dqk;          // (Ms, 2d)
  void* dv;           // (Ms, d)
  float* dtau_part;   // one partial per wavefront item
  const int* csr_tok;
  const int* win_start;
  const int* win_len;
  int n_win;
  int d;
  int H;
  const float* tau;
  float tau_min;
};

template <int T, int DH, typename IO>
__global__ __launch_bounds__(256) void k_win_attn_bwd(AttnBwdArgs A) {
  typedef typename IO::T io_t;
  const io_t* gqk = (const io_t*)A.qk;
  const io_t* gv = (const io_t*)A.v;
  const io_t* gdo = (const io_t*)A.dout;
  io_t* gdqk = (io_t*)A.dqk;
  io_t* gdv = (io_t*)A.dv;
  constexpr int G = GD_WAVE / T;
  constexpr int LD = DH + 4;
  constexpr int GS = T * LD + 16;
  constexpr int TILE = G * GS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  float* s0 = smem + wib * (2 * TILE + 2 * GD_WAVE);
  float* s1 = s0 + TILE;
  float* sLse = s1 + TILE;
  float* sD = sLse + GD_WAVE;
  const int groups = A.H / G;
  const long long item = (long long)blockIdx.x * 4 + wib;
  if (item >= (long long)A.n_win * groups) return;
  const int w = (int)(item / groups);
  const int hg = (int)(item % groups);
  const int sub = lane / T, r = lane % T;
  const int h = hg * G + sub;
  const int n = A.win_len[w];
  const int start = A.win_start[w];
  const bool act = r < n;
  const int t = act ? A.csr_tok[start + r] : 0;
  const float tau_c = fmaxf(*A.tau, A.tau_min);
  const float inv_tau = 1.f / tau_c;

  float q[DH], dO[DH];
  float qin = 0.f, kin = 0.f;  // 1 / max(|q|, eps)
  if (act) {
    float tmp[DH];
    IO::template load<DH>(gqk + (long long)t * 2 * A.d + A.d + h * DH, tmp);
    normalize<DH>(tmp, kin);
    store_row<DH>(s0 + sub * GS + r * LD, tmp);  // K^
    IO::template load<DH>(gv + (long long)t * A.d + h * DH, tmp);
    store_row<DH>(s1 + sub * GS + r * LD, tmp);  // V
    IO::template load<DH>(gqk + (long long)t * 2 * A.d + h * DH, q);
    normalize<DH>(q, qin);
    IO::template load<DH>(gdo + (long long)t * A.d + h * DH, dO);
  }
  __builtin_amdgcn_wave_barrier();
  float dtau = 0.f;
  const float* b0 = s0 + sub * GS;
  const float* b1 = s1 + sub * GS;
  // ---- phase A: lane = query row i
  if (act) {
    float s[T];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      float a = -INFINITY;
      if (j < n) a = dot_lds<DH>(q, b0 + j * LD) * inv_tau;
      s[j] = a;
      m = fmaxf(m, a);
    }
    // s[j] <- e_j = exp(a_ij - m); l = sum e_j; D = sum p_j (dO_i . V_j)
    float l = 0.f, D = 0.f;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      if (j < n) {
        const float e = expf(s[j] - m);
        l += e;
        D = fmaf(e, dot_lds<DH>(dO, b1 + j * LD), D);
        s[j] = e;
      }
    }
    const float il = 1.f / l;
    D *= il;
    float dq[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) dq[c] = 0.f;
#pragma unroll
    for (int j = 0; j < T; ++j) {
      if (j < n) {
        const float p = s[j] * il;
        const float dS = p * (dot_lds<DH>(dO, b1 + j * LD) - D);
        const float a = (s[j] > 0.f) ? logf(s[j]) + m : 0.f;  // a_ij recovered from e_j (p = 0 contributes 0)
        dtau = fmaf(-dS, a * inv_tau, dtau);       // d a_ij / d tau_c = -a_ij / tau_c
        const float g = dS * inv_tau;
#pragma unroll
        for (int c = 0; c < DH; c += 4) {
          float4 kk = *reinterpret_cast<const float4*>(b0 + j * LD + c);
          dq[c] = fmaf(g, kk.x, dq[c]);
          dq[c + 1] = fmaf(g, kk.y, dq[c + 1]);
          dq[c + 2] = fmaf(g, kk.z, dq[c + 2]);
          dq[c + 3] = fmaf(g, kk.w, dq[c + 3]);
        }
      }
    }
    // through q^ = q / max(|q|, eps):  dq = (dq^ - q^ (q^ . dq^)) / max(|q|, eps)
    float pr = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) pr = fmaf(q[c], dq[c], pr);
#pragma unroll
    for (int c = 0; c < DH; ++c) dq[c] = (dq[c] - q[c] * pr) * qin;
    IO::template store<DH>(gdqk + (long long)t * 2 * A.d + h * DH, dq);
    sLse[lane] = m + logf(l);
    sD[lane] = D;
  }
  __builtin_amdgcn_wave_barrier();
  // ---- phase B: lane = key row j; tiles now hold Q^ and dO
  float k[DH], vv[DH];
  if (act) {
    load_row<DH>(s0 + sub * GS + r * LD, k);
    load_row<DH>(s1 + sub * GS + r * LD, vv);
  }
  __builtin_amdgcn_wave_barrier();
  if (act) {
    store_row<DH>(s0 + sub * GS + r * LD, q);
    store_row<DH>(s1 + sub * GS + r * LD, dO);
  }
  __builtin_amdgcn_wave_barrier();
  if (act) {
    float dk[DH], dvv[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) {
      dk[c] = 0.f;
      dvv[c] = 0.f;
    }
    const float* lse = sLse + sub * T;
    const float* Dr = sD + sub * T;
    for (int i = 0; i < n; ++i) {
      const float a = dot_lds<DH>(k, b0 + i * LD) * inv_tau;
      const float p = expf(a - lse[i]);
      const float g = dot_lds<DH>(vv, b1 + i * LD);
      const float dS = p * (g - Dr[i]) * inv_tau;
#pragma unroll
      for (int c = 0; c < DH; c += 4) {
        float4 qq = *reinterpret_cast<const float4*>(b0 + i * LD + c);
        float4 dd = *reinterpret_cast<const float4*>(b1 + i * LD + c);
        dk[c] = fmaf(dS, qq.x, dk[c]);
        dk[c + 1] = fmaf(dS, qq.y, dk[c + 1]);
        dk[c + 2] = fmaf(dS, qq.z, dk[c + 2]);
        dk[c + 3] = fmaf(dS, qq.w, dk[c + 3]);
        dvv[c] = fmaf(p, dd.x, dvv[c]);
        dvv[c + 1] = fmaf(p, dd.y, dvv[c + 1]);
        dvv[c + 2] = fmaf(p, dd.z, dvv[c + 2]);
        dvv[c + 3] = fmaf(p, dd.w, dvv[c + 3]);
      }
    }
    float pr = 0.f;
#pragma unroll
    for (int c = 0; c < DH; ++c) pr = fmaf(k[c], dk[c], pr);
#pragma unroll
    for (int c = 0; c < DH; ++c) dk[c] = (dk[c] - k[c] * pr) * kin;
    IO::template store<DH>(gdqk + (long long)t * 2 * A.d + A.d + h * DH, dk);
    IO::template store<DH>(gdv + (long long)t * A.d + h * DH, dvv);
  }
  dtau = gd_wave_sum(dtau);
  if (lane == 0) {
    // the level owns n_win * H partial slots; the n_win * H / G items fill the first ones and zero the rest (no separate clear)
    const long long used = (long long)A.n_win * groups, all = (long long)A.n_win * A.H;
    A.dtau_part[item] = dtau;
    for (long long i = used + item; i < all; i += used) A.dtau_part[i] = 0.f;
  }
}

// sum of `n` values over one 1024-thread workgroup, fixed association order (every thread returns the total)
__device__ inline float gd_block1024_sum(const float* __restrict__ part, long long n, float* sh /* [16] */) {
  // four independent accumulators per lane, 8 loads in flight (a single dependent chain made this latency-bound:
  // 33 us for 127 k partials)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  long long i = threadIdx.x;
#pragma unroll 2
  for (; i + 3 * 1024 < n; i += 4 * 1024) {
    a0 += part[i];
    a1 += part[i + 1024];
    a2 += part[i + 2048];
    a3 += part[i + 3072];
  }
  for (; i < n; i += 1024) a0 += part[i];
  const float acc = gd_wave_sum((a0 + a1) + (a2 + a3));
  __syncthreads();                       // sh may still be read from a previous call
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x / 64] = acc;
  __syncthreads();
  float s = 0.f;
  for (int k = 0; k < 1024 / GD_WAVE; ++k) s += sh[k];
  return s;
}

// deterministic sum of `n` partials into out[0] (single workgroup, fixed association order)
// gate (optional): out is zeroed unless gate[0] >= gate_min (gradient of clamp(tau, min) w.r.t. tau)
__global__ __launch_bounds__(1024) void k_sum_partials(const float* __restrict__ part, long long n, float scale,
                                                       float* __restrict__ out, int accumulate,
                                                       const float* __restrict__ gate, float gate_min) {
  __shared__ float sh[1024 / GD_WAVE];
  float s = gd_block1024_sum(part, n, sh);
  if (threadIdx.x == 0) {
    s *= scale;
    if (gate && !(gate[0] >= gate_min)) s = 0.f;
    out[0] = accumulate ? out[0] + s : s;
  }
}

// weighted-mean finish of the Chamfer loss (pytorch3d point_reduction / batch_reduction "mean" with weights):
// out = {sum(term) * inv, inv},  inv = 1 / sum(weights) (0 when no weight is positive)
__global__ __launch_bounds__(1024) void k_weighted_mean_finish(const float* __restrict__ term, const float* __restrict__ weights,
                                                               long long n, float* __restrict__ out) {
  __shared__ float sh[1024 / GD_WAVE];
  __shared__ float sh2[1024 / GD_WAVE];
  // both arrays in one sweep, four 16-byte loads of each in flight per thread (a single workgroup walks 2 x 127 k floats: the launch is
  // a chain of round trips - 30 us as two sweeps with four scalar loads in flight); fixed association order
  float t4[4] = {0.f, 0.f, 0.f, 0.f}, w4[4] = {0.f, 0.f, 0.f, 0.f};
  const long long n4 = ((reinterpret_cast<unsigned long long>(term) | reinterpret_cast<unsigned long long>(weights)) & 15ull) == 0 ? n / 4 : 0;
  const float4* tv = reinterpret_cast<const float4*>(term);
  const float4* wv = reinterpret_cast<const float4*>(weights);
  for (long long i0 = threadIdx.x; i0 < n4; i0 += 4 * 1024) {
    float4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * 1024 < n4 ? i0 + u * 1024 : n4 - 1;       // unconditional (clamped), masked below
      a[u] = tv[i];
      b[u] = wv[i];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float m = i0 + u * 1024 < n4 ? 1.f : 0.f;
      t4[u] = fmaf(m, (a[u].x + a[u].y) + (a[u].z + a[u].w), t4[u]);
      w4[u] = fmaf(m, (b[u].x + b[u].y) + (b[u].z + b[u].w), w4[u]);
    }
  }
  for (long long i = 4 * n4 + threadIdx.x; i < n; i += 1024) {
    t4[0] += term[i];
    w4[0] += weights[i];
  }
  const float ta = gd_wave_sum((t4[0] + t4[1]) + (t4[2] + t4[3])), wa = gd_wave_sum((w4[0] + w4[1]) + (w4[2] + w4[3]));
  if ((threadIdx.x & 63) == 0) {
    sh[threadIdx.x / 64] = ta;
    sh2[threadIdx.x / 64] = wa;
  }
  __syncthreads();
  float t = 0.f, w = 0.f;
  for (int k = 0; k < 1024 / GD_WAVE; ++k) {
    t += sh[k];
    w += sh2[k];
  }
  if (threadIdx.x == 0) {
    const float inv = w > 0.f ? 1.0f / fmaxf(w, 1e-30f) : 0.f;
    out[0] = t * inv;
    out[1] = inv;
  }
}

extern "C" int gdmae_weighted_mean_finish(const float* term, const float* weights, long long n, float* out, void* stream) {
  hipLaunchKernelGGL(k_weighted_mean_finish, dim3(1), dim3(1024), 0, (hipStream_t)stream, term, weights, n, out);
  GD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gdmae_sum_partials(const float* part, long long n, float scale, float* out, int accumulate, void* stream) {
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, n, scale, out, accumulate,
                     (const float*)nullptr, 0.f);
  GD_LAUNCH_CHECK();
  return 0;
}

// out[0] = (gate[0] >= gate_min) ? scale * sum(part) : 0    (d loss / d tau through clamp(tau, tau_min))
extern "C" int gdmae_sum_partials_gated(const float* part, long long n, float scale, float* out, const float* gate,
                                        float gate_min, void* stream) {
  hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, n, scale, out, 0, gate, gate_min);
  GD_LAUNCH_CHECK();
  return 0;
}

template <int T, int DH, typename IO>
static int launch_fwd(const AttnArgs& A, hipStream_t st) {
  constexpr int G = GD_WAVE / T;
  const long long items = (long long)A.n_win * (A.H / G);
  const size_t lds = 4 * (2 * G * (T * (DH + 4) + 16)) * sizeof(float);
  hipLaunchKernelGGL((k_win_attn_fwd<T, DH, IO>), dim3(gd_div_up(items, 4)), dim3(256), lds, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}
template <int T, int DH, typename IO>
static int launch_bwd(const AttnBwdArgs& A, hipStream_t st) {
  constexpr int G = GD_WAVE / T;
  const long long items = (long long)A.n_win * (A.H / G);
  const size_t lds = 4 * (2 * G * (T * (DH + 4) + 16) + 2 * GD_WAVE) * sizeof(float);
  hipLaunchKernelGGL((k_win_attn_bwd<T, DH, IO>), dim3(gd_div_up(items, 4)), dim3(256), lds, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}

template <typename IO>
static int dispatch_fwd(const AttnArgs& A, int T, int DH, hipStream_t st) {
  if (DH == 16) {
    if (T == 16) return launch_fwd<16, 16, IO>(A, st);
    if (T == 32) return launch_fwd<32, 16, IO>(A, st);
    return launch_fwd<64, 16, IO>(A, st);
  }
  if (T == 16) return launch_fwd<16, 32, IO>(A, st);
  if (T == 32) return launch_fwd<32, 32, IO>(A, st);
  return launch_fwd<64, 32, IO>(A, st);
}
template <typename IO>
static int dispatch_bwd(const AttnBwdArgs& A, int T, int DH, hipStream_t st) {
  if (DH == 16) {
    if (T == 16) return launch_bwd<16, 16, IO>(A, st);
    if (T == 32) return launch_bwd<32, 16, IO>(A, st);
    return launch_bwd<64, 16, IO>(A, st);
  }
  if (T == 16) return launch_bwd<16, 32, IO>(A, st);
  if (T == 32) return launch_bwd<32, 32, IO>(A, st);
  return launch_bwd<64, 32, IO>(A, st);
}

// MFMA variant for the dense levels (attention_mfma.hip)
int gd_attn_mfma_fwd(const void* qk, const void* v, void* out, int io_bf16, const int* csr_tok, const int* win_start,
                     const int* win_len, int n_win, int T, int d, int H, const float* tau, float tau_min, hipStream_t st);
int gd_attn_mfma_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, int io_bf16, float* dtau_part,
                     const int* csr_tok, const int* win_start, const int* win_len, int n_win, int T, int d, int H,
                     const float* tau, float tau_min, hipStream_t st);

// bf16-MFMA variant of the T = 32 / 64 levels for bf16 token I/O (attention_t32.hip)
int gd_attn_t32_fwd(const void* qk, const void* v, void* out, const int* csr_tok, const int* win_start, const int* win_len, int n_win,
                    int T, int d, int H, const float* tau, float tau_min, hipStream_t st);
int gd_attn_t32_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, float* dtau_part, const int* csr_tok,
                    const int* win_start, const int* win_len, int n_win, int T, int d, int H, const float* tau, float tau_min,
                    hipStream_t st);

int gd_attn_t3264_fwd(const void* qk, const void* v, void* out, const int* csr_tok, const int* ws32, const int* wl32, int n32, const int* ws64,
                      const int* wl64, int n64, int d, int H, const float* tau, float tau_min, hipStream_t st);
int gd_attn_t3264_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, const int* csr_tok, const int* ws32,
                      const int* wl32, int n32, float* part32, const int* ws64, const int* wl64, int n64, float* part64, int d, int H,
                      const float* tau, float tau_min, hipStream_t st);

// workgroup-cooperative kernels, all levels of a layer in one launch per direction (attention_coop.hip, round 5): whole row segments
// through LDS, 16 bytes per lane; the forward leaves the rows' log-sum-exp for the backward, which also reads the forward's output
int gd_attn_levels_fwd(const void* qk, const void* v, void* out, float* lse, const int* csr_tok, const int* ws16, const int* wl16, int n16,
                       const int* ws32, const int* wl32, int n32, const int* ws64, const int* wl64, int n64, int d, int H, const float* tau,
                       float tau_min, hipStream_t st);
int gd_attn_levels_bwd(const void* qk, const void* v, const void* o, const float* lse, const void* dout, void* dqk, void* dv, const int* csr_tok,
                       const int* ws16, const int* wl16, int n16, float* part16, const int* ws32, const int* wl32, int n32, float* part32,
                       const int* ws64, const int* wl64, int n64, float* part64, int d, int H, const float* tau, float tau_min, hipStream_t st);

// bf16-MFMA variant of the T = 16 level for bf16 token I/O (attention_t16.hip)
int gd_attn_t16_fwd(const void* qk, const void* v, void* out, const int* csr_tok, const int* win_start, const int* win_len, int n_win,
                    int d, int H, const float* tau, float tau_min, hipStream_t st);
int gd_attn_t16_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, float* dtau_part, const int* csr_tok,
                    const int* win_start, const int* win_len, int n_win, int d, int H, const float* tau, float tau_min, hipStream_t st);

// 0: bf16 I/O on the bf16 matrix-core kernels at every level (attention_t16.hip, attention_t32.hip), fp32 I/O
//    on the exact-fp32 MFMA kernels for T >= 32 and the lane-per-query VALU kernels for T = 16;
// 1: VALU kernels only;  2: like 0 but always the exact-fp32 MFMA kernels for T >= 32 and the VALU kernels for T = 16;
// 3: like 0 with the round-4 one-wavefront-per-(window, head) kernels for the bf16 T = 32 / 64 levels (attention_t32.hip) instead of the
//    workgroup-cooperative ones (attention_coop.hip) - A/B reference
static int g_attn_impl = 0;
static inline bool attn_auto() { return g_attn_impl == 0 || g_attn_impl == 3; }
extern "C" int gdmae_set_attention_impl(int impl) {
  GD_REQUIRE(impl >= 0 && impl <= 3, "attention impl: 0 (auto), 1 (VALU only), 2 (fp32 MFMA) or 3 (auto, per-(window, head) bf16 kernels)");
  g_attn_impl = impl;
  return 0;
}

// One occupancy level: windows [0, n_win) of (win_start, win_len); T = padded tokens of the level.
// io_bf16 = 0: qk / v / out are fp32; 1: bf16 (arithmetic is fp32 in registers either way).
extern "C" int gdmae_window_attention_fwd(const void* qk, const void* v, void* out, int io_bf16, const int* csr_tok,
                                          const int* win_start, const int* win_len, int n_win, int T, int d, int H,
                                          const float* tau, float tau_min, void* stream) {
  if (n_win <= 0) return 0;
  GD_REQUIRE(d % H == 0, "d % H");
  const int DH = d / H;
  GD_REQUIRE(DH == 16 || DH == 32, "head dim must be 16 or 32");
  GD_REQUIRE(T == 16 || T == 32 || T == 64, "T must be 16/32/64");
  GD_REQUIRE(H % (GD_WAVE / T) == 0, "heads must pack evenly into a wavefront");
  hipStream_t st = (hipStream_t)stream;
  if (attn_auto() && T == 16 && io_bf16 && H % 4 == 0)
    return gd_attn_t16_fwd(qk, v, out, csr_tok, win_start, win_len, n_win, d, H, tau, tau_min, st);
  if (g_attn_impl == 0 && T >= 32 && io_bf16 && H % 4 == 0)
    return T == 32 ? gd_attn_levels_fwd(qk, v, out, nullptr, csr_tok, nullptr, nullptr, 0, win_start, win_len, n_win, nullptr, nullptr, 0, d, H, tau,
                                        tau_min, st)
                   : gd_attn_levels_fwd(qk, v, out, nullptr, csr_tok, nullptr, nullptr, 0, nullptr, nullptr, 0, win_start, win_len, n_win, d, H, tau,
                                        tau_min, st);
  if (attn_auto() && T >= 32 && io_bf16)
    return gd_attn_t32_fwd(qk, v, out, csr_tok, win_start, win_len, n_win, T, d, H, tau, tau_min, st);
  if (g_attn_impl != 1 && T >= 32)
    return gd_attn_mfma_fwd(qk, v, out, io_bf16, csr_tok, win_start, win_len, n_win, T, d, H, tau, tau_min, st);
  AttnArgs A{qk, v, out, csr_tok, win_start, win_len, n_win, d, H, tau, tau_min};
  return io_bf16 ? dispatch_fwd<IoBF16>(A, T, DH, st) : dispatch_fwd<IoF32>(A, T, DH, st);
}

// dtau_part must hold n_win * H floats (one partial per (window, head) at most).
extern "C" int gdmae_window_attention_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, int io_bf16,
                                          float* dtau_part, const int* csr_tok, const int* win_start, const int* win_len,
                                          int n_win, int T, int d, int H, const float* tau, float tau_min, void* stream) {
  if (n_win <= 0) return 0;
  GD_REQUIRE(d % H == 0, "d % H");
  const int DH = d / H;
  GD_REQUIRE(DH == 16 || DH == 32, "head dim must be 16 or 32");
  GD_REQUIRE(T == 16 || T == 32 || T == 64, "T must be 16/32/64");
  GD_REQUIRE(H % (GD_WAVE / T) == 0, "heads must pack evenly into a wavefront");
  hipStream_t st = (hipStream_t)stream;
  if (attn_auto() && T == 16 && io_bf16 && H % 4 == 0)
    return gd_attn_t16_bwd(qk, v, dout, dqk, dv, dtau_part, csr_tok, win_start, win_len, n_win, d, H, tau, tau_min, st);
  if (attn_auto() && T >= 32 && io_bf16)
    return gd_attn_t32_bwd(qk, v, dout, dqk, dv, dtau_part, csr_tok, win_start, win_len, n_win, T, d, H, tau, tau_min, st);
  if (g_attn_impl != 1 && T >= 32)
    return gd_attn_mfma_bwd(qk, v, dout, dqk, dv, io_bf16, dtau_part, csr_tok, win_start, win_len, n_win, T, d, H, tau,
                            tau_min, st);
  AttnBwdArgs A{qk, v, dout, dqk, dv, dtau_part, csr_tok, win_start, win_len, n_win, d, H, tau, tau_min};
  return io_bf16 ? dispatch_bwd<IoBF16>(A, T, DH, st) : dispatch_bwd<IoF32>(A, T, DH, st);
}


// All occupancy levels of one shift (the windows of level l are win_start / win_len [sum_{k<l} n_win[k], ...)), as the layer
// executor issues them: bf16 rows on the matrix-core kernels go out as two launches (T = 16; T = 32 and T = 64 together),
// everything else level by level.  Backward: dtau_part holds sum_l n_win[l] * H partial slots, level after level.
// Both entries are measurement slots (common.h GdTimed: HIP-event brackets on the launch stream, bench.py's roofline leg).
long long g_attn_tokens = 0;   // tokens of the layer whose attention entry is called next (set by the layer executor)
void gd_attn_timing_tokens(long long n) { g_attn_tokens = n; }
static double attn_alg_bytes(int n_levels, const int* n_win, const int* win_len_unused, long long n_tok, int d, int es, int rows_per_tok) {
  long long nw = 0;
  for (int l = 0; l < n_levels; ++l) nw += n_win[l];
  return (double)n_tok * ((double)rows_per_tok * d * es + 4.0) + 8.0 * (double)nw;
}

static bool levels_fast_path(int io_bf16, int n_levels, const int* max_tokens, int H, int d) {
  if (!attn_auto() || !io_bf16 || H % 4 != 0 || d % H != 0 || (d / H != 16 && d / H != 32)) return false;
  for (int l = 0; l < n_levels; ++l)
    if (max_tokens[l] != 16 && max_tokens[l] != 32 && max_tokens[l] != 64) return false;
  return true;
}
// 1 when gdmae_window_attention_levels_fwd with these arguments (and the current gdmae_set_attention_impl) writes `lse` - the caller hands
// `out` / `lse` to the backward only then (a backward after an implementation switch must not read rows the forward never wrote)
extern "C" int gdmae_window_attention_levels_writes_lse(int io_bf16, int n_levels, const int* max_tokens, int d, int H) {
  return (levels_fast_path(io_bf16, n_levels, max_tokens, H, d) && g_attn_impl == 0) ? 1 : 0;
}
extern "C" int gdmae_window_attention_levels_fwd(const void* qk, const void* v, void* out, int io_bf16, const int* csr_tok,
                                                 const int* win_start, const int* win_len, int n_levels, const int* n_win,
                                                 const int* max_tokens, int d, int H, const float* tau, float tau_min, float* lse,
                                                 void* stream) {
  hipStream_t st = (hipStream_t)stream;
  // algorithmic bytes: q, k, v rows read + out row written per token (4 d elements) + CSR; the token count is not an argument
  // of this entry - the caller (encoder_layer.hip) adds it through gd_attn_timing_tokens
  // side bytes (what the launch moves besides by design): the (n_tok, H) fp32 log-sum-exp rows the cooperative path leaves for the backward
  const bool lse_path = levels_fast_path(io_bf16, n_levels, max_tokens, H, d) && g_attn_impl == 0;
  GdTimed timed(GD_T_ATTN_FWD, st, attn_alg_bytes(n_levels, n_win, nullptr, g_attn_tokens, d, io_bf16 ? 2 : 4, 4), 0.0,
                (lse_path && lse) ? 4.0 * H * (double)g_attn_tokens : 0.0);
  if (!levels_fast_path(io_bf16, n_levels, max_tokens, H, d)) {
    int base = 0;
    for (int l = 0; l < n_levels; ++l) {
      const int rc = gdmae_window_attention_fwd(qk, v, out, io_bf16, csr_tok, win_start + base, win_len + base, n_win[l], max_tokens[l], d, H,
                                                tau, tau_min, stream);
      if (rc != 0) return rc;
      base += n_win[l];
    }
    return 0;
  }
  const int* ws[3] = {nullptr, nullptr, nullptr};
  const int* wl[3] = {nullptr, nullptr, nullptr};
  int nw[3] = {0, 0, 0}, base = 0;
  for (int l = 0; l < n_levels; ++l) {
    const int k = max_tokens[l] == 16 ? 0 : (max_tokens[l] == 32 ? 1 : 2);
    GD_REQUIRE(nw[k] == 0, "window attention: two levels with the same token capacity");
    ws[k] = win_start + base; wl[k] = win_len + base; nw[k] = n_win[l];
    base += n_win[l];
  }
  if (g_attn_impl == 0)
    return gd_attn_levels_fwd(qk, v, out, lse, csr_tok, ws[0], wl[0], nw[0], ws[1], wl[1], nw[1], ws[2], wl[2], nw[2], d, H, tau, tau_min, st);
  if (nw[0] > 0) {
    const int rc = gd_attn_t16_fwd(qk, v, out, csr_tok, ws[0], wl[0], nw[0], d, H, tau, tau_min, st);
    if (rc != 0) return rc;
  }
  return gd_attn_t3264_fwd(qk, v, out, csr_tok, ws[1], wl[1], nw[1], ws[2], wl[2], nw[2], d, H, tau, tau_min, st);
}

extern "C" int gdmae_window_attention_levels_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, int io_bf16,
                                                 float* dtau_part, const int* csr_tok, const int* win_start, const int* win_len,
                                                 int n_levels, const int* n_win, const int* max_tokens, int d, int H, const float* tau,
                                                 float tau_min, const void* out, const float* lse, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  // algorithmic bytes = the minimum of an attention backward (q, k, v, dOut rows in, dq, dk, dv rows out: 7 rows per token); side bytes =
  // what this design reads on top of it INSTEAD of re-deriving the softmax statistics - the forward's output rows O (D = dO . O) and the
  // log-sum-exp rows - and the per-(window, head) dtau partial slots it writes
  double side = 0.0;
  {
    long long nwin = 0;
    for (int l = 0; l < n_levels; ++l) nwin += n_win[l];
    side = 4.0 * H * (double)nwin;
    if (levels_fast_path(io_bf16, n_levels, max_tokens, H, d) && g_attn_impl == 0 && out != nullptr && lse != nullptr)
      side += (double)g_attn_tokens * ((double)d * (io_bf16 ? 2 : 4) + 4.0 * H);
  }
  GdTimed timed(GD_T_ATTN_BWD, st, attn_alg_bytes(n_levels, n_win, nullptr, g_attn_tokens, d, io_bf16 ? 2 : 4, 7), 0.0, side);
  if (!levels_fast_path(io_bf16, n_levels, max_tokens, H, d)) {
    int base = 0;
    long long pbase = 0;
    for (int l = 0; l < n_levels; ++l) {
      const int rc = gdmae_window_attention_bwd(qk, v, dout, dqk, dv, io_bf16, dtau_part + pbase, csr_tok, win_start + base, win_len + base,
                                                n_win[l], max_tokens[l], d, H, tau, tau_min, stream);
      if (rc != 0) return rc;
      base += n_win[l];
      pbase += (long long)n_win[l] * H;
    }
    return 0;
  }
  const int* ws[3] = {nullptr, nullptr, nullptr};
  const int* wl[3] = {nullptr, nullptr, nullptr};
  float* part[3] = {nullptr, nullptr, nullptr};
  int nw[3] = {0, 0, 0}, base = 0;
  long long pbase = 0;
  for (int l = 0; l < n_levels; ++l) {
    const int k = max_tokens[l] == 16 ? 0 : (max_tokens[l] == 32 ? 1 : 2);
    GD_REQUIRE(nw[k] == 0, "window attention: two levels with the same token capacity");
    ws[k] = win_start + base; wl[k] = win_len + base; nw[k] = n_win[l]; part[k] = dtau_part + pbase;
    base += n_win[l];
    pbase += (long long)n_win[l] * H;
  }
  if (g_attn_impl == 0 && out != nullptr && lse != nullptr)
    return gd_attn_levels_bwd(qk, v, out, lse, dout, dqk, dv, csr_tok, ws[0], wl[0], nw[0], part[0], ws[1], wl[1], nw[1], part[1], ws[2], wl[2], nw[2],
                              part[2], d, H, tau, tau_min, st);
  if (nw[0] > 0) {
    const int rc = gd_attn_t16_bwd(qk, v, dout, dqk, dv, part[0], csr_tok, ws[0], wl[0], nw[0], d, H, tau, tau_min, st);
    if (rc != 0) return rc;
  }
  return gd_attn_t3264_bwd(qk, v, dout, dqk, dv, csr_tok, ws[1], wl[1], nw[1], part[1], ws[2], wl[2], nw[2], part[2], d, H, tau, tau_min, st);
}
