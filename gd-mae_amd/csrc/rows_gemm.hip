// The decoder's ConvTranspose2d(k = s, stride s, no bias) blocks on token rows (reference
// pcdet/models/backbones_3d/spt_backbone_mae.py:30-45 `decoder_deblocks`, applied to the densified stage maps at :125-131): with kernel
// size = stride the s x s output sites of an input site do not overlap, so on the ACTIVE tokens the block is a row product
//
//     P (n, s*s*cout) = X (n, cin) Wm,   Wm[ci][(q, c)] = w[ci][c][q],  q = dy * s + dx,  w = deconv.weight (cin, cout, s, s)
//
// whose row (token t, q) is the deconvolution output at the full-resolution site of (t, dy, dx).  Forward, input gradient and weight
// gradient ran through hipBLASLt until round 4 (12 library launches + weight permutes / casts per step); here:
//   k_deconv_pack        both MFMA-fragment-ordered images of w (forward (s*s*cout, cin), input gradient (cin, s*s*cout)) in one launch
//   k_rows_gemm          P = X Wm: a workgroup = a row tile x a 128 / 512-column slice, rows guarded (token counts are not padded)
//   k_rows_gemm_kc       dX = dP Wm^T: K = s*s*cout up to 2048 in chunks of <= 512 through one LDS tile, accumulators stay in registers
//   weight gradient      Wm-shaped dWm = X^T dP through the grouped TN kernel (dw_grouped.hip, guarded row loads) and
//   k_deconv_dw_reduce   its fixed-order split-K reduce, accumulated straight into weight.grad's (cin, cout, s, s) layout
// Same MFMA mapping and rounding points as the token GEMMs (tok_tiles.h): bf16 operands, fp32 accumulation, bf16 results.
#include "../../include/gdmae_hip.h"
#include "dw_grouped.h"
#include "tok_tiles.h"

namespace {

// ---- weight images ---------------------------------------------------------------------------------
// image of an (M, K) matrix A: dst[(ks * (M / 32) + mb) * 64 + lane] = 8 bf16 A[mb * 32 + (lane & 31)][ks * 16 + (lane >> 5) * 8 + j]
__global__ __launch_bounds__(256) void k_deconv_pack(const float* __restrict__ w, int cin, int cout, int ss, uint4* __restrict__ fwd,
                                                     uint4* __restrict__ bwd) {
  const int N = ss * cout;
  const int total = (N / 32) * (cin / 16) * 64;          // both images hold N * cin / 8 fragments-of-8
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * total; i += gridDim.x * blockDim.x) {
    const bool is_bwd = i >= total;
    const int e = is_bwd ? i - total : i;
    const int lane = e & 63;
    float f[8];
    if (!is_bwd) {                                        // A = Wm^T: rows r = (q, c), columns ci
      const int MB = N / 32, mb = (e >> 6) % MB, ks = (e >> 6) / MB;
      const int r = mb * 32 + (lane & 31), q = r / cout, c = r - q * cout;
      const int k0 = ks * 16 + (lane >> 5) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = w[((long long)(k0 + j) * cout + c) * ss + q];
    } else {                                              // A = Wm: rows ci, columns k = (q, c)
      const int MB = cin / 32, mb = (e >> 6) % MB, ks = (e >> 6) / MB;
      const int ci = mb * 32 + (lane & 31);
      const int k0 = ks * 16 + (lane >> 5) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j, q = k / cout, c = k - q * cout;
        f[j] = w[((long long)ci * cout + c) * ss + q];
      }
    }
    (is_bwd ? bwd : fwd)[e] = tg_pack8(f);
  }
}

// ---- guarded tile load: rows >= n repeat row n - 1 (finite garbage; their results are never stored) -----------------
template <int W, int ROWS>
__device__ __forceinline__ void rg_load_tile(const unsigned short* __restrict__ src, long long ld, int c0, long long row0, long long n, unsigned char* dst,
                                             int P, int tid) {
  constexpr int CPR = W / 8, RPP = 512 / CPR;
  const int c = tid % CPR, r = tid / CPR;
#pragma unroll
  for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
    const int row = p * RPP + r;
    if (RPP > ROWS && row >= ROWS) break;
    long long g = row0 + row;
    g = g < n ? g : n - 1;
    *(uint4*)(dst + row * P + c * 16) = *(const uint4*)(src + g * ld + c0 + c * 8);
  }
}
template <int W, int ROWS>
__device__ __forceinline__ void rg_store_tile(const unsigned char* src, int P, unsigned short* __restrict__ dst, long long ld, int c0, long long row0,
                                              long long n, int tid) {
  constexpr int CPR = W / 8, RPP = 512 / CPR;
  const int c = tid % CPR, r = tid / CPR;
#pragma unroll
  for (int p = 0; p < (ROWS + RPP - 1) / RPP; ++p) {
    const int row = p * RPP + r;
    if (RPP > ROWS && row >= ROWS) break;
    if (row0 + row < n) *(uint4*)(dst + (row0 + row) * ld + c0 + c * 8) = *(const uint4*)(src + row * P + c * 16);
  }
}

template <int NSL>
struct RgRows {
  static constexpr int value = NSL >= 256 ? 32 : 64;
};

struct RgArgs {
  const unsigned short* X;      // (n, K) bf16
  const uint4* Wp;              // packed (N, K)
  unsigned short* Y;            // (n, N) bf16
  long long n;
  int N;
};

// Y[:, slice] = X Wp[slice]^T; blockIdx.y = column slice of NSL channels
template <int KD, int NSL>
__global__ __launch_bounds__(512, 4) void k_rows_gemm(RgArgs A) {
  constexpr int ROWS = RgRows<NSL>::value;
  constexpr int XP = KD * 2 + 16, SP = NSL * 2 + 16;
  using S = TlShape<KD, NSL, ROWS>;
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long row0 = (long long)blockIdx.x * ROWS;
  const int sl = blockIdx.y;
  TlProd<KD, NSL, ROWS> pr;
  pr.prefetch(A.Wp + (size_t)sl * (NSL / 32) * 64, nullptr, wv, lane, A.N / 32);
  rg_load_tile<KD, ROWS>(A.X, KD, 0, row0, A.n, lds, XP, tid);
  __syncthreads();
  f32x16 acc[S::MPW][S::NPW];
  tl_zero(acc);
  pr.run(lds, XP, wv, lane, acc);
  __syncthreads();
  tl_stage<KD, NSL, ROWS>(acc, nullptr, lds, SP, wv, lane);
  __syncthreads();
  rg_store_tile<NSL, ROWS>(lds, SP, A.Y, A.N, sl * NSL, row0, A.n, tid);
}

struct RkArgs {
  const unsigned short* G;      // (n, KT) bf16
  const uint4* Wp;              // packed (ND, KT)
  unsigned short* Y;            // (n, ND) bf16
  long long n;
  int KT;
};

// Y = G Wp^T with K = KT walked in chunks of KC through one LDS tile
template <int KC, int ND>
__global__ __launch_bounds__(512, 4) void k_rows_gemm_kc(RkArgs A) {
  constexpr int ROWS = RgRows<ND>::value;
  constexpr int XP = KC * 2 + 16, SP = ND * 2 + 16;
  using S = TlShape<KC, ND, ROWS>;
  extern __shared__ __align__(16) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long long row0 = (long long)blockIdx.x * ROWS;
  f32x16 acc[S::MPW][S::NPW];
  tl_zero(acc);
  const int chunks = A.KT / KC;
  for (int ch = 0; ch < chunks; ++ch) {
    TlProd<KC, ND, ROWS> pr;
    pr.prefetch(A.Wp + (size_t)ch * (KC / 16) * (ND / 32) * 64, nullptr, wv, lane);
    if (ch) __syncthreads();                           // every wavefront is done with the previous chunk's tile
    rg_load_tile<KC, ROWS>(A.G, A.KT, ch * KC, row0, A.n, lds, XP, tid);
    __syncthreads();
    pr.run(lds, XP, wv, lane, acc);
  }
  __syncthreads();
  tl_stage<KC, ND, ROWS>(acc, nullptr, lds, SP, wv, lane);
  __syncthreads();
  rg_store_tile<ND, ROWS>(lds, SP, A.Y, ND, 0, row0, A.n, tid);
}

// dW[(ci * cout + c) * ss + q] += sum_s part[s][ci][q * cout + c]   (part: (S, cin, ss * cout) fp32, fixed order)
__global__ __launch_bounds__(256) void k_deconv_dw_reduce(const float* __restrict__ part, int S, int cin, int cout, int ss, float* __restrict__ dW) {
  const long long P = (long long)cin * ss * cout;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < P; e += (long long)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += part[(long long)s * P + e];
    const int n = (int)(e % (ss * cout)), ci = (int)(e / (ss * cout));
    const int q = n / cout, c = n - q * cout;
    dW[((long long)ci * cout + c) * ss + q] += a;
  }
}

template <typename K>
int rg_set_lds(K kernel, int bytes) {
  GD_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  return 0;
}

bool shapes_ok(int cin, int cout, int s) { return (cin == 128 || cin == 256) && cout == 128 && (s == 1 || s == 2 || s == 4); }

}  // namespace

extern "C" size_t gdmae_deconv_rows_packed_bytes(int cin, int cout, int s) { return (size_t)cin * cout * s * s * 2; }

extern "C" int gdmae_deconv_rows_pack(const float* weight, int cin, int cout, int s, void* packed_fwd, void* packed_bwd, void* stream) {
  GD_REQUIRE(shapes_ok(cin, cout, s), "deconv_rows: cin 128 / 256, cout 128, stride 1 / 2 / 4");
  const int total = cin * cout * s * s / 8;
  hipLaunchKernelGGL(k_deconv_pack, dim3(gd_div_up(2ll * total, 256) < 1024 ? gd_div_up(2ll * total, 256) : 1024), dim3(256), 0, (hipStream_t)stream,
                     weight, cin, cout, s * s, (uint4*)packed_fwd, (uint4*)packed_bwd);
  GD_LAUNCH_CHECK();
  return 0;
}

// P (n, s*s*cout) bf16 = X (n, cin) bf16 x the forward image
extern "C" int gdmae_deconv_rows_fwd(const void* X, long long n, int cin, int cout, int s, const void* packed_fwd, void* P, void* stream) {
  GD_REQUIRE(shapes_ok(cin, cout, s), "deconv_rows: cin 128 / 256, cout 128, stride 1 / 2 / 4");
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int N = s * s * cout;
  RgArgs A{(const unsigned short*)X, (const uint4*)packed_fwd, (unsigned short*)P, n, N};
  GdTimed timed(GD_T_ROWS_GEMM, st, 2.0 * n * (cin + N) + 2.0 * cin * N, 2.0 * n * cin * N);
#define RG_CASE(K_, NSL_)                                                                                       \
  {                                                                                                             \
    constexpr int rows = RgRows<NSL_>::value;                                                                   \
    constexpr int lds = rows * ((K_ > NSL_ ? K_ : NSL_) * 2 + 16);                                              \
    static bool once = false;                                                                                   \
    if (!once) { if (int rc = rg_set_lds(k_rows_gemm<K_, NSL_>, lds)) return rc; once = true; }                 \
    hipLaunchKernelGGL((k_rows_gemm<K_, NSL_>), dim3((unsigned)gd_div_up(n, rows), N / NSL_), dim3(512), lds, st, A); \
  }
  if (cin == 128 && N == 128) RG_CASE(128, 128)
  else if (cin == 128) RG_CASE(128, 512)
  else if (N == 128) RG_CASE(256, 128)
  else RG_CASE(256, 512)
#undef RG_CASE
  GD_LAUNCH_CHECK();
  return 0;
}

// dX (n, cin) bf16 = dP (n, s*s*cout) bf16 x the input-gradient image
extern "C" int gdmae_deconv_rows_bwd_input(const void* dP, long long n, int cin, int cout, int s, const void* packed_bwd, void* dX, void* stream) {
  GD_REQUIRE(shapes_ok(cin, cout, s), "deconv_rows: cin 128 / 256, cout 128, stride 1 / 2 / 4");
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int KT = s * s * cout;
  RkArgs A{(const unsigned short*)dP, (const uint4*)packed_bwd, (unsigned short*)dX, n, KT};
  GdTimed timed(GD_T_ROWS_GEMM, st, 2.0 * n * (cin + KT) + 2.0 * cin * KT, 2.0 * n * cin * KT);
#define RK_CASE(KC_, ND_)                                                                                       \
  {                                                                                                             \
    constexpr int rows = RgRows<ND_>::value;                                                                    \
    constexpr int lds = rows * ((KC_ > ND_ ? KC_ : ND_) * 2 + 16);                                              \
    static bool once = false;                                                                                   \
    if (!once) { if (int rc = rg_set_lds(k_rows_gemm_kc<KC_, ND_>, lds)) return rc; once = true; }              \
    hipLaunchKernelGGL((k_rows_gemm_kc<KC_, ND_>), dim3((unsigned)gd_div_up(n, rows)), dim3(512), lds, st, A);  \
  }
  if (KT == 128 && cin == 128) RK_CASE(128, 128)
  else if (KT == 128) RK_CASE(128, 256)
  else if (cin == 128) RK_CASE(512, 128)
  else RK_CASE(512, 256)
#undef RK_CASE
  GD_LAUNCH_CHECK();
  return 0;
}

// weight.grad (cin, cout, s, s) fp32 += X^T dP; workspace: gdmae_deconv_rows_dw_workspace_bytes(n, cin, cout, s)
static int dw_slices(long long n, int cin, int N, long long* n_pad) { return gd_dw_pick(n, (cin / 128) * (N / 128), 512, n_pad); }
extern "C" size_t gdmae_deconv_rows_dw_workspace_bytes(long long n, int cin, int cout, int s) {
  long long n_pad = 0;
  const int S = dw_slices(n, cin, s * s * cout, &n_pad);
  return gd_align((size_t)S * cin * s * s * cout * sizeof(float));
}
extern "C" int gdmae_deconv_rows_bwd_weight(const void* X, const void* dP, long long n, int cin, int cout, int s, float* dW, void* workspace,
                                            void* stream) {
  GD_REQUIRE(shapes_ok(cin, cout, s), "deconv_rows: cin 128 / 256, cout 128, stride 1 / 2 / 4");
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int N = s * s * cout;
  long long n_pad = 0;
  const int S = dw_slices(n, cin, N, &n_pad);
  GdDwGroup Gp;
  Gp.n_jobs = 1;
  Gp.job[0] = GdDwJob{X, dP, cin, N, (float*)workspace, nullptr, 0, nullptr, 0, 0};
  Gp.guard_rows = 1;                                     // neither operand is allocated beyond n rows
  if (int rc = gd_dw_grouped_s(st, Gp, n_pad, n, S)) return rc;
  hipLaunchKernelGGL(k_deconv_dw_reduce, dim3(gd_div_up((long long)cin * N, 256)), dim3(256), 0, st, (const float*)workspace, S, cin, cout, s * s, dW);
  GD_LAUNCH_CHECK();
  return 0;
}
