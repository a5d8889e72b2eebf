// Weight gradients of the token-wise linear layers:  dW (M, K) += G^T (M, rows) X (rows, K)  with rows = 20-40 k tokens,
// M, K in {128, 256, 512}, G / X bf16 row-major (the backward of the nn.Linear layers of
// sst_basic_block.py:57-84 / cosine_msa.py in-/out-projections).
//
// The contraction runs over the ROW index, which is the slow axis of both operands ("TN" product): an MFMA fragment needs
// 8 consecutive rows of one column.  The tiles are staged row-major in LDS with 16-byte coalesced loads and the fragments
// are assembled by 16-bit LDS reads straight into register halves (ds_read_u16_d16 / _d16_hi): consecutive lanes read
// consecutive columns of one row (conflict-free), no transposed copy of the activations is ever written.
//
// One workgroup (4 wavefronts, 2 x 2 blocks of 64 x 64) owns a 128 x 128 tile of dW and a slice of the rows (split-K over
// workgroups: tiles x slices ~ one workgroup per CU), loops over 64-row chunks with register-staged double buffering and
// writes its fp32 partial tile; the partial tiles are summed in a fixed order by the caller's reduce kernel
// (k_splitk_acc_jobs, encoder_layer.hip), so the result is deterministic.  The workgroups of the first K tile also
// produce the column sums of G over their slice (= the bias gradient) from the values they stage anyway.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
union DwFrag {
  unsigned u[4];
  bf16x8 v;
};

#define DW_TILE 128
#define DW_CHUNK 64                 // rows per LDS stage
#define DW_PITCH 272                // bytes per staged row: 128 bf16 + 16

struct DwArgs {
  const unsigned short* G;          // (rows, M)
  const unsigned short* X;          // (rows, K)
  int M, K;
  long long rows_per_slice;         // multiple of DW_CHUNK
  float* part;                      // (S, M, K) fp32 partial products
  float* colpart;                   // optional (S, M): column sums of G per slice
};

__device__ inline void dw_stage_load(const unsigned short* base, int ld, long long row0, int col0, int tid, uint4 (&q)[4]) {
  const int c = tid & 15, r = tid >> 4;
#pragma unroll
  for (int p = 0; p < 4; ++p) q[p] = *(const uint4*)(base + (row0 + p * 16 + r) * ld + col0 + c * 8);
}
__device__ inline void dw_stage_store(unsigned char* lds, int tid, const uint4 (&q)[4]) {
  const int c = tid & 15, r = tid >> 4;
#pragma unroll
  for (int p = 0; p < 4; ++p) *(uint4*)(lds + (p * 16 + r) * DW_PITCH + c * 16) = q[p];
}
// 8 consecutive rows (row0 .. row0 + 7) of one column -> one MFMA operand fragment
__device__ inline bf16x8 dw_frag(const unsigned char* p) {
  DwFrag f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned lo = *(const unsigned short*)(p + (2 * j) * DW_PITCH);
    const unsigned hi = *(const unsigned short*)(p + (2 * j + 1) * DW_PITCH);
    f.u[j] = lo | (hi << 16);
  }
  return f.v;
}

__global__ __launch_bounds__(256, 2) void k_dw_gemm(DwArgs A) {
  __shared__ __align__(16) unsigned char lds[2][2][DW_CHUNK * DW_PITCH];   // [buffer][G | X]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tm = blockIdx.x, tk = blockIdx.y, s = blockIdx.z;
  const long long r0 = (long long)s * A.rows_per_slice;
  const int nchunk = (int)(A.rows_per_slice / DW_CHUNK);
  const int wm = wv >> 1, wk = wv & 1;                 // 64 x 64 block of this wavefront
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
  const bool want_cs = A.colpart != nullptr && tk == 0;

  uint4 qg[4], qx[4];
  dw_stage_load(A.G, A.M, r0, tm * DW_TILE, tid, qg);
  dw_stage_load(A.X, A.K, r0, tk * DW_TILE, tid, qx);
  for (int c = 0; c < nchunk; ++c) {
    unsigned char* bg = lds[c & 1][0];
    unsigned char* bx = lds[c & 1][1];
    dw_stage_store(bg, tid, qg);
    dw_stage_store(bx, tid, qx);
    if (want_cs) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const unsigned w[4] = {qg[p].x, qg[p].y, qg[p].z, qg[p].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          cs[2 * j] += __uint_as_float(w[j] << 16);
          cs[2 * j + 1] += __uint_as_float(w[j] & 0xFFFF0000u);
        }
      }
    }
    __syncthreads();                                   // chunk c staged; buffer (c + 1) & 1 was last read in iteration c - 1
    if (c + 1 < nchunk) {
      dw_stage_load(A.G, A.M, r0 + (long long)(c + 1) * DW_CHUNK, tm * DW_TILE, tid, qg);
      dw_stage_load(A.X, A.K, r0 + (long long)(c + 1) * DW_CHUNK, tk * DW_TILE, tid, qx);
    }
    const unsigned char* pa = bg + ((lane >> 5) * 8) * DW_PITCH + (wm * 64 + (lane & 31)) * 2;
    const unsigned char* pb = bx + ((lane >> 5) * 8) * DW_PITCH + (wk * 64 + (lane & 31)) * 2;
#pragma unroll
    for (int ks = 0; ks < DW_CHUNK / 16; ++ks) {
      const bf16x8 a0 = dw_frag(pa + ks * 16 * DW_PITCH), a1 = dw_frag(pa + ks * 16 * DW_PITCH + 64);
      const bf16x8 b0 = dw_frag(pb + ks * 16 * DW_PITCH), b1 = dw_frag(pb + ks * 16 * DW_PITCH + 64);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  // ---- partial tile: D[row = m by register, column = k by lane]
  float* out = A.part + ((long long)s * A.M + tm * DW_TILE + wm * 64) * A.K + tk * DW_TILE + wk * 64;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        out[(long long)m * A.K + j * 32 + (lane & 31)] = acc[i][j][e];
      }
  if (want_cs) {
    __syncthreads();
    float* red = (float*)lds;                         // (16 row groups, 128 columns)
    const int c = tid & 15, r = tid >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) red[r * DW_TILE + c * 8 + j] = cs[j];
    __syncthreads();
    if (tid < DW_TILE) {
      float t = 0.f;
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) t += red[rr * DW_TILE + tid];
      A.colpart[(long long)s * A.M + tm * DW_TILE + tid] = t;
    }
  }
}

bool gd_dw_gemm_supported(int M, int K) { return M % DW_TILE == 0 && K % DW_TILE == 0 && M <= 1024 && K <= 1024; }

// number of row slices for a problem (rows a multiple of 2048): tiles x slices ~ 256 workgroups, >= 2 chunks per slice
int gd_dw_gemm_slices(long long rows, int M, int K) {
  const int tiles = (M / DW_TILE) * (K / DW_TILE);
  int S = 256 / tiles;
  if (S < 1) S = 1;
  while (S > 1 && (rows % ((long long)S * DW_CHUNK) != 0 || rows / S < 2 * DW_CHUNK)) S >>= 1;
  return S;
}

// part: (S, M, K) fp32, colpart: (S, M) fp32 or null; S = gd_dw_gemm_slices(rows, M, K)
int gd_dw_gemm(hipStream_t st, const void* G, const void* X, long long rows, int M, int K, float* part, float* colpart) {
  GD_REQUIRE(gd_dw_gemm_supported(M, K) && rows % 2048 == 0 && rows > 0, "dw_gemm: unsupported problem");
  const int S = gd_dw_gemm_slices(rows, M, K);
  DwArgs A;
  A.G = (const unsigned short*)G; A.X = (const unsigned short*)X; A.M = M; A.K = K;
  A.rows_per_slice = rows / S;
  A.part = part; A.colpart = colpart;
  hipLaunchKernelGGL(k_dw_gemm, dim3(M / DW_TILE, K / DW_TILE, S), dim3(256), 0, st, A);
  GD_LAUNCH_CHECK();
  return 0;
}

// C ABI (tests): dW (M, K) fp32 = G^T X, dbias (M) fp32 = column sums of G (optional); workspace >= gdmae_dw_gemm_workspace_bytes
extern "C" size_t gdmae_dw_gemm_workspace_bytes(long long rows, int M, int K) {
  const int S = gd_dw_gemm_slices(rows, M, K);
  return gd_align((size_t)S * M * K * sizeof(float)) + gd_align((size_t)S * M * sizeof(float));
}
__global__ __launch_bounds__(256) void k_dw_reduce(const float* __restrict__ part, int S, long long P, float* __restrict__ dst) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += part[(long long)s * P + i];
    dst[i] = a;
  }
}
extern "C" int gdmae_dw_gemm(const void* G, const void* X, long long rows, int M, int K, float* dW, float* dbias, void* workspace,
                             void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GD_REQUIRE(gd_dw_gemm_supported(M, K) && rows % 2048 == 0 && rows > 0, "dw_gemm: M, K multiples of 128, rows a multiple of 2048");
  const int S = gd_dw_gemm_slices(rows, M, K);
  float* part = (float*)workspace;
  float* colpart = (float*)((char*)workspace + gd_align((size_t)S * M * K * sizeof(float)));
  {
    const int rc = gd_dw_gemm(st, G, X, rows, M, K, part, dbias ? colpart : nullptr);
    if (rc != 0) return rc;
  }
  hipLaunchKernelGGL(k_dw_reduce, dim3(256), dim3(256), 0, st, (const float*)part, S, (long long)M * K, dW);
  GD_LAUNCH_CHECK();
  if (dbias) {
    hipLaunchKernelGGL(k_dw_reduce, dim3(4), dim3(256), 0, st, (const float*)colpart, S, (long long)M, dbias);
    GD_LAUNCH_CHECK();
  }
  return 0;
}
