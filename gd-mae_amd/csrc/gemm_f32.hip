// fp32 GEMM of the PARITY mode on the matrix cores (the fp32 mode carries north_star's 1e-4 bound): every product the
// fp32 path used to hand to hipBLASLt - token GEMMs of the encoder layers, im2col products of the sparse convolutions, their input
// gradients, the batched split-K weight gradients, the deconvolution / prediction-head linears - through one hand-written kernel
// with EXACT fp32 arithmetic: v_mfma_f32_32x32x2_f32 multiplies fp32 operands and accumulates in fp32 (no bf16 / tf32 rounding of
// the inputs), in a fixed order that does not depend on timing (the library's algorithm choice did: csrc/gemm.hip).
//
//   column-major  C (M x N, ldc) = op(A) op(B) [+ bias(M)],  op = identity or transpose, strided batches
//
// One workgroup (4 wavefronts = 2 x 2 blocks of 32 x 32) owns a 64 x 64 tile of C; k advances in tiles of 16 staged in LDS as
// As[k][m] / Bs[k][n] (the MFMA's operand order: lane = (row | column, k parity)), the next tile's elements are in flight in
// registers while the current one is multiplied.  Loads follow whichever index is contiguous in memory for the given transposes;
// (16-byte loads when base, leading dimension and batch stride allow); all extents are guarded, so any M, N, K is served.  A 64 x 128
// tile with two column blocks per wavefront measured slower on the step (47.8 vs 45.4 ms: the products are skinny, N = 128 - 512).  Throughput is secondary here (the parity mode runs at a quarter of the bf16
// mode's rate whatever the GEMMs do); what matters is that the mode's arithmetic is this repository's own.
#include "common.h"
#include "gemm.h"
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {
constexpr int kTM = 64, kTN = 64, kTK = 16, kPad = 4;

struct GfArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K, lda, ldb, ldc, ta, tb;
  long long sA, sB, sC;
  int swap_xy;              // blockIdx.x walks the N tiles (the longer extent goes to x: y is limited to 65535)
  int vec_a, vec_b;         // 16-byte loads along the contiguous index of op(A) / op(B) (base, leading dimension and batch stride aligned)
};

__global__ __launch_bounds__(256) void k_gemm_f32(GfArgs G) {
  __shared__ float As[kTK][kTM + kPad];
  __shared__ float Bs[kTK][kTN + kPad];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
  const int m0 = (G.swap_xy ? blockIdx.y : blockIdx.x) * kTM, n0 = (G.swap_xy ? blockIdx.x : blockIdx.y) * kTN;
  const float* __restrict__ A = G.A + (long long)blockIdx.z * G.sA;
  const float* __restrict__ B = G.B + (long long)blockIdx.z * G.sB;
  float* __restrict__ C = G.C + (long long)blockIdx.z * G.sC;
  // element (m, k) of op(A): !ta -> A[m + k lda] (m contiguous), ta -> A[k + m lda] (k contiguous); same for op(B)(k, n)
  int am[4], ak[4], bn[4], bk[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (!G.ta) { am[i] = tid & 63; ak[i] = (tid >> 6) + 4 * i; } else { ak[i] = tid & 15; am[i] = (tid >> 4) + 16 * i; }
    if (G.tb) { bn[i] = tid & 63; bk[i] = (tid >> 6) + 4 * i; } else { bk[i] = tid & 15; bn[i] = (tid >> 4) + 16 * i; }
  }
  // vectorised variants: one float4 per thread along the contiguous index (the four elements share the other index)
  if (G.vec_a) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!G.ta) { am[i] = 4 * (tid & 15) + i; ak[i] = tid >> 4; } else { ak[i] = 4 * (tid & 3) + i; am[i] = tid >> 2; }
    }
  }
  if (G.vec_b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (G.tb) { bn[i] = 4 * (tid & 15) + i; bk[i] = tid >> 4; } else { bk[i] = 4 * (tid & 3) + i; bn[i] = tid >> 2; }
    }
  }
  auto load_a = [&](int k0, float (&v)[4]) {
    if (G.vec_a) {
      const int m = m0 + am[0], k = k0 + ak[0];
      const bool in = G.ta ? (m < G.M && k + 3 < G.K) : (m + 3 < G.M && k < G.K);
      if (in) {
        const float4 q = *(const float4*)(G.ta ? A + k + (long long)m * G.lda : A + m + (long long)k * G.lda);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + am[i], k = k0 + ak[i];
      v[i] = (m < G.M && k < G.K) ? (G.ta ? A[k + (long long)m * G.lda] : A[m + (long long)k * G.lda]) : 0.f;
    }
  };
  auto load_b = [&](int k0, float (&v)[4]) {
    if (G.vec_b) {
      const int n = n0 + bn[0], k = k0 + bk[0];
      const bool in = G.tb ? (n + 3 < G.N && k < G.K) : (n < G.N && k + 3 < G.K);
      if (in) {
        const float4 q = *(const float4*)(G.tb ? B + n + (long long)k * G.ldb : B + k + (long long)n * G.ldb);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = n0 + bn[i], k = k0 + bk[i];
      v[i] = (n < G.N && k < G.K) ? (G.tb ? B[n + (long long)k * G.ldb] : B[k + (long long)n * G.ldb]) : 0.f;
    }
  };
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float va[4], vb[4];
  load_a(0, va);
  load_b(0, vb);
  for (int k0 = 0; k0 < G.K; k0 += kTK) {
    __syncthreads();                                   // the previous tile has been consumed
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      As[ak[i]][am[i]] = va[i];
      Bs[bk[i]][bn[i]] = vb[i];
    }
    __syncthreads();
    if (k0 + kTK < G.K) {                              // next tile: in flight behind the MFMAs
      load_a(k0 + kTK, va);
      load_b(k0 + kTK, vb);
    }
#pragma unroll
    for (int kk = 0; kk < kTK / 2; ++kk) {
      const float a = As[2 * kk + (lane >> 5)][wm * 32 + (lane & 31)];
      const float b = Bs[2 * kk + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  // acc[4 q + e]: row m = 8 q + e + 4 (lane >> 5) of the wavefront's block, column n = lane & 31
  const int n = n0 + wn * 32 + (lane & 31);
  if (n < G.N) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + wm * 32 + 8 * q + 4 * (lane >> 5);
      float v[4] = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (m + e < G.M) C[(m + e) + (long long)n * G.ldc] = v[e] + (G.bias ? G.bias[m + e] : 0.f);
    }
  }
}
// K = 0: the empty product - C = bias (or 0), same element addressing as the product's epilogue
__global__ __launch_bounds__(256) void k_gemm_f32_k0(float* __restrict__ C, const float* __restrict__ bias, int M, int N, int ldc, long long sC, int batch) {
  const long long total = (long long)M * N * batch;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i % M);
    const long long r = i / M;
    const int n = (int)(r % N);
    const long long b = r / N;
    C[m + (long long)n * ldc + b * sC] = bias ? bias[m] : 0.f;
  }
}
}  // namespace

int gd_gemm_f32(hipStream_t st, bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, int batch, long long sA, long long sB, long long sC) {
  GD_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 1 && batch <= 65535, "gemm_f32: bad extents");
  if (M == 0 || N == 0) return 0;                      // nothing to write (an empty stage / batch)
  if (K == 0) {
    const long long total = (long long)M * N * batch;
    hipLaunchKernelGGL(k_gemm_f32_k0, dim3((unsigned)(total / 256 + 1 > 4096 ? 4096 : total / 256 + 1)), dim3(256), 0, st, C, bias, M, N, ldc, sC, batch);
    GD_LAUNCH_CHECK();
    return 0;
  }
  const unsigned gm = (unsigned)gd_div_up(M, kTM), gn = (unsigned)gd_div_up(N, kTN);
  const int swap = gn > gm;
  auto aligned = [](const float* p, int ld, long long stride) { return ((uintptr_t)p & 15) == 0 && ld % 4 == 0 && stride % 4 == 0; };
  GfArgs G{A, B, C, bias, M, N, K, lda, ldb, ldc, ta ? 1 : 0, tb ? 1 : 0, sA, sB, sC, swap, aligned(A, lda, sA) ? 1 : 0, aligned(B, ldb, sB) ? 1 : 0};
  const dim3 grid(swap ? gn : gm, swap ? gm : gn, (unsigned)batch);
  GD_REQUIRE(grid.y <= 65535, "gemm_f32: both extents too large for one launch");
  hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, st, G);
  GD_LAUNCH_CHECK();
  return 0;
}
