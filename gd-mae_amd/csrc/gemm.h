// hipBLASLt GEMM front end of the library (gemm.hip): column-major C(MxN) = op(A) op(B) [+ bias(M)], one cached
// algorithm per problem shape.  Internal header (the C ABI of the row-major entry points is in include/gdmae_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/library_types.h>

#define GD_LT_WORKSPACE ((size_t)32u << 20)

// `loose`: extents differ from call to call (token / point / site counts); the algorithm is cached per bucket of
// extents (3 significant bits) and the matrix layouts are rebuilt per call.  Otherwise exact shapes are cached.
int gd_gemm(hipStream_t st, bool ta, bool tb, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
            hipDataType tab, hipDataType tc, const void* bias, int batch, long long sA, long long sB, long long sC, void* ws,
            size_t ws_bytes, bool loose = false);
// dst[i] (+)= sum_{s < S} part[s * P + i], P % 4 == 0
int gd_splitk_acc(hipStream_t st, const float* part, int S, long long P, float* dst, int accumulate);

// fp32 operands, fp32 result, exact fp32 MFMA arithmetic in a fixed order (gemm_f32.hip): what gd_gemm runs for HIP_R_32F
// (GDMAE_GEMM_F32=0: hipBLASLt instead, A/B reference)
int gd_gemm_f32(hipStream_t st, bool ta, bool tb, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, int batch, long long sA, long long sB, long long sC);
