// Row gather / scatter (rulebook im2col, densify, gather-at-pillars) and the per-pillar segmented max.
//
// Replaces (SURVEY.md §8 rows a4 scatter part, a6 data movement, a16 gather):
//   torch_scatter.scatter_max                reference pcdet/models/backbones_3d/vfe/dyn_vfe.py:109
//   spconv gather / scatter around its GEMMs reference call sites pcdet/utils/spconv_utils.py:41-43
//   SparseConvTensor.dense()                 pcdet/models/backbones_3d/spt_backbone_mae.py:128
//   spatial_features.permute(0,2,3,1)[b,y,x] spt_backbone_mae.py:141-143
//
// All of these are pure HBM data movement: rows are moved as 16-byte vectors, one row segment per
// lane group, fully coalesced; the kernels are dtype agnostic (row = bytes).  The segmented max walks
// each pillar's CSR list in canonical (ascending point index) order with lanes on channels, so ties
// resolve to the lowest point index and no atomics are involved.
#include "common.h"

// out[r, k, :] = idx[r*K + k] >= 0 ? src[idx[r*K+k], :] : 0        (row_vec = row bytes / 16)
// (the table side - src of a gather, dst of a scatter - may be a column slice of wider rows:
//  table row j starts at vector j * tab_stride + tab_off)
__global__ __launch_bounds__(256) void k_gather_rows(const uint4* __restrict__ src, const int* __restrict__ idx,
                                                     long long n_slots, int row_vec, int tab_stride, int tab_off,
                                                     uint4* __restrict__ out) {
  const long long total = n_slots * row_vec;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long s = i / row_vec;
    const int v = (int)(i % row_vec);
    const int j = idx[s];
    uint4 val = make_uint4(0, 0, 0, 0);
    if (j >= 0) val = src[(long long)j * tab_stride + tab_off + v];
    out[i] = val;
  }
}

// dst[idx[r], :] = src[r, :]   (idx unique; rows with idx < 0 are skipped)
__global__ __launch_bounds__(256) void k_scatter_rows(const uint4* __restrict__ src, const int* __restrict__ idx,
                                                      long long n_rows, int row_vec, int tab_stride, int tab_off,
                                                      uint4* __restrict__ dst) {
  const long long total = n_rows * row_vec;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long s = i / row_vec;
    const int v = (int)(i % row_vec);
    const int j = idx[s];
    if (j >= 0) dst[(long long)j * tab_stride + tab_off + v] = src[i];
  }
}

static inline int gd_grid_for(long long total, int block = 256, int cap = 16384) {
  int g = gd_div_up(total > 0 ? total : 1, block);
  return g > cap ? cap : g;
}

extern "C" int gdmae_gather_rows(const void* src, const int* idx, long long n_slots, int row_bytes, void* out,
                                 void* stream) {
  GD_REQUIRE(row_bytes % 16 == 0, "row_bytes must be a multiple of 16");
  if (n_slots <= 0) return 0;
  const int rv = row_bytes / 16;
  hipLaunchKernelGGL(k_gather_rows, dim3(gd_grid_for(n_slots * rv)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)src, idx, n_slots, rv, rv, 0, (uint4*)out);
  GD_LAUNCH_CHECK();
  return 0;
}

// column-slice variants: the table's rows are table_row_bytes wide and the slice starts at table_col_bytes
extern "C" int gdmae_gather_rows_strided(const void* src, const int* idx, long long n_slots, int row_bytes,
                                         int table_row_bytes, int table_col_bytes, void* out, void* stream) {
  GD_REQUIRE(row_bytes % 16 == 0 && table_row_bytes % 16 == 0 && table_col_bytes % 16 == 0, "16-byte granularity");
  GD_REQUIRE(table_col_bytes + row_bytes <= table_row_bytes, "slice exceeds the table row");
  if (n_slots <= 0) return 0;
  const int rv = row_bytes / 16;
  hipLaunchKernelGGL(k_gather_rows, dim3(gd_grid_for(n_slots * rv)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)src, idx, n_slots, rv, table_row_bytes / 16, table_col_bytes / 16, (uint4*)out);
  GD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gdmae_scatter_rows_strided(const void* src, const int* idx, long long n_rows, int row_bytes,
                                          int table_row_bytes, int table_col_bytes, void* dst, void* stream) {
  GD_REQUIRE(row_bytes % 16 == 0 && table_row_bytes % 16 == 0 && table_col_bytes % 16 == 0, "16-byte granularity");
  GD_REQUIRE(table_col_bytes + row_bytes <= table_row_bytes, "slice exceeds the table row");
  if (n_rows <= 0) return 0;
  const int rv = row_bytes / 16;
  hipLaunchKernelGGL(k_scatter_rows, dim3(gd_grid_for(n_rows * rv)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)src, idx, n_rows, rv, table_row_bytes / 16, table_col_bytes / 16, (uint4*)dst);
  GD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gdmae_scatter_rows(const void* src, const int* idx, long long n_rows, int row_bytes, void* dst,
                                  void* stream) {
  GD_REQUIRE(row_bytes % 16 == 0, "row_bytes must be a multiple of 16");
  if (n_rows <= 0) return 0;
  const int rv = row_bytes / 16;
  hipLaunchKernelGGL(k_scatter_rows, dim3(gd_grid_for(n_rows * rv)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)src, idx, n_rows, rv, rv, 0, (uint4*)dst);
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// segmented max over the pillar CSR: out[p, c] = max_{i in pillar p} x[i, c], arg[p, c] = that i
// (lowest i on ties).  One wavefront per pillar, lanes on channels (float2 per lane for C = 128).
// ------------------------------------------------------------------------------------------
// LPP = lanes per point = C / 4 (each lane owns 4 consecutive channels as one 16-byte load); the 64 / LPP lane
// groups of the wavefront walk interleaved points of the pillar (several independent row loads in flight),
// then the groups are combined with xor-shuffles (ties -> lowest point id, i.e. canonical order).
template <int LPP>
__global__ __launch_bounds__(256) void k_segment_max(const float* __restrict__ x, const int* __restrict__ pt_off,
                                                     const int* __restrict__ csr, int M, float* __restrict__ out,
                                                     int* __restrict__ arg) {
  constexpr int C = LPP * 4;
  constexpr int G = GD_WAVE / LPP;  // points in flight per step
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  const int grp = lane / LPP, cl = lane % LPP;
  for (int p = blockIdx.x * 4 + wib; p < M; p += gridDim.x * 4) {
    const int off = pt_off[p];
    const int cnt = pt_off[p + 1] - off;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
#pragma unroll 4
    for (int j = grp; j < cnt; j += G) {
      const int i = csr[off + j];
      const float4 v = *reinterpret_cast<const float4*>(x + (long long)i * C + cl * 4);
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (vv[e] > best[e] || bi[e] == 0x7fffffff) {   // first point always taken (also NaN / -inf rows)
          best[e] = vv[e];
          bi[e] = i;
        }
      }
    }
#pragma unroll
    for (int d = LPP; d < GD_WAVE; d <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ob = __shfl_xor(best[e], d, GD_WAVE);
        const int oi = __shfl_xor(bi[e], d, GD_WAVE);
        const bool take = (oi != 0x7fffffff) && (bi[e] == 0x7fffffff || ob > best[e] || (ob == best[e] && oi < bi[e]));
        if (take) {
          best[e] = ob;
          bi[e] = oi;
        }
      }
    }
    if (grp == 0) {
      *reinterpret_cast<float4*>(out + (long long)p * C + cl * 4) = make_float4(best[0], best[1], best[2], best[3]);
      *reinterpret_cast<int4*>(arg + (long long)p * C + cl * 4) = make_int4(bi[0], bi[1], bi[2], bi[3]);
    }
  }
}

// dx[i, c] = (arg[inv[i], c] == i) ? dout[inv[i], c] : 0
__global__ __launch_bounds__(256) void k_segment_max_bwd(const float* __restrict__ dout, const int* __restrict__ arg,
                                                         const int* __restrict__ inv, long long N, int C,
                                                         float* __restrict__ dx) {
  const long long total = N * C;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / C;
    const int c = (int)(e % C);
    const long long q = (long long)inv[i] * C + c;
    dx[e] = (arg[q] == (int)i) ? dout[q] : 0.f;
  }
}

extern "C" int gdmae_segment_max(const float* x, const int* pt_off, const int* csr, int M, int C, float* out, int* arg,
                                 void* stream) {
  if (M <= 0) return 0;
  const dim3 grid(gd_grid_for(M, 4, 16384)), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (C) {
    case 16: hipLaunchKernelGGL((k_segment_max<4>), grid, block, 0, st, x, pt_off, csr, M, out, arg); break;
    case 32: hipLaunchKernelGGL((k_segment_max<8>), grid, block, 0, st, x, pt_off, csr, M, out, arg); break;
    case 64: hipLaunchKernelGGL((k_segment_max<16>), grid, block, 0, st, x, pt_off, csr, M, out, arg); break;
    case 128: hipLaunchKernelGGL((k_segment_max<32>), grid, block, 0, st, x, pt_off, csr, M, out, arg); break;
    case 256: hipLaunchKernelGGL((k_segment_max<64>), grid, block, 0, st, x, pt_off, csr, M, out, arg); break;
    default: GD_REQUIRE(false, "segment_max supports C in {16, 32, 64, 128, 256}");
  }
  GD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gdmae_segment_max_bwd(const float* dout, const int* arg, const int* inv, long long N, int C, float* dx,
                                     void* stream) {
  if (N <= 0) return 0;
  hipLaunchKernelGGL(k_segment_max_bwd, dim3(gd_grid_for(N * C)), dim3(256), 0, (hipStream_t)stream, dout, arg, inv, N, C,
                     dx);
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// point decoration (dyn_vfe.py:85-105): feat = [xyz - pillar centre (3), xyz+features (F), xyz - pillar mean (3)]
// computed with the reference's op order ((c + 0.5) * vs + lo, no FMA contraction).
// ------------------------------------------------------------------------------------------
struct DecoParams {
  float lo[3], vs[3];
  int ncols;
};
__global__ __launch_bounds__(256) void k_decorate(const float* __restrict__ pts, const long long* __restrict__ coords,
                                                  const int* __restrict__ inv, const float* __restrict__ mean, long long N,
                                                  DecoParams P, float* __restrict__ out) {
  const int F = P.ncols - 1;
  const int D = F + 6;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
    const float* r = pts + i * P.ncols;
    const long long* c = coords + 4 * i;  // b, z, y, x
    const float* m = mean + (long long)inv[i] * F;
    float* o = out + i * D;
    const float x = r[1], y = r[2], z = r[3];
    o[0] = __fsub_rn(x, __fadd_rn(__fmul_rn(__fadd_rn((float)c[3], 0.5f), P.vs[0]), P.lo[0]));
    o[1] = __fsub_rn(y, __fadd_rn(__fmul_rn(__fadd_rn((float)c[2], 0.5f), P.vs[1]), P.lo[1]));
    o[2] = __fsub_rn(z, __fadd_rn(__fmul_rn(__fadd_rn((float)c[1], 0.5f), P.vs[2]), P.lo[2]));
    for (int k = 0; k < F; ++k) o[3 + k] = r[1 + k];
    o[3 + F] = __fsub_rn(x, m[0]);
    o[4 + F] = __fsub_rn(y, m[1]);
    o[5 + F] = __fsub_rn(z, m[2]);
  }
}

extern "C" int gdmae_decorate_points(const float* points, const long long* point_coords, const int* inverse32,
                                     const float* pillar_mean, long long N, int n_cols, const float* lo, const float* vs,
                                     float* out, void* stream) {
  if (N <= 0) return 0;
  DecoParams P;
  for (int i = 0; i < 3; ++i) {
    P.lo[i] = lo[i];
    P.vs[i] = vs[i];
  }
  P.ncols = n_cols;
  hipLaunchKernelGGL(k_decorate, dim3(gd_grid_for(N)), dim3(256), 0, (hipStream_t)stream, points, point_coords, inverse32,
                     pillar_mean, N, P, out);
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// per-column sum and sum of squares of a row-major (R, C) matrix (fp32 or bf16), fp64 result.
// Used for BatchNorm statistics of dense channels-last maps (reference nn.BatchNorm2d in
// spt_backbone_mae.py:40,47) and for column sums of dense gradients.  Each workgroup reduces a
// contiguous chunk of rows with 16-byte loads (fp32 partials), a second kernel combines the
// per-workgroup partials in fp64 in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------
__device__ inline void unpack8(const uint4& u, float (&f)[8]) {
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f[2 * k] = __uint_as_float(w[k] << 16);
    f[2 * k + 1] = __uint_as_float(w[k] & 0xFFFF0000u);
  }
}

template <bool BF16>
__global__ __launch_bounds__(256) void k_colstats_partial(const uint4* __restrict__ x, long long R, int C,
                                                          float* __restrict__ part /* (grid, 2, C) */) {
  constexpr int EPV = BF16 ? 8 : 4;  // elements per 16-byte vector
  const int vec_per_row = C / EPV;
  const int rows_per_iter = 256 / vec_per_row;       // threads beyond rows_per_iter * vec_per_row idle
  const int vr = threadIdx.x / vec_per_row;          // row within the iteration
  const int vc = threadIdx.x % vec_per_row;          // vector column
  const bool live = vr < rows_per_iter;
  const long long chunk = (R + gridDim.x - 1) / gridDim.x;
  const long long r0 = blockIdx.x * chunk;
  const long long r1 = r0 + chunk < R ? r0 + chunk : R;
  float s[EPV], q[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) s[e] = q[e] = 0.f;
  for (long long r = r0 + vr; live && r < r1; r += rows_per_iter) {
    const uint4 u = x[r * vec_per_row + vc];
    float f[8];
    if (BF16) {
      unpack8(u, f);
    } else {
      f[0] = __uint_as_float(u.x);
      f[1] = __uint_as_float(u.y);
      f[2] = __uint_as_float(u.z);
      f[3] = __uint_as_float(u.w);
    }
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      s[e] += f[e];
      q[e] = fmaf(f[e], f[e], q[e]);
    }
  }
  extern __shared__ float sh[];  // (rows_per_iter, 2, C)
  if (live) {
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      sh[(vr * 2 + 0) * C + vc * EPV + e] = s[e];
      sh[(vr * 2 + 1) * C + vc * EPV + e] = q[e];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += 256) {
    float acc = 0.f;
    for (int rr = 0; rr < rows_per_iter; ++rr) acc += sh[rr * 2 * C + c];
    part[(long long)blockIdx.x * 2 * C + c] = acc;
  }
}

__global__ __launch_bounds__(256) void k_colstats_final(const float* __restrict__ part, int nblk, int C2,
                                                        double* __restrict__ out) {
  __shared__ double sh[16][17];
  const int cl = threadIdx.x & 15, ps = threadIdx.x >> 4;   // 16 columns x 16 partial slices per workgroup
  const int c = blockIdx.x * 16 + cl;
  double acc = 0.0;
  if (c < C2) {
#pragma unroll 8
    for (int b = ps; b < nblk; b += 16) acc += (double)part[(long long)b * C2 + c];
  }
  sh[ps][cl] = acc;
  __syncthreads();
  if (ps == 0 && c < C2) {
    double s = 0.0;
    for (int k = 0; k < 16; ++k) s += sh[k][cl];
    out[c] = s;
  }
}

extern "C" size_t gdmae_colstats_workspace_bytes(int C) { return (size_t)1024 * 2 * C * sizeof(float); }

// out: double[2*C] = {sum[0..C), sumsq[0..C)};  is_bf16: 0 = fp32 rows, 1 = bf16 rows
extern "C" int gdmae_colstats(const void* x, long long R, int C, int is_bf16, double* out, void* workspace, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int epv = is_bf16 ? 8 : 4;
  GD_REQUIRE(C % epv == 0 && (C / epv) <= 256, "C must be a multiple of 8 (bf16) / 4 (fp32) and <= 2048 / 1024");
  int nblk = (int)(R / 64 > 512 ? 512 : (R / 64 > 0 ? R / 64 : 1));   // few partials: the fp64 combine stays short
  if (R >= (1ll << 20)) nblk = 1024;                                   // dense maps (1.7 M rows): 4 workgroups per CU
  const int rows_per_iter = 256 / (C / epv);
  const size_t lds = (size_t)rows_per_iter * 2 * C * sizeof(float);
  GD_REQUIRE(lds <= 64 * 1024, "colstats LDS");
  if (is_bf16)
    hipLaunchKernelGGL((k_colstats_partial<true>), dim3(nblk), dim3(256), lds, st, (const uint4*)x, R, C, (float*)workspace);
  else
    hipLaunchKernelGGL((k_colstats_partial<false>), dim3(nblk), dim3(256), lds, st, (const uint4*)x, R, C, (float*)workspace);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_colstats_final, dim3(gd_div_up(2 * C, 16)), dim3(256), 0, st, (const float*)workspace, nblk, 2 * C, out);
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// BatchNorm(train) bookkeeping as ONE launch each (instead of ~25 C-sized torch vector ops per BatchNorm):
//   gdmae_bn_fold       : column statistics of x -> mean, rstd, folded affine a = gamma*rstd, b = beta - a*mean,
//                         and the running_mean / running_var / num_batches_tracked update of nn.BatchNorm
//   gdmae_bn_bwd_coeffs : column sums of the row-kernel backward -> dgamma, dbeta and the per-channel c0, c1 of
//                         dx = a*dh + c0 + c1*x   (chain rule through mean and variance)
// `count` is the number of samples the statistics are over (rows of x, or ALL dense sites when x holds only the
// non-zero rows of an implicit dense map).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bn_fold_final(const float* __restrict__ part, int nblk, int C, double count,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       double eps, double momentum, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, long long* __restrict__ num_batches,
                                                       double* __restrict__ stats, float* __restrict__ ab,
                                                       float* __restrict__ mv) {
  __shared__ double sh[2][16][17];
  const int cl = threadIdx.x & 15, ps = threadIdx.x >> 4;   // 16 columns x 16 partial slices per workgroup
  const int c = blockIdx.x * 16 + cl;
  // eight partial rows (sixteen loads) in flight per thread, requested unconditionally (clamped row, masked value): the launch is a
  // chain of dependent round trips, not bytes (fixed order: the result is bit-repeatable)
  double a1 = 0.0, a2 = 0.0;
  {
    const int cc = c < C ? c : C - 1;
    const float* p = part + cc;
    for (int b0 = ps; b0 < nblk; b0 += 128) {
      float v1[8], v2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int b = b0 + 16 * u < nblk ? b0 + 16 * u : nblk - 1;
        v1[u] = p[(long long)b * 2 * C];
        v2[u] = p[(long long)b * 2 * C + C];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bool ok = b0 + 16 * u < nblk;
        a1 += ok ? (double)v1[u] : 0.0;
        a2 += ok ? (double)v2[u] : 0.0;
      }
    }
  }
  sh[0][ps][cl] = a1;
  sh[1][ps][cl] = a2;
  __syncthreads();
  if (ps == 0 && c < C) {
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < 16; ++k) { s1 += sh[0][k][cl]; s2 += sh[1][k][cl]; }
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double r = 1.0 / sqrt(var + eps);
    const double a = (double)gamma[c] * r;
    stats[c] = mean;
    stats[C + c] = r;
    ab[c] = (float)a;
    ab[C + c] = (float)((double)beta[c] - a * mean);
    mv[c] = (float)mean;
    mv[C + c] = (float)var;
    if (running_mean) {
      const float m = (float)momentum;
      const float unb = (float)(var * (count / (count > 1.0 ? count - 1.0 : 1.0)));
      running_mean[c] = running_mean[c] * (1.f - m) + m * (float)mean;
      running_var[c] = running_var[c] * (1.f - m) + m * unb;
      if (c == 0 && num_batches) *num_batches += 1;
    }
  }
}

extern "C" int gdmae_bn_fold(const void* x, long long R, int C, int is_bf16, double count, const float* gamma,
                             const float* beta, double eps, double momentum, float* running_mean, float* running_var,
                             long long* num_batches, double* stats, float* ab, float* mv, void* workspace, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  const int epv = is_bf16 ? 8 : 4;
  GD_REQUIRE(C % epv == 0 && (C / epv) <= 256, "C must be a multiple of 8 (bf16) / 4 (fp32) and <= 2048 / 1024");
  int nblk = (int)(R / 64 > 512 ? 512 : (R / 64 > 0 ? R / 64 : 1));
  if (R >= (1ll << 20)) nblk = 1024;
  const int rows_per_iter = 256 / (C / epv);
  const size_t lds = (size_t)rows_per_iter * 2 * C * sizeof(float);
  GD_REQUIRE(lds <= 64 * 1024, "colstats LDS");
  if (is_bf16)
    hipLaunchKernelGGL((k_colstats_partial<true>), dim3(nblk), dim3(256), lds, st, (const uint4*)x, R, C, (float*)workspace);
  else
    hipLaunchKernelGGL((k_colstats_partial<false>), dim3(nblk), dim3(256), lds, st, (const uint4*)x, R, C, (float*)workspace);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_bn_fold_final, dim3(gd_div_up(C, 16)), dim3(256), 0, st, (const float*)workspace, nblk, C, count, gamma,
                     beta, eps, momentum, running_mean, running_var, num_batches, stats, ab, mv);
  GD_LAUNCH_CHECK();
  return 0;
}

// the second halves of gdmae_bn_fold / gdmae_colstats for kernels that produce their own per-workgroup partials
// (part[(nblk, 2, C)] resp. part[(nblk, C2)], fp32), e.g. vfe_fused.hip
int gd_bn_fold_from_partials(hipStream_t st, const float* part, int nblk, int C, double count, const float* gamma,
                             const float* beta, double eps, double momentum, float* running_mean, float* running_var,
                             long long* num_batches, double* stats, float* ab, float* mv) {
  hipLaunchKernelGGL(k_bn_fold_final, dim3(gd_div_up(C, 16)), dim3(256), 0, st, part, nblk, C, count, gamma, beta, eps, momentum,
                     running_mean, running_var, num_batches, stats, ab, mv);
  GD_LAUNCH_CHECK();
  return 0;
}
// C ABI of the same: BatchNorm fold from partial rows part (nblk, 2, C) fp32 = per-row-block {column sums, column sums of squares} that a
// producer's epilogue left (csrc/conv_dense.hip gdmae_conv3x3_dense_stats) - gdmae_bn_fold without its pass over x
extern "C" int gdmae_bn_fold_partials(const float* part, int nblk, int C, double count, const float* gamma, const float* beta, double eps,
                                      double momentum, float* running_mean, float* running_var, long long* num_batches, double* stats,
                                      float* ab, float* mv, void* stream) {
  GD_REQUIRE(part != nullptr && nblk >= 1 && C >= 1, "bn_fold_partials: partial rows");
  return gd_bn_fold_from_partials((hipStream_t)stream, part, nblk, C, count, gamma, beta, eps, momentum, running_mean, running_var,
                                  num_batches, stats, ab, mv);
}
int gd_partials_to_f64(hipStream_t st, const float* part, int nblk, int C2, double* out) {
  hipLaunchKernelGGL(k_colstats_final, dim3(gd_div_up(C2, 16)), dim3(256), 0, st, part, nblk, C2, out);
  GD_LAUNCH_CHECK();
  return 0;
}

// st: double[n_st * C] column sums {dh, dh*x, (g)};  tot (optional, needs n_st == 3): column sums of the incoming
// gradient over ALL dense sites - every site that is not a row of x holds the constant relu(b), so the background's
// share of dbeta is (tot - sum g) * [b > 0].
__global__ __launch_bounds__(256) void k_bn_bwd_coeffs(const double* __restrict__ st, int n_st, const double* __restrict__ stats,
                                                       const float* __restrict__ ab, const float* __restrict__ gamma, int C,
                                                       double count, const double* __restrict__ tot,
                                                       float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate,
                                                       float* __restrict__ c01) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mean = stats[c], r = stats[C + c];
  const double a = (double)ab[c], b = (double)ab[C + c];
  double db = st[c];
  if (tot && n_st >= 3 && b > 0.0) db += tot[c] - st[2 * C + c];
  const double da = st[C + c] - db * mean;                 // total derivative w.r.t. a (b = beta - a * mean)
  const double dv = -0.5 * (da * (double)gamma[c]) * r * r * r;   // a = gamma * rsqrt(var + eps)
  const double dmu = -db * a - 2.0 * mean * dv;            // var = s2 / count - mean^2
  c01[c] = (float)(dmu / count);
  c01[C + c] = (float)(2.0 * dv / count);
  const float dg = (float)(da * r), dbf = (float)db;
  if (accumulate) { dgamma[c] += dg; dbeta[c] += dbf; }
  else { dgamma[c] = dg; dbeta[c] = dbf; }
}

// the same from fp32 partial rows part (nblk, n_st, C) (gdmae_rows_bwd_stats with out == NULL): the fp64 column sums are formed
// here - one workgroup per 8 channels, 32 slices of the rows per channel meeting in LDS in a fixed order
__global__ __launch_bounds__(256) void k_bn_bwd_coeffs_rows(const float* __restrict__ part, int nblk, int n_st,
                                                            const double* __restrict__ stats, const float* __restrict__ ab,
                                                            const float* __restrict__ gamma, int C, double count,
                                                            const double* __restrict__ tot, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int accumulate, float* __restrict__ c01) {
  __shared__ double sh[3][32][9];
  const int cl = threadIdx.x & 7, ps = threadIdx.x >> 3;        // 8 channels x 32 row slices per workgroup
  const int c = blockIdx.x * 8 + cl;
  const long long stride = (long long)n_st * C;
  {
    // the three statistics of eight partial rows (24 loads) in flight per thread, unconditional (clamped row / statistic, masked
    // value): twelve dependent round trips per launch became two to four
    const int cc = c < C ? c : C - 1;
    double acc[3] = {0.0, 0.0, 0.0};
    for (int b0 = ps; b0 < nblk; b0 += 256) {
      float x[3][8];
#pragma unroll
      for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int b = b0 + 32 * u < nblk ? b0 + 32 * u : nblk - 1;
          x[v][u] = part[(long long)b * stride + (long long)(v < n_st ? v : 0) * C + cc];
        }
#pragma unroll
      for (int v = 0; v < 3; ++v)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[v] += b0 + 32 * u < nblk ? (double)x[v][u] : 0.0;
    }
#pragma unroll
    for (int v = 0; v < 3; ++v) sh[v][ps][cl] = acc[v];
  }
  __syncthreads();
  if (ps != 0 || c >= C) return;
  double s[3] = {0.0, 0.0, 0.0};
  for (int v = 0; v < n_st; ++v)
    for (int k = 0; k < 32; ++k) s[v] += sh[v][k][cl];
  const double mean = stats[c], r = stats[C + c];
  const double a = (double)ab[c], bb = (double)ab[C + c];
  double db = s[0];
  if (tot && n_st >= 3 && bb > 0.0) db += tot[c] - s[2];
  const double da = s[1] - db * mean;
  const double dv = -0.5 * (da * (double)gamma[c]) * r * r * r;
  const double dmu = -db * a - 2.0 * mean * dv;
  c01[c] = (float)(dmu / count);
  c01[C + c] = (float)(2.0 * dv / count);
  const float dg = (float)(da * r), dbf = (float)db;
  if (accumulate) { dgamma[c] += dg; dbeta[c] += dbf; }
  else { dgamma[c] = dg; dbeta[c] = dbf; }
}
extern "C" int gdmae_bn_bwd_coeffs_rows(const float* part, int nblk, int n_st, const double* stats, const float* ab, const float* gamma,
                                        int C, double count, const double* tot, float* dgamma, float* dbeta, int accumulate, float* c01,
                                        void* stream) {
  GD_REQUIRE((n_st == 2 || n_st == 3) && nblk >= 1, "n_st must be 2 or 3");
  hipLaunchKernelGGL(k_bn_bwd_coeffs_rows, dim3(gd_div_up(C, 8)), dim3(256), 0, (hipStream_t)stream, part, nblk, n_st, stats, ab, gamma, C,
                     count, tot, dgamma, dbeta, accumulate, c01);
  GD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gdmae_bn_bwd_coeffs(const double* st, int n_st, const double* stats, const float* ab, const float* gamma, int C,
                                   double count, const double* tot, float* dgamma, float* dbeta, int accumulate, float* c01,
                                   void* stream) {
  GD_REQUIRE(n_st == 2 || n_st == 3, "n_st must be 2 or 3");
  hipLaunchKernelGGL(k_bn_bwd_coeffs, dim3(gd_div_up(C, 256)), dim3(256), 0, (hipStream_t)stream, st, n_st, stats, ab, gamma, C,
                     count, tot, dgamma, dbeta, accumulate, c01);
  GD_LAUNCH_CHECK();
  return 0;
}
