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
__global__ __launch_bounds__(256) void k_gather_rows(const uint4* __restrict__ src, const int* __restrict__ idx,
                                                     long long n_slots, int row_vec, uint4* __restrict__ out) {
  const long long total = n_slots * row_vec;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long s = i / row_vec;
    const int v = (int)(i % row_vec);
    const int j = idx[s];
    uint4 val = make_uint4(0, 0, 0, 0);
    if (j >= 0) val = src[(long long)j * row_vec + v];
    out[i] = val;
  }
}

// dst[idx[r], :] = src[r, :]   (idx unique; rows with idx < 0 are skipped)
__global__ __launch_bounds__(256) void k_scatter_rows(const uint4* __restrict__ src, const int* __restrict__ idx,
                                                      long long n_rows, int row_vec, uint4* __restrict__ dst) {
  const long long total = n_rows * row_vec;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long s = i / row_vec;
    const int v = (int)(i % row_vec);
    const int j = idx[s];
    if (j >= 0) dst[(long long)j * row_vec + v] = src[i];
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
                     (const uint4*)src, idx, n_slots, rv, (uint4*)out);
  GD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gdmae_scatter_rows(const void* src, const int* idx, long long n_rows, int row_bytes, void* dst,
                                  void* stream) {
  GD_REQUIRE(row_bytes % 16 == 0, "row_bytes must be a multiple of 16");
  if (n_rows <= 0) return 0;
  const int rv = row_bytes / 16;
  hipLaunchKernelGGL(k_scatter_rows, dim3(gd_grid_for(n_rows * rv)), dim3(256), 0, (hipStream_t)stream,
                     (const uint4*)src, idx, n_rows, rv, (uint4*)dst);
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// segmented max over the pillar CSR: out[p, c] = max_{i in pillar p} x[i, c], arg[p, c] = that i
// (lowest i on ties).  One wavefront per pillar, lanes on channels (float2 per lane for C = 128).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_segment_max(const float* __restrict__ x, const int* __restrict__ pt_off,
                                                     const int* __restrict__ csr, int M, int C, float* __restrict__ out,
                                                     int* __restrict__ arg) {
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  for (int p = blockIdx.x * 4 + wib; p < M; p += gridDim.x * 4) {
    const int off = pt_off[p];
    const int cnt = pt_off[p + 1] - off;
    for (int c = lane; c < C; c += GD_WAVE) {
      float best = -INFINITY;
      int bi = -1;
      for (int j = 0; j < cnt; ++j) {
        const int i = csr[off + j];
        const float v = x[(long long)i * C + c];
        if (v > best || bi < 0) {
          best = v;
          bi = i;
        }
      }
      out[(long long)p * C + c] = best;
      arg[(long long)p * C + c] = bi;
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
  hipLaunchKernelGGL(k_segment_max, dim3(gd_grid_for(M, 4, 8192)), dim3(256), 0, (hipStream_t)stream, x, pt_off, csr, M, C,
                     out, arg);
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
