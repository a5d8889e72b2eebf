// Token-side kernels of the sparse-aware generative decoder (gdmae_hip/decoder.py).
//
// Replaces, per decoder source stage, the dense ConvTranspose2d -> BatchNorm2d -> ReLU -> cat chain of the
// reference (pcdet/models/backbones_3d/spt_backbone_mae.py:30-44,125-132) on the rows that are not the
// per-channel background: P (n, C) are the deconvolution outputs of the active sites (one row per
// (token, dy, dx)), the BatchNorm batch statistics over all dense sites reduce to column sums of P, and
//   forward : Z[site[r], col0:col0+C] = relu(a * P[r] + b)                      (k_rows_affine_relu_scatter)
//   backward: dh = dZ[site[r], slice] * (a P[r] + b > 0);  column sums of dh, dh*P, dZ rows (k_rows_bwd_stats)
//             dP[r] = a * dh + c0 + c1 * P[r]                                   (k_rows_bwd)
// with per-channel a, b, c0, c1 computed by the caller from the column sums (BatchNorm algebra).
// All three are single-pass, HBM-bound row kernels: 16-byte loads, lanes on channels, rows strided over the
// workgroup; Z / dZ may be bf16 (throughput mode) or fp32 rows of `zrow` elements, P / dP fp32 or bf16.
#include "common.h"

__device__ inline float dec_bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ inline unsigned short dec_f2bf(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7F800000u) == 0x7F800000u) return (unsigned short)(u >> 16);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
template <bool BF>
__device__ inline float dec_ld(const void* p, long long i) {
  return BF ? dec_bf2f(((const unsigned short*)p)[i]) : ((const float*)p)[i];
}
template <bool BF>
__device__ inline void dec_st(void* p, long long i, float v) {
  if (BF) ((unsigned short*)p)[i] = dec_f2bf(v);
  else ((float*)p)[i] = v;
}

// C = 128 channels: a wavefront covers a row with 2 channels per lane
template <bool PBF, bool ZBF>
__global__ __launch_bounds__(256) void k_rows_affine_relu_scatter(const void* __restrict__ P, const int* __restrict__ site,
                                                                  long long n, int C, const float* __restrict__ a,
                                                                  const float* __restrict__ b, void* __restrict__ Z,
                                                                  int zrow, int col0) {
  const long long total = n * (long long)C;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / C;
    const int c = (int)(e % C);
    const float h = fmaf(a[c], dec_ld<PBF>(P, e), b[c]);
    dec_st<ZBF>(Z, (long long)site[r] * zrow + col0 + c, h > 0.f ? h : 0.f);
  }
}

// part: (grid, 3, C) fp32 partial column sums of {dh, dh*P, g}
template <bool PBF, bool ZBF>
__global__ __launch_bounds__(256) void k_rows_bwd_stats(const void* __restrict__ P, const int* __restrict__ site, long long n,
                                                        int C, const float* __restrict__ a, const float* __restrict__ b,
                                                        const void* __restrict__ dZ, int zrow, int col0,
                                                        float* __restrict__ part) {
  extern __shared__ float sh[];  // (rows_per_iter, 3, C)
  const int rows_per_iter = 256 / C > 0 ? 256 / C : 1;   // C <= 256
  const int tr = threadIdx.x / C, c = threadIdx.x % C;
  const bool live = tr < rows_per_iter;
  const long long chunk = (n + gridDim.x - 1) / gridDim.x;
  const long long r0 = blockIdx.x * chunk, r1 = r0 + chunk < n ? r0 + chunk : n;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (live) {
    const float ac = a[c], bc = b[c];
    for (long long r = r0 + tr; r < r1; r += rows_per_iter) {
      const float p = dec_ld<PBF>(P, r * C + c);
      const float g = dec_ld<ZBF>(dZ, (long long)site[r] * zrow + col0 + c);
      const float dh = fmaf(ac, p, bc) > 0.f ? g : 0.f;
      s0 += dh;
      s1 = fmaf(dh, p, s1);
      s2 += g;
    }
    sh[(tr * 3 + 0) * C + c] = s0;
    sh[(tr * 3 + 1) * C + c] = s1;
    sh[(tr * 3 + 2) * C + c] = s2;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < 3 * C; q += 256) {
    float acc = 0.f;
    for (int rr = 0; rr < rows_per_iter; ++rr) acc += sh[rr * 3 * C + q];
    part[(long long)blockIdx.x * 3 * C + q] = acc;
  }
}

__global__ __launch_bounds__(256) void k_partials_to_f64(const float* __restrict__ part, int nblk, int C3,
                                                         double* __restrict__ out) {
  __shared__ double sh[16][17];
  const int cl = threadIdx.x & 15, ps = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double acc = 0.0;
  if (c < C3)
    for (int b = ps; b < nblk; b += 16) acc += (double)part[(long long)b * C3 + c];
  sh[ps][cl] = acc;
  __syncthreads();
  if (ps == 0 && c < C3) {
    double s = 0.0;
    for (int k = 0; k < 16; ++k) s += sh[k][cl];
    out[c] = s;
  }
}

template <bool PBF, bool ZBF, bool OBF>
__global__ __launch_bounds__(256) void k_rows_bwd(const void* __restrict__ P, const int* __restrict__ site, long long n, int C,
                                                  const float* __restrict__ a, const float* __restrict__ b,
                                                  const float* __restrict__ c0, const float* __restrict__ c1,
                                                  const void* __restrict__ dZ, int zrow, int col0, void* __restrict__ dP) {
  const long long total = n * (long long)C;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long r = e / C;
    const int c = (int)(e % C);
    const float p = dec_ld<PBF>(P, e);
    const float g = dec_ld<ZBF>(dZ, (long long)site[r] * zrow + col0 + c);
    const float dh = fmaf(a[c], p, b[c]) > 0.f ? g : 0.f;
    dec_st<OBF>(dP, e, fmaf(a[c], dh, fmaf(c1[c], p, c0[c])));
  }
}

static inline int dec_grid(long long total) {
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

extern "C" int gdmae_rows_affine_relu_scatter(const void* P, int p_bf16, const int* site, long long n, int C, const float* a,
                                              const float* b, void* Z, int z_bf16, int z_row_elems, int col0, void* stream) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(dec_grid(n * C)), block(256);
#define GD_LAUNCH(PB, ZB) hipLaunchKernelGGL((k_rows_affine_relu_scatter<PB, ZB>), grid, block, 0, st, P, site, n, C, a, b, Z, z_row_elems, col0)
  if (p_bf16) { if (z_bf16) GD_LAUNCH(true, true); else GD_LAUNCH(true, false); }
  else { if (z_bf16) GD_LAUNCH(false, true); else GD_LAUNCH(false, false); }
#undef GD_LAUNCH
  GD_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t gdmae_rows_bwd_stats_workspace_bytes(int C) { return (size_t)512 * 3 * C * sizeof(float); }

// out: double[3*C] = column sums of {dh, dh * P, dZ rows}
extern "C" int gdmae_rows_bwd_stats(const void* P, int p_bf16, const int* site, long long n, int C, const float* a,
                                    const float* b, const void* dZ, int z_bf16, int z_row_elems, int col0, double* out,
                                    void* workspace, void* stream) {
  GD_REQUIRE(C >= 1 && C <= 256, "C <= 256");
  hipStream_t st = (hipStream_t)stream;
  int nblk = (int)(n / 32 > 512 ? 512 : (n / 32 > 0 ? n / 32 : 1));
  const int rpi = 256 / C > 0 ? 256 / C : 1;
  const size_t lds = (size_t)rpi * 3 * C * sizeof(float);
  float* part = (float*)workspace;
  const dim3 grid(nblk), block(256);
#define GD_LAUNCH(PB, ZB) hipLaunchKernelGGL((k_rows_bwd_stats<PB, ZB>), grid, block, lds, st, P, site, n, C, a, b, dZ, z_row_elems, col0, part)
  if (p_bf16) { if (z_bf16) GD_LAUNCH(true, true); else GD_LAUNCH(true, false); }
  else { if (z_bf16) GD_LAUNCH(false, true); else GD_LAUNCH(false, false); }
#undef GD_LAUNCH
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_partials_to_f64, dim3(gd_div_up(3 * C, 16)), dim3(256), 0, st, part, nblk, 3 * C, out);
  GD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gdmae_rows_bwd(const void* P, int p_bf16, const int* site, long long n, int C, const float* a, const float* b,
                              const float* c0, const float* c1, const void* dZ, int z_bf16, int z_row_elems, int col0,
                              void* dP, int dp_bf16, void* stream) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(dec_grid(n * C)), block(256);
#define GD_LAUNCH(PB, ZB, OB) hipLaunchKernelGGL((k_rows_bwd<PB, ZB, OB>), grid, block, 0, st, P, site, n, C, a, b, c0, c1, dZ, z_row_elems, col0, dP)
  if (p_bf16) {
    if (z_bf16) { if (dp_bf16) GD_LAUNCH(true, true, true); else GD_LAUNCH(true, true, false); }
    else { if (dp_bf16) GD_LAUNCH(true, false, true); else GD_LAUNCH(true, false, false); }
  } else {
    if (z_bf16) { if (dp_bf16) GD_LAUNCH(false, true, true); else GD_LAUNCH(false, true, false); }
    else { if (dp_bf16) GD_LAUNCH(false, false, true); else GD_LAUNCH(false, false, false); }
  }
#undef GD_LAUNCH
  GD_LAUNCH_CHECK();
  return 0;
}
