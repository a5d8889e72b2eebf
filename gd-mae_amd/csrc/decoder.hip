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
#include "conv_tiles.h"

__device__ inline float dec_bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ inline unsigned short dec_f2bf(float f) { return gd_to_bf16(f); }
template <bool BF>
__device__ inline float dec_ld(const void* p, long long i) {
  return BF ? dec_bf2f(((const unsigned short*)p)[i]) : ((const float*)p)[i];
}
template <bool BF>
__device__ inline void dec_st(void* p, long long i, float v) {
  if (BF) ((unsigned short*)p)[i] = dec_f2bf(v);
  else ((float*)p)[i] = v;
}

// 8 consecutive elements (index i is a multiple of 8): one 16-byte (bf16) or two 16-byte (fp32) accesses
template <bool BF>
__device__ inline void dec_ld8(const void* p, long long i, float (&v)[8]) {
  if (BF) {
    const uint4 q = *reinterpret_cast<const uint4*>((const unsigned short*)p + i);
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[2 * j] = __uint_as_float(w[j] << 16);
      v[2 * j + 1] = __uint_as_float(w[j] & 0xFFFF0000u);
    }
  } else {
    const float4 a = *reinterpret_cast<const float4*>((const float*)p + i), b = *reinterpret_cast<const float4*>((const float*)p + i + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}
template <bool BF>
__device__ inline void dec_st8(void* p, long long i, const float (&v)[8]) {
  if (BF) {
    uint4 q;
    q.x = dec_f2bf(v[0]) | ((unsigned)dec_f2bf(v[1]) << 16);
    q.y = dec_f2bf(v[2]) | ((unsigned)dec_f2bf(v[3]) << 16);
    q.z = dec_f2bf(v[4]) | ((unsigned)dec_f2bf(v[5]) << 16);
    q.w = dec_f2bf(v[6]) | ((unsigned)dec_f2bf(v[7]) << 16);
    *reinterpret_cast<uint4*>((unsigned short*)p + i) = q;
  } else {
    *reinterpret_cast<float4*>((float*)p + i) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>((float*)p + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// ---- 8-channels-per-thread variants of the three row kernels (C % 8 == 0, zrow % 8 == 0, col0 % 8 == 0) ----
template <bool PBF, bool ZBF>
__global__ __launch_bounds__(256) void k_rows_affine_relu_scatter_v8(const void* __restrict__ P, const int* __restrict__ site,
                                                                     long long n, int C, const float* __restrict__ a,
                                                                     const float* __restrict__ b, void* __restrict__ Z,
                                                                     int zrow, int col0, const void* __restrict__ sub,
                                                                     const void* __restrict__ add) {
  const int cv = C >> 3;
  const long long total = n * cv;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long r = t / cv;
    const int c = (int)(t % cv) << 3;
    float p[8], av[8], bv[8], o[8];
    dec_ld8<PBF>(P, r * C + c, p);
    dec_ld8<false>(a, c, av);
    dec_ld8<false>(b, c, bv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float h = fmaf(av[j], p[j], bv[j]);
      o[j] = h > 0.f ? h : 0.f;
    }
    if (sub) {      // Z = round(round(relu) - sub): the rows minus a per-channel constant, rounded like the two-step sequence
      float sv[8];
      dec_ld8<ZBF>(sub, c, sv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (ZBF ? dec_bf2f(dec_f2bf(o[j])) : o[j]) - sv[j];
    }
    if (add) {      // Z = round(round(relu) + add[r]): identity shortcut of a dense block (rows of Z's dtype), rounded like y + x
      float rv[8];
      dec_ld8<ZBF>(add, r * C + c, rv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (ZBF ? dec_bf2f(dec_f2bf(o[j])) : o[j]) + rv[j];
    }
    dec_st8<ZBF>(Z, (site ? (long long)site[r] : r) * zrow + col0 + c, o);
  }
}

template <bool PBF, bool ZBF>
__global__ __launch_bounds__(256) void k_rows_bwd_stats_v8(const void* __restrict__ P, const int* __restrict__ site, long long n,
                                                           int C, const float* __restrict__ a, const float* __restrict__ b,
                                                           const void* __restrict__ dZ, int zrow, int col0,
                                                           float* __restrict__ part) {
  extern __shared__ float sh[];  // (rows_per_iter, 3, C)
  const int cv = C >> 3;                       // C <= 256 -> cv <= 32
  const int rows_per_iter = 256 / cv;
  const int tr = threadIdx.x / cv, c = (threadIdx.x % cv) << 3;
  const long long chunk = (n + gridDim.x - 1) / gridDim.x;
  const long long r0 = blockIdx.x * chunk, r1 = r0 + chunk < n ? r0 + chunk : n;
  float s0[8], s1[8], s2[8], av[8], bv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s0[j] = s1[j] = s2[j] = 0.f;
  dec_ld8<false>(a, c, av);
  dec_ld8<false>(b, c, bv);
  for (long long r = r0 + tr; r < r1; r += rows_per_iter) {
    float p[8], g[8];
    dec_ld8<PBF>(P, r * C + c, p);
    dec_ld8<ZBF>(dZ, (site ? (long long)site[r] : r) * zrow + col0 + c, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float dh = fmaf(av[j], p[j], bv[j]) > 0.f ? g[j] : 0.f;
      s0[j] += dh;
      s1[j] = fmaf(dh, p[j], s1[j]);
      s2[j] += g[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sh[(tr * 3 + 0) * C + c + j] = s0[j];
    sh[(tr * 3 + 1) * C + c + j] = s1[j];
    sh[(tr * 3 + 2) * C + c + j] = s2[j];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < 3 * C; q += 256) {
    float acc = 0.f;
    for (int rr = 0; rr < rows_per_iter; ++rr) acc += sh[rr * 3 * C + q];
    part[(long long)blockIdx.x * 3 * C + q] = acc;
  }
}

template <bool PBF, bool ZBF, bool OBF>
__global__ __launch_bounds__(256) void k_rows_bwd_v8(const void* __restrict__ P, const int* __restrict__ site, long long n, int C,
                                                     const float* __restrict__ a, const float* __restrict__ b,
                                                     const float* __restrict__ c0, const float* __restrict__ c1,
                                                     const void* __restrict__ dZ, int zrow, int col0, void* __restrict__ dP) {
  const int cv = C >> 3;
  const long long total = n * cv;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long r = t / cv;
    const int c = (int)(t % cv) << 3;
    float p[8], g[8], av[8], bv[8], k0[8], k1[8], o[8];
    dec_ld8<PBF>(P, r * C + c, p);
    dec_ld8<ZBF>(dZ, (site ? (long long)site[r] : r) * zrow + col0 + c, g);
    dec_ld8<false>(a, c, av);
    dec_ld8<false>(b, c, bv);
    dec_ld8<false>(c0, c, k0);
    dec_ld8<false>(c1, c, k1);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float dh = fmaf(av[j], p[j], bv[j]) > 0.f ? g[j] : 0.f;
      o[j] = fmaf(av[j], dh, fmaf(k1[j], p[j], k0[j]));
    }
    dec_st8<OBF>(dP, r * C + c, o);
  }
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
    dec_st<ZBF>(Z, (site ? (long long)site[r] : r) * zrow + col0 + c, h > 0.f ? h : 0.f);
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
      const float g = dec_ld<ZBF>(dZ, (site ? (long long)site[r] : r) * zrow + col0 + c);
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
    const float g = dec_ld<ZBF>(dZ, (site ? (long long)site[r] : r) * zrow + col0 + c);
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

// Z (R, C) = the row vector v (C elements, C * element size a multiple of 16 bytes) in every row: the background of the
// dense decoder map (1.35 GB in bf16 for 8 frames; torch's expand().contiguous() copy runs at 4.6 TB/s).  Each lane
// keeps ONE 16-byte piece of v and writes it at a fixed stride, so the kernel is nothing but 16-byte stores.
__global__ __launch_bounds__(256) void k_fill_rows(const uint4* __restrict__ v, int vec_per_row, long long total_vecs,
                                                   uint4* __restrict__ Z) {
  // stride = a multiple of vec_per_row, so a thread always lands on the same piece of v
  const long long stride = ((long long)gridDim.x * blockDim.x / vec_per_row) * vec_per_row;
  const long long i0 = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i0 >= stride) return;
  const uint4 q = v[i0 % vec_per_row];
  for (long long i = i0; i < total_vecs; i += stride) Z[i] = q;
}

extern "C" int gdmae_fill_rows(const void* v, long long R, int C, int elem_bytes, void* Z, void* stream) {
  if (R <= 0) return 0;
  GD_REQUIRE(C > 0 && ((long long)C * elem_bytes) % 16 == 0, "fill_rows: row bytes must be a multiple of 16");
  const int vpr = (int)((long long)C * elem_bytes / 16);
  const long long total = R * vpr;
  long long blocks = (total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  if (blocks * 256 < vpr) blocks = (vpr + 255) / 256;
  hipLaunchKernelGGL(k_fill_rows, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)v, vpr, total,
                     (uint4*)Z);
  GD_LAUNCH_CHECK();
  return 0;
}

static int rows_affine_relu_scatter(const void* P, int p_bf16, const int* site, long long n, int C, const float* a, const float* b, void* Z,
                                    int z_bf16, int z_row_elems, int col0, const void* sub, void* stream, const void* add = nullptr);
extern "C" int gdmae_rows_affine_relu_scatter(const void* P, int p_bf16, const int* site, long long n, int C, const float* a,
                                              const float* b, void* Z, int z_bf16, int z_row_elems, int col0, void* stream) {
  return rows_affine_relu_scatter(P, p_bf16, site, n, C, a, b, Z, z_bf16, z_row_elems, col0, nullptr, stream);
}
// Z[site[r] or r, col0 + c] = relu(a P + b) - sub[c]   (sub (C) in Z's dtype; C, the row pitch and col0 multiples of 8)
extern "C" int gdmae_rows_affine_relu_sub(const void* P, int p_bf16, const int* site, long long n, int C, const float* a, const float* b,
                                          const void* sub, void* Z, int z_bf16, int z_row_elems, int col0, void* stream) {
  GD_REQUIRE(sub != nullptr && C % 8 == 0 && z_row_elems % 8 == 0 && col0 % 8 == 0, "rows_affine_relu_sub: 8-channel granularity");
  return rows_affine_relu_scatter(P, p_bf16, site, n, C, a, b, Z, z_bf16, z_row_elems, col0, sub, stream);
}
static int rows_affine_relu_scatter(const void* P, int p_bf16, const int* site, long long n, int C, const float* a, const float* b, void* Z,
                                    int z_bf16, int z_row_elems, int col0, const void* sub, void* stream, const void* add) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(dec_grid(n * C)), block(256);
  const bool v8 = (C % 8 == 0) && (z_row_elems % 8 == 0) && (col0 % 8 == 0);
  const dim3 grid8(dec_grid(n * C / 8));
#define GD_LAUNCH(PB, ZB)                                                                                                         \
  do {                                                                                                                            \
    if (v8) hipLaunchKernelGGL((k_rows_affine_relu_scatter_v8<PB, ZB>), grid8, block, 0, st, P, site, n, C, a, b, Z, z_row_elems, col0, sub, add); \
    else hipLaunchKernelGGL((k_rows_affine_relu_scatter<PB, ZB>), grid, block, 0, st, P, site, n, C, a, b, Z, z_row_elems, col0);  \
  } while (0)
  if (p_bf16) { if (z_bf16) GD_LAUNCH(true, true); else GD_LAUNCH(true, false); }
  else { if (z_bf16) GD_LAUNCH(false, true); else GD_LAUNCH(false, false); }
#undef GD_LAUNCH
  GD_LAUNCH_CHECK();
  return 0;
}
// Z[r] = relu(a P[r] + b) + R[r]  (R, Z (n, C) rows of the same dtype; C % 8 == 0): BatchNorm + ReLU + identity shortcut of a dense
// Conv-BN-ReLU block (sst_bev_backbone.py:36-40), rounded like the two-step sequence
extern "C" int gdmae_rows_affine_relu_add(const void* P, int p_bf16, long long n, int C, const float* a, const float* b, const void* R,
                                          void* Z, int z_bf16, void* stream) {
  GD_REQUIRE(R != nullptr && C % 8 == 0, "rows_affine_relu_add: 8-channel granularity");
  return rows_affine_relu_scatter(P, p_bf16, nullptr, n, C, a, b, Z, z_bf16, C, 0, nullptr, stream, R);
}

extern "C" size_t gdmae_rows_bwd_stats_workspace_bytes(int C) { return (size_t)512 * 3 * C * sizeof(float); }

// number of partial rows gdmae_rows_bwd_stats leaves in its workspace ((rows, 3, C) fp32)
extern "C" int gdmae_rows_bwd_stats_rows(long long n) { return (int)(n / 32 > 512 ? 512 : (n / 32 > 0 ? n / 32 : 1)); }

// out: double[3*C] = column sums of {dh, dh * P, dZ rows}; out == NULL: only the per-workgroup partial rows are produced
// (workspace = (gdmae_rows_bwd_stats_rows(n), 3, C) fp32) for gdmae_bn_bwd_coeffs_rows to reduce
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
  const bool v8 = (C % 8 == 0) && (256 % (C / 8) == 0) && (z_row_elems % 8 == 0) && (col0 % 8 == 0);
  const size_t lds8 = v8 ? (size_t)(256 / (C / 8)) * 3 * C * sizeof(float) : 0;
#define GD_LAUNCH(PB, ZB)                                                                                                                 \
  do {                                                                                                                                    \
    if (v8) hipLaunchKernelGGL((k_rows_bwd_stats_v8<PB, ZB>), grid, block, lds8, st, P, site, n, C, a, b, dZ, z_row_elems, col0, part);    \
    else hipLaunchKernelGGL((k_rows_bwd_stats<PB, ZB>), grid, block, lds, st, P, site, n, C, a, b, dZ, z_row_elems, col0, part);           \
  } while (0)
  if (p_bf16) { if (z_bf16) GD_LAUNCH(true, true); else GD_LAUNCH(true, false); }
  else { if (z_bf16) GD_LAUNCH(false, true); else GD_LAUNCH(false, false); }
#undef GD_LAUNCH
  GD_LAUNCH_CHECK();
  if (out) {
    hipLaunchKernelGGL(k_partials_to_f64, dim3(gd_div_up(3 * C, 16)), dim3(256), 0, st, part, nblk, 3 * C, out);
    GD_LAUNCH_CHECK();
  }
  return 0;
}

extern "C" int gdmae_rows_bwd(const void* P, int p_bf16, const int* site, long long n, int C, const float* a, const float* b,
                              const float* c0, const float* c1, const void* dZ, int z_bf16, int z_row_elems, int col0,
                              void* dP, int dp_bf16, void* stream) {
  if (n <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(dec_grid(n * C)), block(256);
  const bool v8 = (C % 8 == 0) && (z_row_elems % 8 == 0) && (col0 % 8 == 0);
  const dim3 grid8(dec_grid(n * C / 8));
#define GD_LAUNCH(PB, ZB, OB)                                                                                                                 \
  do {                                                                                                                                        \
    if (v8) hipLaunchKernelGGL((k_rows_bwd_v8<PB, ZB, OB>), grid8, block, 0, st, P, site, n, C, a, b, c0, c1, dZ, z_row_elems, col0, dP);      \
    else hipLaunchKernelGGL((k_rows_bwd<PB, ZB, OB>), grid, block, 0, st, P, site, n, C, a, b, c0, c1, dZ, z_row_elems, col0, dP);             \
  } while (0)
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

// ------------------------------------------------------------------------------------------------
// DynVFE tail: BatchNorm1d(train) + ReLU + per-pillar max fused (reference dyn_vfe.py:107-109,
// network_utils.py:7-21).  x (N, C) is the second Linear's output (fp32 or bf16), a/b the BatchNorm affine
// from the column statistics of x:  out[p, c] = max_{i in pillar p} relu(a_c x[i, c] + b_c), arg = that i
// (lowest id on ties, canonical CSR order).  One wavefront per pillar, LPP lanes per point (4 channels per lane).
// ------------------------------------------------------------------------------------------------
template <int LPP, bool XBF>
__global__ __launch_bounds__(256) void k_segmax_affine(const void* __restrict__ x, const int* __restrict__ pt_off,
                                                       const int* __restrict__ csr, int M, const float* __restrict__ a,
                                                       const float* __restrict__ b, float* __restrict__ out,
                                                       int* __restrict__ arg) {
  constexpr int C = LPP * 4;
  constexpr int G = GD_WAVE / LPP;
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  const int grp = lane / LPP, cl = lane % LPP;
  float av[4], bv[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    av[e] = a[cl * 4 + e];
    bv[e] = b[cl * 4 + e];
  }
  for (int p = blockIdx.x * 4 + wib; p < M; p += gridDim.x * 4) {
    const int off = pt_off[p];
    const int cnt = pt_off[p + 1] - off;
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
#pragma unroll 4
    for (int j = grp; j < cnt; j += G) {
      const int i = csr[off + j];
      float vv[4];
      if (XBF) {
        const uint2 u = *reinterpret_cast<const uint2*>((const unsigned short*)x + (long long)i * C + cl * 4);
        vv[0] = __uint_as_float(u.x << 16);
        vv[1] = __uint_as_float(u.x & 0xFFFF0000u);
        vv[2] = __uint_as_float(u.y << 16);
        vv[3] = __uint_as_float(u.y & 0xFFFF0000u);
      } else {
        const float4 v = *reinterpret_cast<const float4*>((const float*)x + (long long)i * C + cl * 4);
        vv[0] = v.x; vv[1] = v.y; vv[2] = v.z; vv[3] = v.w;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float h = fmaxf(fmaf(av[e], vv[e], bv[e]), 0.f);
        if (h > best[e] || bi[e] == 0x7fffffff) {
          best[e] = h;
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

extern "C" int gdmae_segment_max_affine(const void* x, int x_bf16, const int* pillar_pt_off, const int* pillar_pts, int M,
                                        int C, const float* a, const float* b, float* out, int* arg, void* stream) {
  if (M <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  int g = gd_div_up(M, 4);
  if (g > 16384) g = 16384;
  const dim3 grid(g), block(256);
#define GD_SM(L, B) hipLaunchKernelGGL((k_segmax_affine<L, B>), grid, block, 0, st, x, pillar_pt_off, pillar_pts, M, a, b, out, arg)
  if (C == 64) { if (x_bf16) GD_SM(16, true); else GD_SM(16, false); }
  else if (C == 128) { if (x_bf16) GD_SM(32, true); else GD_SM(32, false); }
  else if (C == 256) { if (x_bf16) GD_SM(64, true); else GD_SM(64, false); }
  else GD_REQUIRE(false, "segment_max_affine supports C in {64, 128, 256}");
#undef GD_SM
  GD_LAUNCH_CHECK();
  return 0;
}

// Backward of the fused tail.  dh[i, c] = dout[p, c] if (arg[p, c] == i and out[p, c] > 0) else 0.
// (1) column sums of {dh, dh * x} - only arg-max points contribute, so this walks the (M, C) outputs;
// (2) dx[i, c] = a_c dh[i, c] + c0_c + c1_c x[i, c]  for every point (BatchNorm chain rule), one dense pass.
template <bool XBF>
__global__ __launch_bounds__(256) void k_segmax_bwd_stats(const void* __restrict__ x, const float* __restrict__ out,
                                                          const int* __restrict__ arg, const float* __restrict__ dout,
                                                          long long M, int C, float* __restrict__ part) {
  extern __shared__ float sh[];  // (rows_per_iter, 2, C)
  const int rows_per_iter = 256 / C > 0 ? 256 / C : 1;
  const int tr = threadIdx.x / C, c = threadIdx.x % C;
  const bool live = tr < rows_per_iter;
  const long long chunk = (M + gridDim.x - 1) / gridDim.x;
  const long long r0 = blockIdx.x * chunk, r1 = r0 + chunk < M ? r0 + chunk : M;
  float s0 = 0.f, s1 = 0.f;
  if (live) {
    for (long long p = r0 + tr; p < r1; p += rows_per_iter) {
      const long long q = p * C + c;
      if (out[q] > 0.f) {
        const float g = dout[q];
        const float xv = dec_ld<XBF>(x, (long long)arg[q] * C + c);
        s0 += g;
        s1 = fmaf(g, xv, s1);
      }
    }
    sh[(tr * 2 + 0) * C + c] = s0;
    sh[(tr * 2 + 1) * C + c] = s1;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < 2 * C; q += 256) {
    float acc = 0.f;
    for (int rr = 0; rr < rows_per_iter; ++rr) acc += sh[rr * 2 * C + q];
    part[(long long)blockIdx.x * 2 * C + q] = acc;
  }
}

template <bool XBF, bool OBF>
__global__ __launch_bounds__(256) void k_segmax_bn_bwd(const void* __restrict__ x, const float* __restrict__ out,
                                                       const int* __restrict__ arg, const float* __restrict__ dout,
                                                       const int* __restrict__ inv, long long N, int C,
                                                       const float* __restrict__ a, const float* __restrict__ c0,
                                                       const float* __restrict__ c1, void* __restrict__ dx) {
  const long long total = N * C;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long i = e / C;
    const int c = (int)(e % C);
    const long long q = (long long)inv[i] * C + c;
    const float dh = (arg[q] == (int)i && out[q] > 0.f) ? dout[q] : 0.f;
    dec_st<OBF>(dx, e, fmaf(a[c], dh, fmaf(c1[c], dec_ld<XBF>(x, e), c0[c])));
  }
}

// 8 channels per thread (C % 8 == 0): the pillar rows (out / arg / dout) are read as 32-byte vectors
template <bool XBF, bool OBF>
__global__ __launch_bounds__(256) void k_segmax_bn_bwd_v8(const void* __restrict__ x, const float* __restrict__ out,
                                                          const int* __restrict__ arg, const float* __restrict__ dout,
                                                          const int* __restrict__ inv, long long N, int C,
                                                          const float* __restrict__ a, const float* __restrict__ c0,
                                                          const float* __restrict__ c1, void* __restrict__ dx) {
  const int cv = C >> 3;
  const long long total = N * cv;
  for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    const long long i = t / cv;
    const int c = (int)(t % cv) << 3;
    const long long q = (long long)inv[i] * C + c;
    float xv[8], ov[8], gv[8], av[8], k0[8], k1[8], o[8];
    dec_ld8<XBF>(x, i * C + c, xv);
    dec_ld8<false>(out, q, ov);
    dec_ld8<false>(dout, q, gv);
    dec_ld8<false>(a, c, av);
    dec_ld8<false>(c0, c, k0);
    dec_ld8<false>(c1, c, k1);
    const int4 a0 = *reinterpret_cast<const int4*>(arg + q), a1 = *reinterpret_cast<const int4*>(arg + q + 4);
    const int ai[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float dh = (ai[j] == (int)i && ov[j] > 0.f) ? gv[j] : 0.f;
      o[j] = fmaf(av[j], dh, fmaf(k1[j], xv[j], k0[j]));
    }
    dec_st8<OBF>(dx, i * C + c, o);
  }
}

// Same sums without touching x: at an arg-max with out > 0, out = a x + b exactly as the forward rounded it, so
// x = (out - b) / a (|a| >= 1e-30; the gather of x through arg stays as the fallback for a dead channel).  Turns
// 16 M scattered 2-byte reads into one streaming pass over (out, dout).  8 channels per thread.
template <bool XBF>
__global__ __launch_bounds__(256) void k_segmax_bwd_stats_v8(const void* __restrict__ x, const float* __restrict__ out,
                                                             const int* __restrict__ arg, const float* __restrict__ dout,
                                                             long long M, int C, const float* __restrict__ a,
                                                             const float* __restrict__ b, float* __restrict__ part) {
  extern __shared__ float sh[];  // (rows_per_iter, 2, C)
  const int cv = C >> 3;
  const int rows_per_iter = 256 / cv;
  const int tr = threadIdx.x / cv, c = (threadIdx.x % cv) << 3;
  const long long chunk = (M + gridDim.x - 1) / gridDim.x;
  const long long r0 = blockIdx.x * chunk, r1 = r0 + chunk < M ? r0 + chunk : M;
  float s0[8], s1[8], av[8], bv[8], ia[8];
  dec_ld8<false>(a, c, av);
  dec_ld8<false>(b, c, bv);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s0[j] = s1[j] = 0.f;
    ia[j] = fabsf(av[j]) >= 1e-30f ? 1.f / av[j] : 0.f;
  }
  for (long long p = r0 + tr; p < r1; p += rows_per_iter) {
    const long long q = p * C + c;
    float ov[8], gv[8];
    dec_ld8<false>(out, q, ov);
    dec_ld8<false>(dout, q, gv);
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (ov[j] > 0.f) {
        const float xv = ia[j] != 0.f ? (ov[j] - bv[j]) * ia[j] : dec_ld<XBF>(x, (long long)arg[q + j] * C + c + j);
        s0[j] += gv[j];
        s1[j] = fmaf(gv[j], xv, s1[j]);
      }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    sh[(tr * 2 + 0) * C + c + j] = s0[j];
    sh[(tr * 2 + 1) * C + c + j] = s1[j];
  }
  __syncthreads();
  for (int q = threadIdx.x; q < 2 * C; q += 256) {
    float acc = 0.f;
    for (int rr = 0; rr < rows_per_iter; ++rr) acc += sh[rr * 2 * C + q];
    part[(long long)blockIdx.x * 2 * C + q] = acc;
  }
}

extern "C" int gdmae_segmax_bwd_stats(const void* x, int x_bf16, const float* out, const int* arg, const float* dout,
                                      long long M, int C, const float* a, const float* b,
                                      double* sums /* 2C: {sum dh, sum dh*x} */, void* workspace, void* stream) {
  GD_REQUIRE(C >= 1 && C <= 256, "C <= 256");
  hipStream_t st = (hipStream_t)stream;
  int nblk = (int)(M / 32 > 512 ? 512 : (M / 32 > 0 ? M / 32 : 1));
  const int rpi = 256 / C > 0 ? 256 / C : 1;
  const size_t lds = (size_t)rpi * 2 * C * sizeof(float);
  float* part = (float*)workspace;
  if (a && b && C % 8 == 0 && 256 % (C / 8) == 0) {
    const size_t lds8 = (size_t)(256 / (C / 8)) * 2 * C * sizeof(float);
    if (x_bf16)
      hipLaunchKernelGGL((k_segmax_bwd_stats_v8<true>), dim3(nblk), dim3(256), lds8, st, x, out, arg, dout, M, C, a, b, part);
    else
      hipLaunchKernelGGL((k_segmax_bwd_stats_v8<false>), dim3(nblk), dim3(256), lds8, st, x, out, arg, dout, M, C, a, b, part);
  } else if (x_bf16)
    hipLaunchKernelGGL((k_segmax_bwd_stats<true>), dim3(nblk), dim3(256), lds, st, x, out, arg, dout, M, C, part);
  else
    hipLaunchKernelGGL((k_segmax_bwd_stats<false>), dim3(nblk), dim3(256), lds, st, x, out, arg, dout, M, C, part);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_partials_to_f64, dim3(gd_div_up(2 * C, 16)), dim3(256), 0, st, part, nblk, 2 * C, sums);
  GD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gdmae_segmax_bn_bwd(const void* x, int x_bf16, const float* out, const int* arg, const float* dout,
                                   const int* inverse32, long long N, int C, const float* a, const float* c0, const float* c1,
                                   void* dx, int dx_bf16, void* stream) {
  if (N <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(dec_grid(N * C)), block(256);
  const bool v8 = C % 8 == 0;
  const dim3 grid8(dec_grid(N * C / 8));
#define GD_SB(XB, OB)                                                                                                              \
  do {                                                                                                                             \
    if (v8) hipLaunchKernelGGL((k_segmax_bn_bwd_v8<XB, OB>), grid8, block, 0, st, x, out, arg, dout, inverse32, N, C, a, c0, c1, dx); \
    else hipLaunchKernelGGL((k_segmax_bn_bwd<XB, OB>), grid, block, 0, st, x, out, arg, dout, inverse32, N, C, a, c0, c1, dx);      \
  } while (0)
  if (x_bf16) { if (dx_bf16) GD_SB(true, true); else GD_SB(true, false); }
  else { if (dx_bf16) GD_SB(false, true); else GD_SB(false, false); }
#undef GD_SB
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Backward of the decoder's dense 3x3 convolution restricted to the sites that need it.
//
// The reference back-propagates conv_out (spt_backbone_mae.py:46-52) densely.  Here the output gradient is
//   dY[u] = k0 + k1 * Y[u] + (u is a pillar site ? rows[pillar(u)] : 0)       (BatchNorm2d chain rule, per channel)
// and the input gradient / weight gradient are only needed at the active sites of each source stage, so the
// caller asks for the 9 shifted output-gradient rows of those sites ("taps"):
//   out[t, k, :] = dY[site[t] - k],  k = (ky+1)*3 + (kx+1),  zero outside the map
// and feeds them to two GEMMs (dZ rows = taps @ W^T-matrix, dW = taps^T @ Z-rows).  dY itself is never
// materialised.  One thread = 8 channels of one (site, tap): 16-byte bf16 (or 2x16-byte fp32) accesses.
// ------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(256) void k_conv_grad_taps(const void* __restrict__ Ymap, GdTiles T, const float* __restrict__ k0,
                                                        const float* __restrict__ k1, const float* __restrict__ rows,
                                                        const int* __restrict__ cell2pillar, const int* __restrict__ site,
                                                        long long n, int H, int W, int C, void* __restrict__ out) {
  const int cv = C >> 3;
  const long long total = n * 9 * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(i % cv);
    const long long tk = i / cv;
    const int k = (int)(tk % 9);
    const long long t = tk / 9;
    const int s = site[t];
    const int x = s % W;
    const int r = s / W;
    const int y = r % H;
    const int uy = y - (k / 3 - 1), ux = x - (k % 3 - 1);
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    if (uy >= 0 && uy < H && ux >= 0 && ux < W) {
      const long long u = (long long)(r - y + uy) * W + ux;
      const int c0 = v * 8;
      float yv[8];
      const char* yrow = gd_y_row<BF ? 2 : 4>(Ymap, T, (r - y) / H, uy, ux, H, W, C);
      if (BF) {
        const uint4 q = ((const uint4*)yrow)[c0 >> 3];
        const unsigned w4[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          yv[2 * j] = __uint_as_float(w4[j] << 16);
          yv[2 * j + 1] = __uint_as_float(w4[j] & 0xFFFF0000u);
        }
      } else {
        const float4 q0 = ((const float4*)yrow)[c0 >> 2], q1 = ((const float4*)yrow)[(c0 >> 2) + 1];
        yv[0] = q0.x; yv[1] = q0.y; yv[2] = q0.z; yv[3] = q0.w; yv[4] = q1.x; yv[5] = q1.y; yv[6] = q1.z; yv[7] = q1.w;
      }
      const float4 ka = ((const float4*)k0)[c0 >> 2], kb = ((const float4*)k0)[(c0 >> 2) + 1];
      const float4 la = ((const float4*)k1)[c0 >> 2], lb = ((const float4*)k1)[(c0 >> 2) + 1];
      const float kk0[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
      const float kk1[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaf(kk1[j], yv[j], kk0[j]);
      const int p = cell2pillar[u];
      if (p >= 0) {
        const float4 r0 = ((const float4*)rows)[((long long)p * C + c0) >> 2], r1 = ((const float4*)rows)[(((long long)p * C + c0) >> 2) + 1];
        o[0] += r0.x; o[1] += r0.y; o[2] += r0.z; o[3] += r0.w; o[4] += r1.x; o[5] += r1.y; o[6] += r1.z; o[7] += r1.w;
      }
    }
    if (BF) {
      uint4 q;
      q.x = dec_f2bf(o[0]) | ((unsigned)dec_f2bf(o[1]) << 16);
      q.y = dec_f2bf(o[2]) | ((unsigned)dec_f2bf(o[3]) << 16);
      q.z = dec_f2bf(o[4]) | ((unsigned)dec_f2bf(o[5]) << 16);
      q.w = dec_f2bf(o[6]) | ((unsigned)dec_f2bf(o[7]) << 16);
      ((uint4*)out)[i] = q;
    } else {
      ((float4*)out)[2 * i] = make_float4(o[0], o[1], o[2], o[3]);
      ((float4*)out)[2 * i + 1] = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
}

extern "C" int gdmae_conv3x3_grad_taps(const void* Ymap, int y_bf16, const int* tile_slot, const void* ybg, const float* k0,
                                       const float* k1, const float* rows, const int* cell2pillar, const int* site, long long n,
                                       int H, int W, int C, void* out, void* stream) {
  GD_REQUIRE(C % 8 == 0, "C must be a multiple of 8");
  if (n <= 0) return 0;
  long long g = (n * 9 * (C / 8) + 255) / 256;
  if (g > 65536) g = 65536;
  const GdTiles T{tile_slot, ybg, (H + 7) / 8, (W + 7) / 8};
  if (y_bf16) hipLaunchKernelGGL(k_conv_grad_taps<true>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, Ymap, T, k0, k1, rows, cell2pillar, site, n, H, W, C, out);
  else hipLaunchKernelGGL(k_conv_grad_taps<false>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, Ymap, T, k0, k1, rows, cell2pillar, site, n, H, W, C, out);
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// Border-region sums for the closed-form background share of the 3x3 conv backward (gdmae_hip/decoder.py):
//   regions 1..8 = row 0, row H-1, column 0, column W-1, corners (0,0), (0,W-1), (H-1,0), (H-1,W-1)
//   out[r-1][c]     = sum of Y[b, y, x, c] over the sites of region r                (r = 1..8)
//   out[8 + r-1][c] = sum of rows[p, c] over the pillars whose cell lies in region r
// Deterministic (fixed summation order).  k_border_partial: blocks [0, 4B) = one (edge, batch) each; the other
// blocks = four wavefronts each, one wavefront per (chunk of the pillar list, region) doing a ballot scan.
// ------------------------------------------------------------------------------------------
#define DEC_BORDER_CHUNKS 64
template <bool BF>
__global__ __launch_bounds__(256) void k_border_partial(const void* __restrict__ Y, GdTiles T, const float* __restrict__ rows,
                                                        const int* __restrict__ pillar_cell, int M, int B, int H, int W, int C,
                                                        float* __restrict__ part_edge, float* __restrict__ part_rows) {
  extern __shared__ float sh[];
  const int cv = C >> 3;
  if ((int)blockIdx.x < 4 * B) {
    const int e = blockIdx.x / B, b = blockIdx.x % B;      // e: 0 = row 0, 1 = row H-1, 2 = col 0, 3 = col W-1
    const int len = e < 2 ? W : H;
    const int slots = 256 / cv;
    const int tr = threadIdx.x / cv, c = (threadIdx.x % cv) << 3;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (tr < slots)
      for (int i0 = tr; i0 < len; i0 += 4 * slots) {       // four sites in flight (tile lookup + row: two dependent loads each), added
        float v[4][8];                                      // in the order of the one-at-a-time walk (bit-identical sums)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * slots < len ? i0 + u * slots : len - 1;
          const int y = e == 0 ? 0 : (e == 1 ? H - 1 : i);
          const int x = e == 2 ? 0 : (e == 3 ? W - 1 : i);
          dec_ld8<BF>(gd_y_row<BF ? 2 : 4>(Y, T, b, y, x, H, W, C), c, v[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool ok = i0 + u * slots < len;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = ok ? acc[j] + v[u][j] : acc[j];
        }
      }
    if (tr < slots) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sh[tr * C + c + j] = acc[j];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < C; q += 256) {
      float s = 0.f;
      for (int r = 0; r < slots; ++r) s += sh[r * C + q];
      part_edge[(long long)blockIdx.x * C + q] = s;
    }
    return;
  }
  // pillar rows: one wavefront per (chunk of the pillar list, region); lanes own channels lane, lane + 64, ...
  const int lane = threadIdx.x & 63;
  const int bb = (int)blockIdx.x - 4 * B;
  const int chunk = bb >> 1;
  const int reg = (bb & 1) * 4 + (threadIdx.x >> 6);   // 0..7
  const int per = ((M + DEC_BORDER_CHUNKS - 1) / DEC_BORDER_CHUNKS + 63) / 64 * 64;
  const int pb = chunk * per, pe = pb + per < M ? pb + per : M;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};                                   // C <= 256
  for (int p0 = pb; p0 < pe; p0 += 64) {
    const int p = p0 + lane;
    bool in = false;
    if (p < pe) {
      const int cell = pillar_cell[p];
      const int x = cell % W, y = (cell / W) % H;
      const bool y0 = y == 0, yl = y == H - 1, x0 = x == 0, xl = x == W - 1;
      in = reg == 0 ? y0 : reg == 1 ? yl : reg == 2 ? x0 : reg == 3 ? xl : reg == 4 ? (y0 && x0) : reg == 5 ? (y0 && xl)
                                                                         : reg == 6 ? (yl && x0) : (yl && xl);
    }
    unsigned long long m = __ballot(in);
    while (m) {
      const int k = __ffsll((long long)m) - 1;
      m &= m - 1;
      const float* rp = rows + (long long)(p0 + k) * C;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (lane + 64 * j < C) acc[j] += rp[lane + 64 * j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (lane + 64 * j < C) part_rows[((long long)chunk * 8 + reg) * C + lane + 64 * j] = acc[j];
}

template <bool BF>
__global__ __launch_bounds__(256) void k_border_final(const void* __restrict__ Y, GdTiles T, const float* __restrict__ part_edge,
                                                      const float* __restrict__ part_rows, int B, int H, int W, int C,
                                                      double* __restrict__ out) {
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < 8 * C; q += gridDim.x * blockDim.x) {
    const int r = q / C, c = q % C;
    double s = 0.0;
    if (r < 4) {
      for (int b = 0; b < B; ++b) s += (double)part_edge[(long long)(r * B + b) * C + c];
    } else {
      const int y = (r == 4 || r == 5) ? 0 : H - 1, x = (r == 4 || r == 6) ? 0 : W - 1;
      for (int b = 0; b < B; ++b) s += (double)dec_ld<BF>(gd_y_row<BF ? 2 : 4>(Y, T, b, y, x, H, W, C), c);
    }
    out[q] = s;
    double t = 0.0;
    for (int ch = 0; ch < DEC_BORDER_CHUNKS; ++ch) t += (double)part_rows[(long long)ch * 8 * C + q];
    out[8 * C + q] = t;
  }
}

extern "C" size_t gdmae_border_sums_workspace_bytes(int B, int C) {
  return (size_t)(4 * B + 8 * DEC_BORDER_CHUNKS) * C * sizeof(float);
}

extern "C" int gdmae_border_sums(const void* Y, int y_bf16, const int* tile_slot, const void* ybg, const float* rows,
                                 const int* pillar_cell, int M, int B, int H, int W, int C, double* out /* [16][C] */,
                                 void* workspace, void* stream) {
  const GdTiles T{tile_slot, ybg, (H + 7) / 8, (W + 7) / 8};
  GD_REQUIRE(C % 8 == 0 && C <= 256 && 256 % (C / 8) == 0, "border_sums: C in {64, 128, 256}");
  hipStream_t st = (hipStream_t)stream;
  float* part_edge = (float*)workspace;
  float* part_rows = part_edge + (size_t)4 * B * C;
  const size_t lds = (size_t)(256 / (C / 8)) * C * sizeof(float);
  if (y_bf16) hipLaunchKernelGGL((k_border_partial<true>), dim3(4 * B + 2 * DEC_BORDER_CHUNKS), dim3(256), lds, st, Y, T, rows, pillar_cell, M, B, H, W, C, part_edge, part_rows);
  else hipLaunchKernelGGL((k_border_partial<false>), dim3(4 * B + 2 * DEC_BORDER_CHUNKS), dim3(256), lds, st, Y, T, rows, pillar_cell, M, B, H, W, C, part_edge, part_rows);
  GD_LAUNCH_CHECK();
  if (y_bf16) hipLaunchKernelGGL((k_border_final<true>), dim3(gd_div_up(8 * C, 256)), dim3(256), 0, st, Y, T, part_edge, part_rows, B, H, W, C, out);
  else hipLaunchKernelGGL((k_border_final<false>), dim3(gd_div_up(8 * C, 256)), dim3(256), 0, st, Y, T, part_edge, part_rows, B, H, W, C, out);
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Small algebra of the decoder-head backward (gdmae_hip/decoder.py) as three launches instead of ~20 torch ops:
//   regD[r][o] = cnt[r] k0[o] + regY[r][o] k1[o] + regR[r][o]        (9 border regions of dY = k0 + k1 Y + rows)
//                regY[0] = mean2 * R, regY[1..8] = reg[0..7];  regR[0] = a2 * st2[0:C2], regR[1..8] = reg[8..15]
//   S[k][o]    = sum_r tap_region[k][r] regD[r][o]                   (sum of dY over the sites tap k can reach)
//   W_k[o][i]  = conv_w[o][i][ky][kx], k = 3 ky + kx
//   tot[i]     = sum_{k,o} S[k][o] W_k[o][i]        (column sums of dZ over ALL sites, fp64)
//   dWk[k][o][i] = S[k][o] bg[i]                    (background part of the weight gradient, fp32)
//   Wd[k][o][i]  = W_k[o][i] in the compute dtype   (B operand of the dZ-row GEMMs)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dec_region_S(const double* __restrict__ stats2, const float* __restrict__ ab2,
                                                      const double* __restrict__ st2, const float* __restrict__ k01,
                                                      const double* __restrict__ reg, const double* __restrict__ tap_region,
                                                      const double* __restrict__ cnt, double R, int C2, double* __restrict__ S) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= C2) return;
  const double k0 = (double)k01[o], k1 = (double)k01[C2 + o];
  double regD[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) {
    const double y = r == 0 ? stats2[o] * R : reg[(r - 1) * C2 + o];
    const double rr = r == 0 ? (double)ab2[o] * st2[o] : reg[(8 + r - 1) * C2 + o];
    regD[r] = cnt[r] * k0 + y * k1 + rr;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    double a = 0.0;
#pragma unroll
    for (int r = 0; r < 9; ++r) a += tap_region[k * 9 + r] * regD[r];
    S[k * C2 + o] = a;
  }
}

template <bool BF>
__global__ __launch_bounds__(128) void k_dec_region_W(const double* __restrict__ S, const float* __restrict__ conv_w,
                                                      const void* __restrict__ bgz, int C2, int Cin, float* __restrict__ dWk,
                                                      void* __restrict__ Wd, double* __restrict__ tot_part) {
  const int i = blockIdx.x * 128 + threadIdx.x, oc = blockIdx.y * 8;
  if (i >= Cin) return;
  const double bg = (double)dec_ld<BF>(bgz, i);
  double acc = 0.0;
  for (int oo = 0; oo < 8; ++oo) {
    const int o = oc + oo;
    const float* w = conv_w + ((long long)o * Cin + i) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const float wv = w[k];
      const double s = S[k * C2 + o];
      acc += s * (double)wv;
      const long long e = ((long long)k * C2 + o) * Cin + i;
      dWk[e] = (float)(s * bg);
      dec_st<BF>(Wd, e, wv);
    }
  }
  tot_part[(long long)blockIdx.y * Cin + i] = acc;
}

__global__ __launch_bounds__(256) void k_dec_region_tot(const double* __restrict__ tot_part, int nchunk, int Cin,
                                                        double* __restrict__ tot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cin) return;
  double a = 0.0;
  for (int c = 0; c < nchunk; ++c) a += tot_part[(long long)c * Cin + i];
  tot[i] = a;
}

extern "C" size_t gdmae_decoder_region_workspace_bytes(int C2, int Cin) { return (size_t)(C2 / 8 + 1) * Cin * sizeof(double); }

extern "C" int gdmae_decoder_region_algebra(const double* stats2, const float* ab2, const double* st2, const float* k01,
                                            const double* reg, const double* tap_region, const double* cnt, double R, int C2,
                                            int Cin, const float* conv_w, const void* bgz, int cdt_bf16, double* S, double* tot,
                                            float* dWk, void* Wd, void* workspace, void* stream) {
  GD_REQUIRE(C2 > 0 && C2 % 8 == 0 && Cin > 0, "decoder region algebra: C2 must be a multiple of 8");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_dec_region_S, dim3(gd_div_up(C2, 256)), dim3(256), 0, st, stats2, ab2, st2, k01, reg, tap_region, cnt, R, C2, S);
  GD_LAUNCH_CHECK();
  const dim3 grid(gd_div_up(Cin, 128), C2 / 8);
  double* part = (double*)workspace;
  if (cdt_bf16) hipLaunchKernelGGL((k_dec_region_W<true>), grid, dim3(128), 0, st, (const double*)S, conv_w, bgz, C2, Cin, dWk, Wd, part);
  else hipLaunchKernelGGL((k_dec_region_W<false>), grid, dim3(128), 0, st, (const double*)S, conv_w, bgz, C2, Cin, dWk, Wd, part);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_dec_region_tot, dim3(gd_div_up(Cin, 256)), dim3(256), 0, st, (const double*)part, C2 / 8, Cin, tot);
  GD_LAUNCH_CHECK();
  return 0;
}


// ------------------------------------------------------------------------------------------
// Round 3: the 3x3 conv_out backward WITHOUT the tap matrix.  Round 2 wrote the 9 shifted output-gradient rows of every active
// site (k_conv_grad_taps: 1.4 GB at config B) and read them back twice (two library GEMMs).  Now
//   gdmae_decoder_dy             dYc = bf16(k0 + k1 * Yc + rows[pillar]) on the active tiles, ONCE, in Yc's tile-compact layout
//   gdmae_decoder_site_rulebook  nbr[t, k] = row of dYc holding site[t] - offset(k), -1 outside the map (geometry: built with the plan)
// and the input-gradient rows / the weight gradient are the implicit-GEMM sparse convolution (spconv.hip) and the grouped TN
// launch with gathered rows (dw_grouped.hip) over that rulebook: every dY row is read through L2, nothing is materialised
// nine-fold.  Same values as the taps (one bf16 rounding of dY), same tap convention k = (ky + 1) * 3 + (kx + 1).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_decoder_dy(const unsigned short* __restrict__ Yc, const int* __restrict__ tile_list, int n_act,
                                                    const float* __restrict__ k0, const float* __restrict__ k1,
                                                    const float* __restrict__ rows, const int* __restrict__ cell2pillar, int H, int W,
                                                    int TH, int TW, int C, unsigned short* __restrict__ dYc) {
  // index arithmetic in 32 bits (the launcher checks n_act * 64 * C / 8 < 2^31): a 64-bit division and remainder per 16 bytes are
  // emulated (~200 instructions) next to 16 bytes of traffic each way
  const unsigned cv = (unsigned)C >> 3;
  const unsigned total = (unsigned)n_act * GD_TILE_SITES * cv;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned row32 = i / cv;
    const int v = (int)(i - row32 * cv);
    const long long row = row32;
    const int slot = (int)(row32 >> 6), loc = (int)(row32 & 63);
    const int tile = tile_list[slot];
    const int tx = tile % TW, r = tile / TW, ty = r % TH, b = r / TH;
    const int y = ty * GD_TILE + (loc >> 3), x = tx * GD_TILE + (loc & 7);
    uint4 q = make_uint4(0u, 0u, 0u, 0u);
    if (y < H && x < W) {
      const int c0 = v * 8;
      const uint4 yq = reinterpret_cast<const uint4*>(Yc + row * C)[v];
      const unsigned w4[4] = {yq.x, yq.y, yq.z, yq.w};
      float yv[8], o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        yv[2 * j] = __uint_as_float(w4[j] << 16);
        yv[2 * j + 1] = __uint_as_float(w4[j] & 0xFFFF0000u);
      }
      const float4 ka = ((const float4*)k0)[c0 >> 2], kb = ((const float4*)k0)[(c0 >> 2) + 1];
      const float4 la = ((const float4*)k1)[c0 >> 2], lb = ((const float4*)k1)[(c0 >> 2) + 1];
      const float kk0[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
      const float kk1[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = fmaf(kk1[j], yv[j], kk0[j]);
      const int p = cell2pillar[((long long)b * H + y) * W + x];
      if (p >= 0) {
        const float4 r0 = ((const float4*)rows)[((long long)p * C + c0) >> 2], r1 = ((const float4*)rows)[(((long long)p * C + c0) >> 2) + 1];
        o[0] += r0.x; o[1] += r0.y; o[2] += r0.z; o[3] += r0.w; o[4] += r1.x; o[5] += r1.y; o[6] += r1.z; o[7] += r1.w;
      }
      q.x = dec_f2bf(o[0]) | ((unsigned)dec_f2bf(o[1]) << 16);
      q.y = dec_f2bf(o[2]) | ((unsigned)dec_f2bf(o[3]) << 16);
      q.z = dec_f2bf(o[4]) | ((unsigned)dec_f2bf(o[5]) << 16);
      q.w = dec_f2bf(o[6]) | ((unsigned)dec_f2bf(o[7]) << 16);
    }
    reinterpret_cast<uint4*>(dYc + row * C)[v] = q;
  }
}
extern "C" int gdmae_decoder_dy(const void* Yc, const int* tile_list, int n_act, const float* k0, const float* k1, const float* rows,
                                const int* cell2pillar, int H, int W, int C, void* dYc, void* stream) {
  GD_REQUIRE(C % 8 == 0, "decoder_dy: C must be a multiple of 8");
  if (n_act <= 0) return 0;
  GD_REQUIRE((long long)n_act * GD_TILE_SITES * (C / 8) < (1ll << 31), "decoder_dy: active tiles x 64 x C / 8 must fit 31 bits");
  long long g = ((long long)n_act * GD_TILE_SITES * (C / 8) + 255) / 256;
  if (g > 65536) g = 65536;
  hipLaunchKernelGGL(k_decoder_dy, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)Yc, tile_list, n_act, k0, k1, rows,
                     cell2pillar, H, W, (H + 7) / 8, (W + 7) / 8, C, (unsigned short*)dYc);
  GD_LAUNCH_CHECK();
  return 0;
}

// nbr[t * 9 + k] = tile-compact row of site[t] - (ky - 1, kx - 1), -1 outside the map (or in a tile that is not active: cannot
// happen for the sites the active-tile set was built from).  n_dev: device count of sites (capacity `cap` bounds the grid).
__global__ __launch_bounds__(256) void k_decoder_site_rulebook(const int* __restrict__ site, const int* __restrict__ n_dev, int sites_per_tok,
                                                               const int* __restrict__ tile_slot, int H, int W, int TH, int TW,
                                                               int* __restrict__ nbr) {
  const long long total = (long long)(*n_dev) * sites_per_tok * 9;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long t = total < (1ll << 31) ? (long long)((unsigned)i / 9u) : i / 9;      // (32-bit division where the index fits)
    const int k = (int)(i - t * 9);
    const int s = site[t];
    const int x = s % W, r = s / W, y = r % H, b = r / H;
    const int uy = y - (k / 3 - 1), ux = x - (k % 3 - 1);
    int v = -1;
    if (uy >= 0 && uy < H && ux >= 0 && ux < W) {
      const int sl = tile_slot[(b * TH + (uy >> 3)) * TW + (ux >> 3)];
      if (sl >= 0) v = sl * GD_TILE_SITES + (uy & 7) * GD_TILE + (ux & 7);
    }
    nbr[i] = v;
  }
}
int gd_decoder_site_rulebook(const int* site, const int* n_dev, int sites_per_tok, long long cap_sites, const int* tile_slot, int H, int W,
                             int* nbr, hipStream_t st) {
  long long g = (cap_sites * 9 + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(k_decoder_site_rulebook, dim3((int)g), dim3(256), 0, st, site, n_dev, sites_per_tok, tile_slot, H, W, (H + 7) / 8,
                     (W + 7) / 8, nbr);
  GD_LAUNCH_CHECK();
  return 0;
}
extern "C" int gdmae_decoder_site_rulebook(const int* site, const int* n_dev, int sites_per_tok, long long cap_sites, const int* tile_slot,
                                           int H, int W, int* nbr, void* stream) {
  return gd_decoder_site_rulebook(site, n_dev, sites_per_tok, cap_sites, tile_slot, H, W, nbr, (hipStream_t)stream);
}
