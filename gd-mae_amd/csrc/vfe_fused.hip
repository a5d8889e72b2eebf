// DynVFE first point layer without its (N, 64) intermediates (reference pcdet/models/backbones_3d/vfe/dyn_vfe.py:74-109,
// network_utils.py:7-21: f = [xyz - voxel centre, raw point, xyz - pillar mean];  h = f W^T;  y = relu(BatchNorm1d_train(h))).
//
// N = 1.4 M points per 8-frame batch, so every (N, 64) tensor is 180-370 MB of HBM traffic; h is 11 multiply-adds per
// element.  Here h only ever lives in MFMA accumulators and is recomputed where it is needed:
//   forward   k_vfe1<STATS>  : column sums of h, h^2        -> gd_bn_fold_from_partials (mean/rstd/affine/running stats)
//             k_vfe1<APPLY>  : y = relu(a h + b)            -> the only (N, 64) write of the layer
//   backward  k_vfe1<BSTATS> : dh = g [a h + b > 0];  column sums of dh, dh*h  -> gdmae_bn_bwd_coeffs (dgamma, dbeta, c0, c1)
//             k_vfe1<DW>     : dW += (a dh + c0 + c1 h)^T f  (second MFMA, operand A = the accumulator registers)
// One wave owns a tile of 32 points: v_mfma_f32_32x32x2_f32 with A = features (points x k), B = W^T (k x 32 columns),
// two column blocks holding the even / odd columns so that a lane owns the adjacent pair (2n, 2n+1) of its rows and a
// store instruction writes whole 128-byte (bf16) / 256-byte (fp32) rows.  fp32 inputs, fp32 MFMA: the same code serves
// the fp32 parity mode and the bf16 throughput mode (only y / g change type).  The recomputation is bit-identical
// across the four kernels (same instruction sequence), so the ReLU mask of the backward equals the forward's.
#include "common.h"
#include "gemm.h"

int gd_bn_fold_from_partials(hipStream_t st, const float* part, int nblk, int C, double count, const float* gamma,
                             const float* beta, double eps, double momentum, float* running_mean, float* running_var,
                             long long* num_batches, double* stats, float* ab, float* mv);
int gd_partials_to_f64(hipStream_t st, const float* part, int nblk, int C2, double* out);
extern "C" int gdmae_bn_bwd_coeffs(const double* st, int n_st, const double* stats, const float* ab, const float* gamma, int C,
                                   double count, const double* tot, float* dgamma, float* dbeta, int accumulate, float* c01,
                                   void* stream);

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int VF_C = 64;          // output channels of the first point layer (gd_mae_ssl.yaml MLPS [[64, 128]])
constexpr int VF_WAVES = 4;       // waves per workgroup, one 32-point tile per wave and iteration
constexpr int VF_MAX_GRID = 1024;

struct VfeGeom {
  float lo[3], vs[3];
  int out_f16 = 0;     // VF_APPLY with 16-bit output: the rows leave as fp16 instead of bf16 (gdmae_vfe_point_layer_fwd out_bf16 = 2)
};

enum { VF_STATS = 0, VF_APPLY = 1, VF_BSTATS = 2, VF_DW = 3 };

__device__ inline unsigned short vf_f2bf(float f) { return gd_to_bf16(f); }   // round to nearest even

// decorated features of a point (all zero past the end): the arithmetic of k_decorate (segment.hip), kept in registers.
// `cpp` (coords per pillar): `coords` holds one row per PILLAR (voxel_coords) instead of one per point - the layout of
// the pillar-major rows (gdmae_pillar_major_rows), where the per-point coordinate table is not needed.
// The loads of a point are two dependent round trips (its row and pillar id, then the pillar's cell and mean); a wave walks ~11
// tiles, so they are requested as two pipeline stages: stage A of tile k + 2 and stage B of tile k + 1 while tile k is multiplied
// (every request unconditional on a clamped index, nothing looked at before its use - DESIGN section 9, rules 1 - 3).
template <int F>
struct VfA {
  float r[F + 1];
  int pil;
};
struct VfB {
  long long c1, c2, c3;      // z, y, x cell
  float m[3];
};
template <int F>
__device__ __forceinline__ void vf_load_a(const float* __restrict__ pts, const int* __restrict__ inv, long long i, long long N, VfA<F>& A) {
  const long long ic = i < N ? i : N - 1;
  const float* r = pts + ic * (F + 1);
#pragma unroll
  for (int k = 0; k < F + 1; ++k) A.r[k] = r[k];
  A.pil = inv[ic];
}
template <int F>
__device__ __forceinline__ void vf_load_b(const long long* __restrict__ coords, const float* __restrict__ mean, int cpp, long long i,
                                          long long N, int pil, VfB& B) {
  const long long ic = i < N ? i : N - 1;
  const long long* c = coords + 4 * (cpp ? (long long)pil : ic);   // b, z, y, x
  const float* m = mean + (long long)pil * F;
  B.c1 = c[1]; B.c2 = c[2]; B.c3 = c[3];
  B.m[0] = m[0]; B.m[1] = m[1]; B.m[2] = m[2];
}
template <int F>
__device__ __forceinline__ void vf_features(const VfA<F>& A, const VfB& B, bool live, const VfeGeom& G, float (&f)[F + 6]) {
  const float x = A.r[1], y = A.r[2], z = A.r[3];
  f[0] = __fsub_rn(x, __fadd_rn(__fmul_rn(__fadd_rn((float)B.c3, 0.5f), G.vs[0]), G.lo[0]));
  f[1] = __fsub_rn(y, __fadd_rn(__fmul_rn(__fadd_rn((float)B.c2, 0.5f), G.vs[1]), G.lo[1]));
  f[2] = __fsub_rn(z, __fadd_rn(__fmul_rn(__fadd_rn((float)B.c1, 0.5f), G.vs[2]), G.lo[2]));
#pragma unroll
  for (int k = 0; k < F; ++k) f[3 + k] = A.r[1 + k];
  f[3 + F] = __fsub_rn(x, B.m[0]);
  f[4 + F] = __fsub_rn(y, B.m[1]);
  f[5 + F] = __fsub_rn(z, B.m[2]);
#pragma unroll
  for (int k = 0; k < F + 6; ++k) f[k] = live ? f[k] : 0.f;
}
// gradient rows of a tile (this lane: columns 2 n, 2 n + 1 of its 16 accumulator rows), raw - unpacked where they are used
template <bool BF>
struct VfG {
  unsigned u[BF ? 16 : 1];
  float2 v[BF ? 1 : 16];
};
template <bool BF>
__device__ __forceinline__ void vf_load_g(const void* __restrict__ g, long long base, long long N, int n, int half, VfG<BF>& Gr) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long long row = base + 8 * (r / 4) + 4 * half + (r % 4);
    const long long rc = row < N ? row : N - 1;      // branch-free: 16 loads in flight
    if (BF) Gr.u[r] = ((const unsigned*)g)[rc * (VF_C / 2) + n];
    else Gr.v[r] = ((const float2*)g)[rc * (VF_C / 2) + n];
  }
}

// MODE STATS : part[(block, 2, 64)] = column sums of h, h^2
// MODE APPLY : out[(N, 64)] = relu(a h + b)
// MODE BSTATS: part[(block, 2, 64)] = column sums of dh, dh*h          (dh = g [a h + b > 0])
// MODE DW    : part[(block, 64, D)] = sum_i (a dh + c0 + c1 h)[i, c] f[i, d]
template <int F, int MODE, bool BF>
__global__ __launch_bounds__(VF_WAVES * 64) void k_vfe1(const float* __restrict__ pts, const long long* __restrict__ coords,
                                                        const int* __restrict__ inv, const float* __restrict__ mean,
                                                        int cpp, long long N, VfeGeom G, const float* __restrict__ W,
                                                        const float* __restrict__ ab, const float* __restrict__ c01,
                                                        const void* __restrict__ g, void* __restrict__ out,
                                                        float* __restrict__ part) {
  constexpr int D = F + 6;
  constexpr int KS = (D + 1) / 2;          // MFMA k-steps (K = 2 each)
  constexpr int DP = 2 * KS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, half = lane >> 5;
  __shared__ float sh[VF_WAVES * (MODE == VF_DW ? VF_C * D : 4 * VF_C)];
  __shared__ float shf[MODE == VF_DW ? VF_WAVES * 32 * DP : 1];

  // B operand of h = f W^T: lane (n, half) holds W[col(blk, n)][2 s + half], col(blk, n) = 2 n + blk
  float wb[2][KS];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int d = 2 * s + half;
      wb[blk][s] = d < D ? W[(2 * n + blk) * D + d] : 0.f;
    }
  float ca[2] = {0.f, 0.f}, cb[2] = {0.f, 0.f}, c0[2] = {0.f, 0.f}, c1[2] = {0.f, 0.f};
  if (MODE != VF_STATS) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      ca[blk] = ab[2 * n + blk];
      cb[blk] = ab[VF_C + 2 * n + blk];
      if (MODE == VF_DW) {
        c0[blk] = c01[2 * n + blk];
        c1[blk] = c01[VF_C + 2 * n + blk];
      }
    }
  }
  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
  f32x16 accw[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) accw[0][r] = accw[1][r] = 0.f;

  const long long ntiles = (N + 31) / 32;
  const long long tstride = (long long)gridDim.x * VF_WAVES;
  constexpr bool HAS_G = MODE == VF_BSTATS || MODE == VF_DW;
  // STATS / APPLY: software pipeline over this wave's tiles t, t + tstride, ...: (A1, B1) = the tile being multiplied, A2 = the next
  // tile's rows, requested an iteration ago; its stage B and the stage A of the tile after it are requested at the top.  Measured
  // (tools/vfe_times.sh, 8 frames): STATS 41 -> 38 us, APPLY 71 -> 53 us (the wait that used to sit behind its stores); the modes
  // with gradient rows LOSE with it (BSTATS 57 -> 63, DW 104 -> 109: 37 / 30 more registers = a wave per SIMD less, and their
  // tiles are bound by the fp32 MFMA + VALU work, not by the round trips) and keep the plain order.
  constexpr bool PIPE = !HAS_G;
  VfA<F> A1, A2, A3;
  VfB B1, B2;
  VfG<BF> G1;
  if (PIPE) {
    const long long t = (long long)blockIdx.x * VF_WAVES + wave;
    vf_load_a<F>(pts, inv, t * 32 + n, N, A1);
    vf_load_a<F>(pts, inv, (t + tstride) * 32 + n, N, A2);
    vf_load_b<F>(coords, mean, cpp, t * 32 + n, N, A1.pil, B1);
  }
  for (long long t0 = (long long)blockIdx.x * VF_WAVES; t0 < ntiles; t0 += tstride) {
    const long long base = (t0 + wave) * 32;   // may lie past N: the tile is then all zero rows
    if (PIPE) {
      vf_load_a<F>(pts, inv, base + 2 * tstride * 32 + n, N, A3);
      vf_load_b<F>(coords, mean, cpp, base + tstride * 32 + n, N, A2.pil, B2);       // A2 arrived during the previous tile
      __builtin_amdgcn_sched_barrier(0);          // the requests go out first
    } else {
      // gradient rows of the tile first: 16 independent loads in flight while the features / h are computed
      vf_load_g<BF>(g, base, N, n, half, G1);
      vf_load_a<F>(pts, inv, base + n, N, A1);
      vf_load_b<F>(coords, mean, cpp, base + n, N, A1.pil, B1);
    }
    float g0[16], g1[16];
    if (HAS_G) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long row = base + 8 * (r / 4) + 4 * half + (r % 4);
        if (BF) {
          g0[r] = row < N ? __uint_as_float(G1.u[r] << 16) : 0.f;
          g1[r] = row < N ? __uint_as_float(G1.u[r] & 0xFFFF0000u) : 0.f;
        } else {
          g0[r] = row < N ? G1.v[r].x : 0.f;
          g1[r] = row < N ? G1.v[r].y : 0.f;
        }
      }
    }
    float f[D];
    vf_features<F>(A1, B1, base + n < N, G, f);
    f32x16 h[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) h[0][r] = h[1][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const float f0 = f[2 * s], f1 = (2 * s + 1 < D) ? f[2 * s + 1] : 0.f;
      const float a = half ? f1 : f0;
      h[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[0][s], h[0], 0, 0, 0);
      h[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wb[1][s], h[1], 0, 0, 0);
    }
    // accumulator register r of lane (n, half): row base + 8 (r / 4) + 4 half + (r % 4), columns 2 n and 2 n + 1
    if (MODE == VF_STATS) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
          const float v = h[blk][r];      // rows past N are exactly zero
          s1[blk] += v;
          s2[blk] = fmaf(v, v, s2[blk]);
        }
    } else if (MODE == VF_APPLY) {
      // the prefetched rows are waited for HERE, before this tile's stores go out: loads and stores retire out of order with respect
      // to each other, so with stores pending the next iteration's first look at a loaded register would drain both (rule 4)
#pragma unroll
      for (int k = 0; k < F + 1; ++k) asm volatile("" : "+v"(A3.r[k]));
      asm volatile("" : "+v"(A3.pil));
      asm volatile("" : "+v"(B2.c1), "+v"(B2.c2), "+v"(B2.c3), "+v"(B2.m[0]), "+v"(B2.m[1]), "+v"(B2.m[2]));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long long row = base + 8 * (r / 4) + 4 * half + (r % 4);
        if (row < N) {
          const float y0 = fmaxf(fmaf(ca[0], h[0][r], cb[0]), 0.f), y1 = fmaxf(fmaf(ca[1], h[1][r], cb[1]), 0.f);
          if (BF) ((unsigned*)out)[row * (VF_C / 2) + n] = G.out_f16 ? gd_pack_f16(y0, y1) : ((unsigned)vf_f2bf(y0) | ((unsigned)vf_f2bf(y1) << 16));
          else ((float2*)out)[row * (VF_C / 2) + n] = make_float2(y0, y1);
        }
      }
    } else {
      if (MODE == VF_DW) {
        // the feature tile goes through this wave's own LDS rows (written and read by the same wave: LDS executes a
        // wave's instructions in order, so only the compiler has to be kept from reordering)
        if (half == 0) {
#pragma unroll
          for (int d = 0; d < DP; ++d) shf[(wave * 32 + n) * DP + d] = d < D ? f[d] : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pr = 8 * (r / 4) + 4 * half + (r % 4);
        const float dh0 = fmaf(ca[0], h[0][r], cb[0]) > 0.f ? g0[r] : 0.f;
        const float dh1 = fmaf(ca[1], h[1][r], cb[1]) > 0.f ? g1[r] : 0.f;
        if (MODE == VF_BSTATS) {
          s1[0] += dh0;
          s1[1] += dh1;
          s2[0] = fmaf(dh0, h[0][r], s2[0]);
          s2[1] = fmaf(dh1, h[1][r], s2[1]);
        } else {
          // dW[c][d] += sum over the tile's points: A[i = column][k = half] = dx of point pr (this very register),
          // B[k = half][j = d] = f[point pr][d] from LDS; rows past N have f = 0 and a finite dx
          const float dx0 = fmaf(c1[0], h[0][r], fmaf(ca[0], dh0, c0[0]));
          const float dx1 = fmaf(c1[1], h[1][r], fmaf(ca[1], dh1, c0[1]));
          const float bv = n < DP ? shf[(wave * 32 + pr) * DP + n] : 0.f;
          accw[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dx0, bv, accw[0], 0, 0, 0);
          accw[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dx1, bv, accw[1], 0, 0, 0);
        }
      }
      if (MODE == VF_DW) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (PIPE) {
      A1 = A2;
      A2 = A3;
      B1 = B2;
    }
  }

  if (MODE == VF_STATS || MODE == VF_BSTATS) {
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      s1[blk] += __shfl_xor(s1[blk], 32, 64);
      s2[blk] += __shfl_xor(s2[blk], 32, 64);
    }
    if (half == 0) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        sh[wave * 2 * VF_C + 2 * n + blk] = s1[blk];
        sh[wave * 2 * VF_C + VF_C + 2 * n + blk] = s2[blk];
      }
    }
    __syncthreads();
    if (threadIdx.x < 2 * VF_C) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < VF_WAVES; ++w) a += sh[w * 2 * VF_C + threadIdx.x];
      part[(long long)blockIdx.x * 2 * VF_C + threadIdx.x] = a;
    }
  } else if (MODE == VF_DW) {
    // accumulator register r of lane (n = d, half) of block blk: dW[column 2 (8 (r/4) + 4 half + r%4) + blk][d]
    if (n < D) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int col = 2 * (8 * (r / 4) + 4 * half + (r % 4)) + blk;
          sh[wave * VF_C * D + col * D + n] = accw[blk][r];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < VF_C * D; e += VF_WAVES * 64) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < VF_WAVES; ++w) a += sh[w * VF_C * D + e];
      part[(long long)blockIdx.x * VF_C * D + e] = a;
    }
  }
}

// one resident round of workgroups (occupancy query per instantiation, cached), capped by the partial buffers
template <typename K>
int vf_resident_blocks(K kernel, int slot) {
  static int cache[32] = {0};
  if (cache[slot] == 0) {
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, VF_WAVES * 64, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
    const long long b = (long long)per_cu * prop.multiProcessorCount;
    cache[slot] = (int)(b > VF_MAX_GRID ? VF_MAX_GRID : b);
  }
  return cache[slot];
}

// *grid_out: the number of workgroups launched = the number of partial rows written
template <int MODE>
int vf_launch(hipStream_t st, int F, bool bf, const float* pts, const long long* coords, const int* inv, const float* mean,
              int cpp, long long N, const VfeGeom& G, const float* W, const float* ab, const float* c01, const void* g, void* out,
              float* part, int* grid_out = nullptr) {
  const long long tiles = (N + 31) / 32, blocks = (tiles + VF_WAVES - 1) / VF_WAVES;
  const dim3 block(VF_WAVES * 64);
#define VF_GO(FF, BB)                                                                                                   \
  do {                                                                                                                  \
    const int cap = vf_resident_blocks(k_vfe1<FF, MODE, BB>, ((FF - 3) * 4 + MODE) * 2 + (BB ? 1 : 0));                 \
    const dim3 grid((unsigned)(blocks < cap ? blocks : cap));                                                           \
    if (grid_out) *grid_out = (int)grid.x;                                                                              \
    hipLaunchKernelGGL((k_vfe1<FF, MODE, BB>), grid, block, 0, st, pts, coords, inv, mean, cpp, N, G, W, ab, c01, g, out, part); \
  } while (0)
  if (F == 5) { if (bf) VF_GO(5, true); else VF_GO(5, false); }
  else if (F == 4) { if (bf) VF_GO(4, true); else VF_GO(4, false); }
  else if (F == 3) { if (bf) VF_GO(3, true); else VF_GO(3, false); }
  else GD_REQUIRE(false, "vfe point layer: 3, 4 or 5 point features (x, y, z [, intensity [, elongation]])");
#undef VF_GO
  GD_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// rows of the kept points in pillar-major (CSR) order: points_pm[q] = points[pillar_pts[q]], row_pillar[q] = its pillar.
// n_dev: the device-side point count (gdmae_voxelize counts[0]) - the buffers are capacity-sized, no host sync needed.
__global__ __launch_bounds__(256) void k_pillar_major_rows(const float* __restrict__ pts, int ncols, const int* __restrict__ csr,
                                                           const int* __restrict__ inv, const int* __restrict__ n_dev,
                                                           float* __restrict__ pts_pm, int* __restrict__ rowpil) {
  const long long N = *n_dev;
  for (long long q = blockIdx.x * (long long)blockDim.x + threadIdx.x; q < N; q += (long long)gridDim.x * blockDim.x) {
    const int i = csr[q];
    rowpil[q] = inv[i];
    for (int k = 0; k < ncols; ++k) pts_pm[q * ncols + k] = pts[(long long)i * ncols + k];
  }
}

extern "C" int gdmae_pillar_major_rows(const float* points, int n_cols, const int* pillar_pts, const int* inverse32,
                                       const int* n_dev, long long capacity, float* points_pm, int* row_pillar,
                                       void* stream) {
  if (capacity <= 0) return 0;
  hipLaunchKernelGGL(k_pillar_major_rows, dim3((unsigned)(capacity / 256 + 1 > 8192 ? 8192 : capacity / 256 + 1)), dim3(256), 0, (hipStream_t)stream, points, n_cols,
                     pillar_pts, inverse32, n_dev, points_pm, row_pillar);
  GD_LAUNCH_CHECK();
  return 0;
}

// workspace: per-workgroup partials (the larger of the statistics and the dW partials) + f64 column sums + c0|c1
extern "C" size_t gdmae_vfe_point_layer_workspace_bytes(int n_cols) {
  const int D = n_cols - 1 + 6;
  return gd_align((size_t)VF_MAX_GRID * VF_C * (D > 2 ? D : 2) * sizeof(float)) + gd_align(2 * VF_C * sizeof(double)) +
         gd_align(2 * VF_C * sizeof(float));
}

// y (N, 64) = relu(BatchNorm1d_train(decorate(points) W^T)), row i of y is row i of `points`.  coords_per_pillar: the rows
// are the pillar-major ones of gdmae_pillar_major_rows - `points` = points_pm, `inverse32` = row_pillar and `point_coords`
// = the (M, 4) voxel_coords table, read through the pillar id.  Also the BatchNorm bookkeeping
// (stats f64[128] = mean | rstd, ab f32[128] = a | b, mv f32[128] = mean | biased var, running statistics update).
extern "C" int gdmae_vfe_point_layer_fwd(const float* points, const long long* point_coords, const int* inverse32,
                                         const float* pillar_mean, int coords_per_pillar, long long N, int n_cols,
                                         const float* lo, const float* vs,
                                         const float* W, int C, const float* gamma, const float* beta, double eps,
                                         double momentum, float* running_mean, float* running_var, long long* num_batches,
                                         double* stats, float* ab, float* mv, void* out, int out_bf16, void* workspace,
                                         void* stream) {
  GD_REQUIRE(C == VF_C, "vfe point layer: 64 output channels");
  GD_REQUIRE(N > 0, "vfe point layer: no points");
  hipStream_t st = (hipStream_t)stream;
  VfeGeom G;
  for (int i = 0; i < 3; ++i) { G.lo[i] = lo[i]; G.vs[i] = vs[i]; }
  float* part = (float*)workspace;
  const int F = n_cols - 1;
  int grid = 0;
  int rc = vf_launch<VF_STATS>(st, F, false, points, point_coords, inverse32, pillar_mean, coords_per_pillar, N, G, W, nullptr, nullptr, nullptr,
                               nullptr, part, &grid);
  if (rc) return rc;
  rc = gd_bn_fold_from_partials(st, part, grid, C, (double)N, gamma, beta, eps, momentum, running_mean, running_var,
                                num_batches, stats, ab, mv);
  if (rc) return rc;
  G.out_f16 = out_bf16 == 2;        // 16-bit rows as fp16 (the operand type of gdmae_vfe_max_layer_*_f16)
  return vf_launch<VF_APPLY>(st, F, out_bf16 != 0, points, point_coords, inverse32, pillar_mean, coords_per_pillar, N, G, W, ab, nullptr, nullptr,
                             out, nullptr);
}

// g (N, 64): gradient of y.  dgamma / dbeta / dW (64, 6 + F) are written, or accumulated into when `accumulate`.
extern "C" int gdmae_vfe_point_layer_bwd(const float* points, const long long* point_coords, const int* inverse32,
                                         const float* pillar_mean, int coords_per_pillar, long long N, int n_cols,
                                         const float* lo, const float* vs,
                                         const float* W, int C, const float* gamma, const double* stats, const float* ab,
                                         const void* g, int g_bf16, float* dgamma, float* dbeta, float* dW, int accumulate,
                                         void* workspace, void* stream) {
  GD_REQUIRE(C == VF_C, "vfe point layer: 64 output channels");
  GD_REQUIRE(N > 0, "vfe point layer: no points");
  hipStream_t st = (hipStream_t)stream;
  VfeGeom G;
  for (int i = 0; i < 3; ++i) { G.lo[i] = lo[i]; G.vs[i] = vs[i]; }
  const int F = n_cols - 1, D = F + 6;
  char* p = (char*)workspace;
  float* part = (float*)p;
  p += gd_align((size_t)VF_MAX_GRID * VF_C * (D > 2 ? D : 2) * sizeof(float));
  double* sums = (double*)p;
  p += gd_align(2 * VF_C * sizeof(double));
  float* c01 = (float*)p;
  int grid = 0;
  int rc = vf_launch<VF_BSTATS>(st, F, g_bf16 != 0, points, point_coords, inverse32, pillar_mean, coords_per_pillar, N, G, W, ab, nullptr, g,
                                nullptr, part, &grid);
  if (rc) return rc;
  rc = gd_partials_to_f64(st, part, grid, 2 * C, sums);
  if (rc) return rc;
  rc = gdmae_bn_bwd_coeffs(sums, 2, stats, ab, gamma, C, (double)N, nullptr, dgamma, dbeta, accumulate, c01, stream);
  if (rc) return rc;
  rc = vf_launch<VF_DW>(st, F, g_bf16 != 0, points, point_coords, inverse32, pillar_mean, coords_per_pillar, N, G, W, ab, c01, g, nullptr, part, &grid);
  if (rc) return rc;
  return gd_splitk_acc(st, part, grid, (long long)VF_C * D, dW, accumulate);
}
