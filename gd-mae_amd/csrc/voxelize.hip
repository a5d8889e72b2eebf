// Dynamic pillar voxelization for gfx950: range filter -> pillar key -> direct-address pillar table
// -> canonical CSR of points per pillar -> per-pillar mean.
//
// Replaces, for the GD-MAE hot path (SURVEY.md §8 rows a1-a3 and the rank part of a17):
//   get_in_range_mask + coord build        reference pcdet/utils/common_utils.py:66-76,
//                                           pcdet/models/backbones_3d/vfe/dyn_vfe.py:62-67
//   coords.unique(dim=0, return_inverse)   dyn_vfe.py:68      (ATen sort-based unique)
//   torch_scatter.scatter(reduce='mean')   dyn_vfe.py:81
//   ingroup rank of a point in its pillar  pcdet/ops/sst_ops/src/sst_ops_gpu.cu:22-28 (atomic
//                                           arrival order there; canonical ascending index here)
//
// MI355X design: the pillar key space is the dense BEV grid (B*Z*Y*X cells, 1.75 M for a batch of
// 8 Waymo frames = 7 MB of int32), so "unique" is a direct-address table in HBM/L2 instead of a
// multi-pass sort of 4 x int64 rows: one atomic histogram pass, one packed (flag|count) scan over
// the cells - which yields the pillars already in lexicographic (b,z,y,x) order, i.e. exactly the
// order torch.unique(dim=0) returns - and one fill pass.  Determinism: the fill pass places points
// in arrival order inside a pillar's CSR segment, then one wavefront per pillar ranks the segment
// by point index (<= 64 points: one point per lane, rank by counting through v_readlane), so every
// downstream consumer sees ascending-index order regardless of atomic timing.  No host sync: all
// counts stay on the device (counts[]), launches use bounded grid-stride loops.
#include "common.h"

struct VoxParams {
  float lo[3];
  float vs[3];
  int gx, gy, gz;
  int B;
  int ncols;  // 1 + F
};

// IEEE-exact (p - lo) / vs, truncation toward zero; -ffp-contract must not touch these.
__device__ inline bool vox_coord(float p, float lo, float vs, int g, int& c) {
  float q = __fdiv_rn(__fsub_rn(p, lo), vs);
  // trunc(q) in [0, g-1]  <=>  -1 < q < g   (NaN/inf fail both; q in (-1,0) truncates to 0 and is KEPT,
  // matching `.to(torch.int64)` in common_utils.py:74)
  bool ok = (q > -1.0f) && (q < (float)g);
  c = ok ? (int)q : 0;
  return ok;
}

// Point chunk of a workgroup: chunks are dealt so that the workgroups of one XCD (blockIdx % 8; gridDim.x is a multiple of 8)
// walk ONE contiguous eighth of the point array.  Frames are concatenated in the batch, cells are keyed frame-major, so the
// atomic counters (and later the cell tables) a workgroup touches belong to one XCD's L2 instead of bouncing between all eight
// (k_point_keys: 108 -> ~50 us for 1.44 M points).
#define VOX_FOR_POINTS(i, n0)                                                                                         \
  const long long per8_ = ((n0) + 7) / 8;                                                                             \
  const long long lo8_ = (long long)(blockIdx.x & 7) * per8_;                                                         \
  const long long hi8_ = lo8_ + per8_ < (n0) ? lo8_ + per8_ : (n0);                                                   \
  for (long long i = lo8_ + (long long)(blockIdx.x >> 3) * blockDim.x + threadIdx.x; i < hi8_;                        \
       i += (long long)(gridDim.x >> 3) * blockDim.x)

__global__ __launch_bounds__(256) void k_zero_cells(int* __restrict__ cell_cnt, long long cells, int* __restrict__ big) {
  const long long per8 = ((cells + 7) / 8 + 3) & ~3ll;
  const long long lo = (long long)(blockIdx.x & 7) * per8;
  const long long hi = lo + per8 < cells ? lo + per8 : cells;
  for (long long i = lo + ((long long)(blockIdx.x >> 3) * blockDim.x + threadIdx.x) * 4; i < hi; i += (long long)(gridDim.x >> 3) * blockDim.x * 4) {
    if (i + 4 <= hi) *reinterpret_cast<int4*>(cell_cnt + i) = make_int4(0, 0, 0, 0);
    else
      for (long long j = i; j < hi; ++j) cell_cnt[j] = 0;
  }
  if (blockIdx.x == 0 && threadIdx.x < 2) big[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void k_point_keys(const float* __restrict__ pts, long long n0, VoxParams P,
                                                    int* __restrict__ key, int* __restrict__ slot, int* __restrict__ cell_cnt) {
  VOX_FOR_POINTS(i, n0) {
    const float* r = pts + i * P.ncols;
    float bf = r[0];
    int cx, cy, cz;
    bool ok = vox_coord(r[1], P.lo[0], P.vs[0], P.gx, cx);
    ok &= vox_coord(r[2], P.lo[1], P.vs[1], P.gy, cy);
    ok &= vox_coord(r[3], P.lo[2], P.vs[2], P.gz, cz);
    ok &= (bf > -1.0f) && (bf < (float)P.B);
    int k = -1;
    if (ok) {
      int b = (int)bf;
      k = ((b * P.gz + cz) * P.gy + cy) * P.gx + cx;
      slot[i] = atomicAdd(&cell_cnt[k], 1);   // arrival slot inside the pillar's CSR segment: the fill pass needs no second atomic
    }
    key[i] = k;
  }
}

struct KeepLoad {
  const int* key;
  __device__ int operator()(long long i) const { return key[i] >= 0 ? 1 : 0; }
};
struct KeepStore {
  int* pos;
  __device__ void operator()(long long i, int ex, int) const { pos[i] = ex; }
};

struct CellLoad {
  const int* cell_cnt;
  __device__ unsigned long long operator()(long long c) const {
    int n = cell_cnt[c];
    return n > 0 ? ((1ull << 32) | (unsigned long long)(unsigned)n) : 0ull;
  }
};
struct CellStore {
  VoxParams P;
  int* cell2pillar;
  int* pt_off;
  int* pillar_cell;
  long long* voxel_coords;
  int* sample_off;
  int* big;  // [0] = #items, [1] = #big pillars, then items (pillar, chunk) pairs from big + 2, pillars after big_cap items
  int big_cap;
  __device__ void operator()(long long c, unsigned long long ex, unsigned long long v) const {
    const int p = (int)(ex >> 32);
    const int cnt = (int)(unsigned)v;
    if (big && cnt > GD_WAVE) {   // pillars with more than one wavefront of points: ranked by (pillar, 64-point chunk) work items (null: the
                                  // caller ranks one thread per point and needs no list - two same-address atomics per crowded pillar saved)
      const int nch = (cnt + GD_WAVE - 1) / GD_WAVE;
      const int base = atomicAdd(&big[0], nch);
      for (int k = 0; k < nch; ++k) {
        big[2 + 2 * (base + k)] = p;
        big[2 + 2 * (base + k) + 1] = k;
      }
      big[2 + 2 * big_cap + atomicAdd(&big[1], 1)] = p;
    }
    const int cps = P.gz * P.gy * P.gx;
    const int ci = (int)c;                 // B*Z*Y*X < 2^31: 32-bit divisions (the 64-bit ones are emulated)
    if (ci % cps == 0) sample_off[ci / cps] = p;
    if (v) {
      cell2pillar[c] = p;
      pt_off[p] = (int)(unsigned)ex;
      pillar_cell[p] = ci;
      int x = ci % P.gx;
      int t = ci / P.gx;
      int y = t % P.gy;
      t /= P.gy;
      int z = t % P.gz;
      int b = t / P.gz;
      long long* o = voxel_coords + 4ll * p;
      o[0] = b;
      o[1] = z;
      o[2] = y;
      o[3] = x;
    } else {
      cell2pillar[c] = -1;
    }
  }
};

// counts: [0]=N kept points, [1]=M pillars
__global__ void k_vox_finalize(const unsigned long long* total, const int* n_keep, int B, int* pt_off, int* sample_off,
                               int* counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int M = (int)(*total >> 32);
    int N = (int)(unsigned)(*total);
    pt_off[M] = N;
    sample_off[B] = M;
    counts[0] = *n_keep;
    counts[1] = M;
  }
}

__global__ __launch_bounds__(256) void k_point_fill(const float* __restrict__ pts, long long n0, VoxParams P,
                                                    const int* __restrict__ key, const int* __restrict__ pos,
                                                    const int* __restrict__ cell2pillar, const int* __restrict__ pt_off,
                                                    const int* __restrict__ slot, float* __restrict__ pts_out,
                                                    long long* __restrict__ point_coords, long long* __restrict__ inverse,
                                                    int* __restrict__ inverse32, int* __restrict__ csr_raw,
                                                    int* __restrict__ raw_pillar) {
  VOX_FOR_POINTS(i, n0) {
    int k = key[i];
    if (k < 0) continue;
    int dst = pos[i];
    int p = cell2pillar[k];
    const int q = pt_off[p] + slot[i];
    csr_raw[q] = dst;
    if (raw_pillar) raw_pillar[q] = p;     // pillar of the raw CSR slot: k_rank_points reads it next to csr_raw instead of gathering
    inverse[dst] = p;
    inverse32[dst] = p;
    const float* r = pts + i * P.ncols;
    float* w = pts_out + (long long)dst * P.ncols;
    for (int c = 0; c < P.ncols; ++c) w[c] = r[c];
    int x = k % P.gx;
    int t = k / P.gx;
    int y = t % P.gy;
    t /= P.gy;
    int z = t % P.gz;
    int b = t / P.gz;
    long long* o = point_coords + 4ll * dst;
    o[0] = b;
    o[1] = z;
    o[2] = y;
    o[3] = x;
  }
}

// Sequential (canonical-order) feature sums of one pillar by one wavefront: 64 points are loaded at once (one row
// per lane) and the adds walk them through v_readlane, so the add chain has no dependent global load and the
// result is bit-identical to a sequential CPU index_add_.  `sorted_lane` = id held by this lane for cnt <= 64,
// otherwise ids are re-read from the sorted CSR.
__device__ inline void pillar_mean(const float* __restrict__ pts_out, int ncols, int F, const int* __restrict__ csr, int off,
                                   int cnt, int sorted_lane, int lane, float* __restrict__ mean_row) {
  for (int c0 = 0; c0 < F; c0 += 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int cb = 0; cb < cnt; cb += GD_WAVE) {
      int pid = 0;
      if (cb + lane < cnt) pid = cnt <= GD_WAVE ? sorted_lane : __builtin_nontemporal_load(&csr[off + cb + lane]);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (cb + lane < cnt && c0 + e < F) ? pts_out[(long long)pid * ncols + 1 + c0 + e] : 0.f;
      const int lim = cnt - cb < GD_WAVE ? cnt - cb : GD_WAVE;
      for (int j = 0; j < lim; ++j) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          acc[e] = __fadd_rn(acc[e], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[e]), j)));
      }
    }
    if (lane < 8 && c0 + lane < F) {
      float mine = acc[0];
#pragma unroll
      for (int e = 1; e < 8; ++e) mine = (lane == e) ? acc[e] : mine;
      mean_row[c0 + lane] = __fdiv_rn(mine, (float)cnt);
    }
  }
}

// One wavefront per pillar with <= 64 points (98.5 % of them): one point id per lane, rank by counting through
// v_readlane (canonical ascending order), in-register inverse permutation, sequential mean.  Larger pillars are
// left to k_big_rank / k_big_mean.
__global__ __launch_bounds__(256) void k_pillar_sort_mean(const int* __restrict__ counts, const int* __restrict__ pt_off,
                                                          const int* __restrict__ csr_raw, int* __restrict__ csr,
                                                          int* __restrict__ rank, const float* __restrict__ pts_out,
                                                          int ncols, float* __restrict__ mean, float* __restrict__ pts_pm,
                                                          int* __restrict__ row_pillar) {
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  const int M = counts[1];
  const int F = ncols - 1;
  for (int p = blockIdx.x * 4 + wib; p < M; p += gridDim.x * 4) {
    const int off = __builtin_amdgcn_readfirstlane(pt_off[p]);
    const int cnt = __builtin_amdgcn_readfirstlane(pt_off[p + 1]) - off;   // wave-uniform -> scalar loop control
    if (cnt > GD_WAVE) continue;
    const int own = lane < cnt ? csr_raw[off + lane] : 0x7fffffff;
    int rk = 0;
    for (int j = 0; j < cnt; ++j) rk += (__builtin_amdgcn_readlane(own, j) < own) ? 1 : 0;
    if (lane < cnt) {
      csr[off + rk] = own;
      rank[own] = rk;
      if (pts_pm) {   // pillar-major copy of the row (what gdmae_pillar_major_rows produces in a pass of its own)
        const float* r = pts_out + (long long)own * ncols;
        float* w = pts_pm + (long long)(off + rk) * ncols;
        for (int c = 0; c < ncols; ++c) w[c] = r[c];
        row_pillar[off + rk] = p;
      }
    }
    if (F <= 0) continue;
    int src = 0;   // lane r fetches the id whose rank is r
    for (int j = 0; j < cnt; ++j) src = (__builtin_amdgcn_readlane(rk, j) == lane) ? j : src;
    const int sorted_lane = __shfl(own, src, GD_WAVE);
    pillar_mean(pts_out, ncols, F, csr, off, cnt, sorted_lane, lane, mean + (long long)p * F);
  }
}

// Pillars with > 64 points: one wavefront per (pillar, 64-point chunk) work item ranks its 64 ids against the whole
// pillar (candidates loaded 64 at a time and broadcast by v_readlane), so a 1000-point wall pillar is 16 independent
// ~1000-step items instead of one 16000-step serial tail.
__global__ __launch_bounds__(256) void k_big_rank(const int* __restrict__ big, const int* __restrict__ pt_off,
                                                  const int* __restrict__ csr_raw, int* __restrict__ csr,
                                                  int* __restrict__ rank, const float* __restrict__ pts_out, int ncols,
                                                  float* __restrict__ pts_pm, int* __restrict__ row_pillar) {
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  const int n_items = big[0];
  for (int it = blockIdx.x * 4 + wib; it < n_items; it += gridDim.x * 4) {
    const int p = __builtin_amdgcn_readfirstlane(big[2 + 2 * it]);
    const int base = __builtin_amdgcn_readfirstlane(big[2 + 2 * it + 1]) * GD_WAVE;
    const int off = __builtin_amdgcn_readfirstlane(pt_off[p]);
    const int cnt = __builtin_amdgcn_readfirstlane(pt_off[p + 1]) - off;
    const int own = base + lane < cnt ? csr_raw[off + base + lane] : 0x7fffffff;
    int rk = 0;
    for (int cb = 0; cb < cnt; cb += GD_WAVE) {
      const int u = cb + lane < cnt ? csr_raw[off + cb + lane] : 0x7fffffff;
      const int lim = cnt - cb < GD_WAVE ? cnt - cb : GD_WAVE;
      for (int j = 0; j < lim; ++j) rk += (__builtin_amdgcn_readlane(u, j) < own) ? 1 : 0;
    }
    if (base + lane < cnt) {
      csr[off + rk] = own;
      rank[own] = rk;
      if (pts_pm) {
        const float* r = pts_out + (long long)own * ncols;
        float* w = pts_pm + (long long)(off + rk) * ncols;
        for (int c = 0; c < ncols; ++c) w[c] = r[c];
        row_pillar[off + rk] = p;
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_big_mean(const int* __restrict__ big, int big_cap, const int* __restrict__ pt_off,
                                                  const int* __restrict__ csr, const float* __restrict__ pts_out, int ncols,
                                                  float* __restrict__ mean) {
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  const int n_big = big[1];
  const int F = ncols - 1;
  for (int b = blockIdx.x * 4 + wib; b < n_big; b += gridDim.x * 4) {
    const int p = __builtin_amdgcn_readfirstlane(big[2 + 2 * big_cap + b]);
    const int off = __builtin_amdgcn_readfirstlane(pt_off[p]);
    const int cnt = __builtin_amdgcn_readfirstlane(pt_off[p + 1]) - off;
    pillar_mean(pts_out, ncols, F, csr, off, cnt, 0, lane, mean + (long long)p * F);
  }
}

extern "C" size_t gdmae_voxelize_workspace_bytes(long long n_points, int batch_size, int gx, int gy, int gz) {
  long long cells = (long long)batch_size * gx * gy * gz;
  size_t b = 0;
  b += gd_align(sizeof(int) * cells);                               // cell_cnt (zeroed at entry)
  b += gd_align(sizeof(int) * cells);                               // cell2pillar
  b += gd_align(sizeof(int) * n_points) * 5;                        // key, pos, csr_raw, slot, raw_pillar
  b += gd_align(sizeof(unsigned long long) * gd_scan_ws_elems(cells > n_points ? cells : n_points));
  b += gd_align(sizeof(unsigned long long) * 2) + gd_align(sizeof(int) * 2);
  b += gd_align(sizeof(int) * (2 + 3 * (n_points / 32 + 2)));          // big-pillar work items
  return b + 4096;
}

// Plan path: rank of every point inside its pillar by ONE THREAD PER POINT over the raw (arrival-order) CSR - the thread counts
// the ids of its pillar's segment that are smaller than its own (the lanes of a pillar read the same addresses: broadcast loads
// from L1).  Any pillar size in one kernel: a 1000-point wall pillar is 1000 threads x 1000 loads instead of the serial
// tail of k_big_rank, and the 10-point average pillar no longer idles 54 lanes of a wavefront.  Also writes the pillar-major
// row of the point (gdmae_pillar_major_rows).
__global__ __launch_bounds__(256) void k_rank_points(const int* __restrict__ counts, const int* __restrict__ csr_raw,
                                                     const int* __restrict__ raw_pillar, const int* __restrict__ pt_off,
                                                     int* __restrict__ csr, int* __restrict__ rank, const float* __restrict__ pts_out,
                                                     int ncols, float* __restrict__ pts_pm, int* __restrict__ row_pillar) {
  const int N = counts[0];
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < N; q += gridDim.x * blockDim.x) {
    const int own = csr_raw[q];
    const int p = raw_pillar[q];                  // written next to csr_raw by the fill pass: both loads are coalesced
    const int off = pt_off[p], cnt = pt_off[p + 1] - off;
    const int* seg = csr_raw + off;
    int rk = 0;
    int j = 0;
    for (; j + 16 <= cnt; j += 16) {              // 16 independent (broadcast) loads in flight: a 1000-point pillar is 62 round trips
      int u[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) u[e] = seg[j + e];
#pragma unroll
      for (int e = 0; e < 16; ++e) rk += u[e] < own;
    }
    for (; j + 4 <= cnt; j += 4) {
      const int a = seg[j], b = seg[j + 1], c = seg[j + 2], d = seg[j + 3];
      rk += (a < own) + (b < own) + (c < own) + (d < own);
    }
    for (; j < cnt; ++j) rk += seg[j] < own;
    csr[off + rk] = own;
    rank[own] = rk;
    if (pts_pm) {
      const float* r = pts_out + (long long)own * ncols;
      float* w = pts_pm + (long long)(off + rk) * ncols;
      for (int c = 0; c < ncols; ++c) w[c] = r[c];
      row_pillar[off + rk] = p;
    }
  }
}
// Sequential (canonical ascending-id order, bit-identical to a CPU index_add_) feature sum of a pillar by one thread per
// (pillar, channel) over the pillar-major rows: consecutive rows, independent loads, one dependent add chain.
__global__ __launch_bounds__(256) void k_pillar_mean_rows(const int* __restrict__ counts, const int* __restrict__ pt_off,
                                                          const float* __restrict__ pts_pm, int ncols, float* __restrict__ mean) {
  const int F = ncols - 1;
  const long long total = (long long)counts[1] * F;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(e / F), c = (int)(e % F);
    const int off = pt_off[p], cnt = pt_off[p + 1] - off;
    const float* r = pts_pm + (long long)off * ncols + 1 + c;
    float acc = 0.f;
    int j = 0;
    for (; j + 16 <= cnt; j += 16) {              // the loads are independent of the add chain: 16 in flight per trip
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = r[(long long)(j + e) * ncols];
#pragma unroll
      for (int e = 0; e < 16; ++e) acc = __fadd_rn(acc, v[e]);
    }
    for (; j + 4 <= cnt; j += 4) {
      const float v0 = r[(long long)j * ncols], v1 = r[(long long)(j + 1) * ncols], v2 = r[(long long)(j + 2) * ncols],
                  v3 = r[(long long)(j + 3) * ncols];
      acc = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc, v0), v1), v2), v3);
    }
    for (; j < cnt; ++j) acc = __fadd_rn(acc, r[(long long)j * ncols]);
    mean[e] = __fdiv_rn(acc, (float)cnt);
  }
}

// the one-thread finalize of the classic path, as the grand-total hooks of the two single-launch scans
struct KeepTotal {
  int* counts;
  __device__ void operator()(int n_keep) const { counts[0] = n_keep; }
};
struct CellTotal {
  int B;
  int* pt_off;
  int* sample_off;
  int* counts;
  __device__ void operator()(unsigned long long t) const {
    const int M = (int)(t >> 32), N = (int)(unsigned)t;
    pt_off[M] = N;
    sample_off[B] = M;
    counts[1] = M;
  }
};

__global__ __launch_bounds__(GD_SCAN_BLOCK) void k_vox_scans(long long n_points, KeepLoad kl, KeepStore ks, KeepTotal kt, GdScanState S0,
                                                             long long cells, CellLoad cl, CellStore cs, CellTotal ct, GdScanState S1) {
  if (blockIdx.y == 0) {
    if ((int)blockIdx.x >= S0.nb) return;
    gd_scan_lb_tile<int>(n_points, kl, ks, kt, (int*)nullptr, S0.ticket, S0.flags, S0.agg, S0.incl, (unsigned)S0.nb);
  } else {
    if ((int)blockIdx.x >= S1.nb) return;
    gd_scan_lb_tile<unsigned long long>(cells, cl, cs, ct, (unsigned long long*)nullptr, S1.ticket, S1.flags, S1.agg, S1.incl, (unsigned)S1.nb);
  }
}

// Body of gdmae_voxelize.  lb_state == null: the classic schedule (three-launch scans, finalize launch, no pillar-major rows).
// lb_state != null (geometry plan, plan.hip): 2 * gd_scan_lb_state_bytes-sized ZEROED scan states -> single-launch scans with the
// finalize folded in; points_pm / row_pillar (optional) are written by the ranking kernels instead of a pass of their own.
int gd_voxelize_impl(const float* points, long long n_points, int n_cols, const float* lo, const float* vs, const int* grid_xyz,
                     int batch_size, float* points_out, long long* point_coords, long long* inverse, int* inverse32,
                     long long* voxel_coords, int* pillar_cell, int* pillar_pt_off, int* pillar_pts, int* point_rank,
                     int* sample_pillar_off, float* pillar_mean, int* cell2pillar_out, int* counts, void* workspace, size_t workspace_bytes,
                     void* lb_state, float* points_pm, int* row_pillar, hipStream_t st) {
  GD_REQUIRE(n_cols >= 4 && n_cols <= 65, "n_cols must be 1+F with 3 <= F <= 64");
  VoxParams P;
  for (int i = 0; i < 3; ++i) {
    P.lo[i] = lo[i];
    P.vs[i] = vs[i];
  }
  P.gx = grid_xyz[0];
  P.gy = grid_xyz[1];
  P.gz = grid_xyz[2];
  P.B = batch_size;
  P.ncols = n_cols;
  const long long cells = (long long)batch_size * P.gx * P.gy * P.gz;
  GD_REQUIRE(cells > 0 && cells < (1ll << 31), "B*Z*Y*X must fit int32");
  GD_REQUIRE(n_points < (1ll << 31), "too many points");
  GD_REQUIRE(workspace_bytes >= gdmae_voxelize_workspace_bytes(n_points, batch_size, P.gx, P.gy, P.gz),
             "voxelize workspace too small");
  GdArena A(workspace, workspace_bytes);
  int* cell_cnt = A.take<int>(cells);
  int* cell2pillar = cell2pillar_out ? cell2pillar_out : A.take<int>(cells);
  int* key = A.take<int>(n_points);
  int* pos = A.take<int>(n_points);
  int* csr_raw = A.take<int>(n_points);
  int* slot = A.take<int>(n_points);
  int* raw_pillar_ws = A.take<int>(n_points);
  unsigned long long* scan_ws = A.take<unsigned long long>(gd_scan_ws_elems(cells > n_points ? cells : n_points));
  unsigned long long* total = A.take<unsigned long long>(2);
  int* n_keep = A.take<int>(2);
  const int big_cap = (int)(n_points / 32 + 2);
  int* big = A.take<int>(2 + 3 * (size_t)big_cap);

  if (lb_state) {
    // zeroed by the XCD that will own the counters (frame-major eighths, as VOX_FOR_POINTS deals the points): the atomics of
    // k_point_keys then find their lines in the local L2 instead of pulling them out of another XCD's
    hipLaunchKernelGGL(k_zero_cells, dim3(2048), dim3(256), 0, st, cell_cnt, cells, big);
    GD_LAUNCH_CHECK();
  } else {
    GD_CHECK(hipMemsetAsync(cell_cnt, 0, sizeof(int) * cells, st));
    GD_CHECK(hipMemsetAsync(big, 0, sizeof(int) * 2, st));
  }
  int grid_pts = n_points > 0 ? (gd_div_up(n_points, 256) < 4096 ? gd_div_up(n_points, 256) : 4096) : 8;
  grid_pts = (grid_pts + 7) / 8 * 8;              // VOX_FOR_POINTS: a multiple of 8 workgroups
  if (n_points > 0) {
    hipLaunchKernelGGL(k_point_keys, dim3(grid_pts), dim3(256), 0, st, points, n_points, P, key, slot, cell_cnt);
    GD_LAUNCH_CHECK();
  }
  CellStore cs{P, cell2pillar, pillar_pt_off, pillar_cell, voxel_coords, sample_pillar_off, (lb_state && points_pm) ? nullptr : big, big_cap};
  if (lb_state) {
    // the kept-point scan and the cell scan only need k_point_keys: one launch (blockIdx.y picks the scan)
    char* s0 = (char*)lb_state;
    char* s1 = s0 + gd_scan_lb_state_bytes<int>(n_points);
    const GdScanState S0 = gd_scan_lb_state<int>(n_points, s0), S1 = gd_scan_lb_state<unsigned long long>(cells, s1);
    hipLaunchKernelGGL(k_vox_scans, dim3(S0.nb > S1.nb ? S0.nb : S1.nb, 2), dim3(GD_SCAN_BLOCK), 0, st, n_points, KeepLoad{key}, KeepStore{pos},
                       KeepTotal{counts}, S0, cells, CellLoad{cell_cnt}, cs, CellTotal{batch_size, pillar_pt_off, sample_pillar_off, counts}, S1);
    GD_LAUNCH_CHECK();
  } else {
    int rc = gd_device_scan<int>(n_points, KeepLoad{key}, KeepStore{pos}, n_keep, (int*)scan_ws, st);
    if (rc) return rc;
    rc = gd_device_scan<unsigned long long>(cells, CellLoad{cell_cnt}, cs, total, scan_ws, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_vox_finalize, dim3(1), dim3(64), 0, st, total, n_keep, batch_size, pillar_pt_off,
                       sample_pillar_off, counts);
    GD_LAUNCH_CHECK();
  }
  if (n_points > 0) {
    int* raw_pillar = (lb_state && points_pm) ? raw_pillar_ws : nullptr;
    hipLaunchKernelGGL(k_point_fill, dim3(grid_pts), dim3(256), 0, st, points, n_points, P, key, pos, cell2pillar,
                       pillar_pt_off, slot, points_out, point_coords, inverse, inverse32, csr_raw, raw_pillar);
    GD_LAUNCH_CHECK();
    if (lb_state && points_pm) {
      // plan path: one thread per point ranks it (any pillar size) and writes its pillar-major row; the means walk those rows
      hipLaunchKernelGGL(k_rank_points, dim3(grid_pts), dim3(256), 0, st, counts, csr_raw, (const int*)raw_pillar_ws, pillar_pt_off, pillar_pts, point_rank,
                         points_out, n_cols, points_pm, row_pillar);
      GD_LAUNCH_CHECK();
      hipLaunchKernelGGL(k_pillar_mean_rows, dim3(grid_pts), dim3(256), 0, st, counts, pillar_pt_off, points_pm, n_cols, pillar_mean);
      GD_LAUNCH_CHECK();
      return 0;
    }
    hipLaunchKernelGGL(k_pillar_sort_mean, dim3(4096), dim3(256), 0, st, counts, pillar_pt_off, csr_raw, pillar_pts,
                       point_rank, points_out, n_cols, pillar_mean, points_pm, row_pillar);
    GD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_big_rank, dim3(2048), dim3(256), 0, st, big, pillar_pt_off, csr_raw, pillar_pts, point_rank, (const float*)points_out,
                       n_cols, points_pm, row_pillar);
    GD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_big_mean, dim3(1024), dim3(256), 0, st, big, big_cap, pillar_pt_off, pillar_pts, points_out, n_cols,
                       pillar_mean);
    GD_LAUNCH_CHECK();
  }
  return 0;
}
size_t gd_voxelize_lb_state_bytes(long long n_points, long long cells) {
  return gd_scan_lb_state_bytes<int>(n_points) + gd_scan_lb_state_bytes<unsigned long long>(cells);
}

// See include/gdmae_hip.h for the contract.
extern "C" int gdmae_voxelize(const float* points, long long n_points, int n_cols, const float* lo, const float* vs,
                              const int* grid_xyz, int batch_size, float* points_out, long long* point_coords,
                              long long* inverse, int* inverse32, long long* voxel_coords, int* pillar_cell,
                              int* pillar_pt_off, int* pillar_pts, int* point_rank, int* sample_pillar_off,
                              float* pillar_mean, int* cell2pillar_out, int* counts, void* workspace,
                              size_t workspace_bytes, void* stream) {
  return gd_voxelize_impl(points, n_points, n_cols, lo, vs, grid_xyz, batch_size, points_out, point_coords, inverse, inverse32,
                          voxel_coords, pillar_cell, pillar_pt_off, pillar_pts, point_rank, sample_pillar_off, pillar_mean,
                          cell2pillar_out, counts, workspace, workspace_bytes, nullptr, nullptr, nullptr, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// Drop-in equivalents of the reference's own native op API for this path (pybind module
// pcdet.ops.sst_ops.sst_ops_cuda, pcdet/ops/sst_ops/src/sst_ops_api.cpp:6-9), for arbitrary group ids:
//   ingroup_inds_wrapper(group_inds, out_inds)        -> gdmae_ingroup_inds
//   group_inner_inds_wrapper(inverse_inds, group_inds) -> gdmae_group_inner_inds
// Same outputs as sst_ops_gpu.cu:14-39 evaluated sequentially (canonical ascending-index order instead
// of atomic arrival order), the counter scratch is caller provided, launches go to the caller's stream,
// and failures are returned instead of exit(-1).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_group_count(const long long* __restrict__ gid, long long n, long long n_groups,
                                                     int* __restrict__ cnt, int* __restrict__ bad) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    long long g = gid[i];
    if (g < 0 || g >= n_groups) {
      *bad = 1;
      continue;
    }
    atomicAdd(&cnt[g], 1);
  }
}
struct CntLoad {
  const int* cnt;
  __device__ int operator()(long long g) const { return cnt[g]; }
};
struct OffStore {
  int* off;
  __device__ void operator()(long long g, int ex, int) const { off[g] = ex; }
};
__global__ void k_group_finalize(const int* total, long long n_groups, int* off, int* counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    off[n_groups] = *total;
    counts[0] = *total;
    counts[1] = (int)n_groups;
  }
}
__global__ __launch_bounds__(256) void k_group_fill(const long long* __restrict__ gid, long long n, long long n_groups,
                                                    const int* __restrict__ off, int* __restrict__ cnt,
                                                    int* __restrict__ csr_raw) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    long long g = gid[i];
    if (g < 0 || g >= n_groups) continue;
    int slot = atomicSub(&cnt[g], 1) - 1;
    csr_raw[off[g] + slot] = (int)i;
  }
}
__global__ __launch_bounds__(256) void k_rank_to_i64(const int* __restrict__ rank, long long n, long long* __restrict__ out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    out[i] = rank[i];
}
__global__ __launch_bounds__(256) void k_group_inner(const int* __restrict__ off, const int* __restrict__ csr, long long M,
                                                     int K, long long* __restrict__ group_inds) {
  const long long total = M * K;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
    const long long g = e / K;
    const int k = (int)(e % K);
    const int o = off[g], cnt = off[g + 1] - o;
    group_inds[e] = cnt == 0 ? -1 : (long long)csr[o + (k < cnt ? k : k % cnt)];
  }
}

extern "C" size_t gdmae_group_workspace_bytes(long long n, long long n_groups) {
  return gd_align(sizeof(int) * (n_groups + 1)) * 2 + gd_align(sizeof(int) * n) * 3 +
         gd_align(sizeof(int) * (gd_scan_ws_elems(n_groups) + 8)) + gd_align(sizeof(int) * (2 + 2 * (n / 32 + 2))) + 4096;
}

__global__ __launch_bounds__(256) void k_group_big_items(const int* __restrict__ off, long long n_groups, int* __restrict__ big) {
  for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < n_groups; g += (long long)gridDim.x * blockDim.x) {
    const int cnt = off[g + 1] - off[g];
    if (cnt > GD_WAVE) {
      const int nch = (cnt + GD_WAVE - 1) / GD_WAVE;
      const int base = atomicAdd(&big[0], nch);
      for (int k = 0; k < nch; ++k) {
        big[2 + 2 * (base + k)] = (int)g;
        big[2 + 2 * (base + k) + 1] = k;
      }
    }
  }
}

static int gd_group_csr(const long long* gid, long long n, long long n_groups, GdArena& A, int** off_out, int** csr_out,
                        int** rank_out, hipStream_t st) {
  GD_REQUIRE(n < (1ll << 31) && n_groups < (1ll << 31) && n_groups > 0, "sizes must fit int32");
  int* cnt = A.take<int>(n_groups + 1);
  int* off = A.take<int>(n_groups + 1);
  int* csr_raw = A.take<int>(n);
  int* csr = A.take<int>(n);
  int* rank = A.take<int>(n);
  int* scan_ws = A.take<int>(gd_scan_ws_elems(n_groups) + 8);
  int* total = scan_ws + gd_scan_ws_elems(n_groups);
  int* counts = total + 2;
  int* bad = total + 4;
  int* big = A.take<int>(2 + 2 * (size_t)(n / 32 + 2));
  GD_REQUIRE(A.ok(), "group workspace too small");
  GD_CHECK(hipMemsetAsync(big, 0, sizeof(int) * 2, st));
  GD_CHECK(hipMemsetAsync(cnt, 0, sizeof(int) * (n_groups + 1), st));
  GD_CHECK(hipMemsetAsync(bad, 0, sizeof(int), st));
  int grid = gd_div_up(n > 0 ? n : 1, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_group_count, dim3(grid), dim3(256), 0, st, gid, n, n_groups, cnt, bad);
  GD_LAUNCH_CHECK();
  int rc = gd_device_scan<int>(n_groups, CntLoad{cnt}, OffStore{off}, total, scan_ws, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_group_finalize, dim3(1), dim3(64), 0, st, total, n_groups, off, counts);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_group_fill, dim3(grid), dim3(256), 0, st, gid, n, n_groups, off, cnt, csr_raw);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_pillar_sort_mean, dim3(2048), dim3(256), 0, st, counts, off, csr_raw, csr, rank,
                     (const float*)nullptr, 1, (float*)nullptr, (float*)nullptr, (int*)nullptr);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_group_big_items, dim3(gd_div_up(n_groups, 256) < 2048 ? gd_div_up(n_groups, 256) : 2048), dim3(256), 0,
                     st, off, n_groups, big);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_big_rank, dim3(2048), dim3(256), 0, st, big, off, csr_raw, csr, rank, (const float*)nullptr, 1, (float*)nullptr,
                     (int*)nullptr);
  GD_LAUNCH_CHECK();
  *off_out = off;
  *csr_out = csr;
  *rank_out = rank;
  return 0;
}

extern "C" int gdmae_ingroup_inds(const long long* group_inds, long long n, long long n_groups, long long* out_inds,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) return 0;
  GD_REQUIRE(workspace_bytes >= gdmae_group_workspace_bytes(n, n_groups), "group workspace too small");
  GdArena A(workspace, workspace_bytes);
  int *off, *csr, *rank;
  int rc = gd_group_csr(group_inds, n, n_groups, A, &off, &csr, &rank, st);
  if (rc) return rc;
  int grid = gd_div_up(n, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_rank_to_i64, dim3(grid), dim3(256), 0, st, rank, n, out_inds);
  GD_LAUNCH_CHECK();
  return 0;
}

extern "C" int gdmae_group_inner_inds(const long long* inverse_inds, long long n, long long M, int K,
                                      long long* group_inds, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GD_REQUIRE(M > 0 && K > 0, "M, K");
  GD_REQUIRE(workspace_bytes >= gdmae_group_workspace_bytes(n, M), "group workspace too small");
  GdArena A(workspace, workspace_bytes);
  int *off, *csr, *rank;
  int rc = gd_group_csr(inverse_inds, n, M, A, &off, &csr, &rank, st);
  if (rc) return rc;
  int grid = gd_div_up(M * K, 256);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_group_inner, dim3(grid), dim3(256), 0, st, off, csr, M, K, group_inds);
  GD_LAUNCH_CHECK();
  return 0;
}
