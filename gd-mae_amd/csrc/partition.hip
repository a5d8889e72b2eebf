// Geometry plan for the sparse regional transformer: token sets per stage, sparse-conv rulebooks and
// the shifted-window partition, all as direct-address lookups on dense BEV occupancy maps.
//
// Replaces (SURVEY.md §8 rows a6 index side, a7-a10):
//   spconv rulebook build (hash table) for SparseConv2d k3 s2 p1 / SubMConv2d k3
//                                 reference call sites pcdet/utils/spconv_utils.py:41-43
//   get_window_coors              pcdet/models/model_utils/sst_utils.py:6-47
//   get_inner_win_inds            pcdet/ops/sst_ops/src/sst_ops_gpu.cu:14-20 (atomic arrival order)
//   drop_single_shift/bincount    pcdet/models/backbones_3d/spt_backbone.py:32-51
//   make_continuous_inds, get_flat2win_inds   sst_utils.py:50-104 (unique + sort + .item() per level)
//
// MI355X design: a stage's active set lives as (a) an ascending list of linear cell keys and (b) a
// dense int32 map cell -> token id (-1 = empty) that stays L2/Infinity-Cache resident (1.75 M cells for
// 8 Waymo frames).  An 8x8 window is exactly one 64-lane wavefront: lane l looks up cell (l/8, l%8) of
// the window, a single ballot gives the window population and popcount(ballot & lanes_below) is the
// canonical rank of the lane's token (tokens are ordered by (y,x), which is lane order) - no atomics,
// no sort, deterministic.  Dense window indices per occupancy level and the token CSR come from one
// packed scan over the window grid.  Nothing here syncs with the host; counts stay in `counts`.
#include "common.h"

struct Dims {
  int B, Y, X;
};

// ------------------------------------------------------------------------------------------
// stage-1 tokens = visible pillars
// ------------------------------------------------------------------------------------------
struct VisLoad {
  const float* mask;
  const int* counts;  // counts[1] = M
  __device__ int operator()(long long p) const { return (p < counts[1] && mask[p] == 0.f) ? 1 : 0; }
};
struct VisStore {
  const int* counts;
  const int* pillar_cell;
  int* tok_pillar;
  int* tok_cell;
  int* map;
  __device__ void operator()(long long p, int ex, int v) const {
    if (v) {
      tok_pillar[ex] = (int)p;
      int c = pillar_cell[p];
      tok_cell[ex] = c;
      map[c] = ex;
    }
  }
};

extern "C" int gdmae_visible_tokens(const float* mask, const int* pillar_cell, const int* vox_counts, long long m_cap,
                                    long long n_cells, int* tok_pillar, int* tok_cell, int* map, int* n_tok,
                                    void* scan_ws, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GD_CHECK(hipMemsetAsync(map, 0xFF, sizeof(int) * n_cells, st));
  return gd_device_scan<int>(m_cap, VisLoad{mask, vox_counts}, VisStore{vox_counts, pillar_cell, tok_pillar, tok_cell, map},
                             n_tok, (int*)scan_ws, st);
}

// all pillars as tokens (fine-tune path / no masking): map = cell2pillar
// ------------------------------------------------------------------------------------------
// strided conv k3 s2 p1: output active set
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mark_down(const int* __restrict__ n_in, const int* __restrict__ tok_cell,
                                                   Dims di, Dims dn, int* __restrict__ flag) {
  const int M = *n_in;
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < M; t += gridDim.x * blockDim.x) {
    int c = tok_cell[t];
    int x = c % di.X;
    int r = c / di.X;
    int y = r % di.Y;
    int b = r / di.Y;
    // input i feeds output o through tap k iff i = 2*o - 1 + k  ->  o = (i + 1 - k) / 2
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      int ty = y + 1 - ky;
      if (ty < 0 || (ty & 1)) continue;
      int oy = ty >> 1;
      if (oy >= dn.Y) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        int tx = x + 1 - kx;
        if (tx < 0 || (tx & 1)) continue;
        int ox = tx >> 1;
        if (ox >= dn.X) continue;
        flag[(b * dn.Y + oy) * dn.X + ox] = 1;
      }
    }
  }
}

struct FlagLoad {
  const int* flag;
  __device__ int operator()(long long c) const { return flag[c]; }
};
struct FlagStore {
  int* tok_cell;
  int* map;
  __device__ void operator()(long long c, int ex, int v) const {
    if (v) {
      tok_cell[ex] = (int)c;
      map[c] = ex;
    } else {
      map[c] = -1;
    }
  }
};

extern "C" int gdmae_downsample_tokens(const int* n_in, const int* tok_cell_in, long long cap_in, int B, int Yi, int Xi,
                                       int* tok_cell_out, int* map_out, int* n_out, int* flag_ws, void* scan_ws,
                                       void* stream) {
  hipStream_t st = (hipStream_t)stream;
  Dims di{B, Yi, Xi};
  Dims dn{B, (Yi + 2 - 3) / 2 + 1, (Xi + 2 - 3) / 2 + 1};
  const long long cells = (long long)B * dn.Y * dn.X;
  GD_CHECK(hipMemsetAsync(flag_ws, 0, sizeof(int) * cells, st));
  int grid = gd_div_up(cap_in > 0 ? cap_in : 1, 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_mark_down, dim3(grid), dim3(256), 0, st, n_in, tok_cell_in, di, dn, flag_ws);
  GD_LAUNCH_CHECK();
  return gd_device_scan<int>(cells, FlagLoad{flag_ws}, FlagStore{tok_cell_out, map_out}, n_out, (int*)scan_ws, st);
}

// ------------------------------------------------------------------------------------------
// rulebooks: nbr[t*9 + k] = input token feeding output token t through tap k (ky*3+kx), or -1
//   mode 0: submanifold k3           in = p + k - 1                (same token set)
//   mode 1: strided k3 s2 p1 forward in = 2*o - 1 + k              (map = input-resolution map)
//   mode 2: strided transposed       out o = (i + 1 - k)/2         (map = output-resolution map; used by
//                                                                    the backward: din[i] = sum_k W_k^T dout[o])
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rulebook(const int* __restrict__ n_tok, const int* __restrict__ tok_cell, Dims dt,
                                                  Dims dm, const int* __restrict__ map, int mode, int* __restrict__ nbr) {
  const long long total = (long long)(*n_tok) * 9;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i / 9), k = (int)(i % 9);
    const int ky = k / 3, kx = k % 3;
    int c = tok_cell[t];
    int x = c % dt.X;
    int r = c / dt.X;
    int y = r % dt.Y;
    int b = r / dt.Y;
    int my, mx;
    bool ok = true;
    if (mode == 0) {
      my = y + ky - 1;
      mx = x + kx - 1;
    } else if (mode == 1) {
      my = 2 * y - 1 + ky;
      mx = 2 * x - 1 + kx;
    } else {
      int ty = y + 1 - ky, tx = x + 1 - kx;
      ok = !(ty & 1) && !(tx & 1) && ty >= 0 && tx >= 0;
      my = ty >> 1;
      mx = tx >> 1;
    }
    ok = ok && my >= 0 && my < dm.Y && mx >= 0 && mx < dm.X;
    nbr[i] = ok ? map[(b * dm.Y + my) * dm.X + mx] : -1;
  }
}

extern "C" int gdmae_rulebook(const int* n_tok, const int* tok_cell, long long cap, int B, int Yt, int Xt, int Ym,
                              int Xm, const int* map, int mode, int* nbr, void* stream) {
  GD_REQUIRE(mode >= 0 && mode <= 2, "rulebook mode");
  int grid = gd_div_up((cap > 0 ? cap : 1) * 9, 256);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_rulebook, dim3(grid), dim3(256), 0, (hipStream_t)stream, n_tok, tok_cell, Dims{B, Yt, Xt},
                     Dims{B, Ym, Xm}, map, mode, nbr);
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------
// shifted-window partition
// ------------------------------------------------------------------------------------------
struct WinParams {
  int B, Y, X;       // token grid
  int wx, wy;        // window shape (wx*wy <= 64)
  int sx, sy;        // offsets added before the division (win for the un-shifted pass, win/2 for the shifted)
  int nwx, nwy, nwz; // window grid per sample (ceil(g/w)+1)
  int nlev;
  int lo[3], hi[3], T[3];
};

__device__ inline int win_token(const WinParams& P, const int* map, int w, int lane, int& ref_id) {
  const int wyi = w % P.nwy;
  const int r = w / P.nwy;
  const int wxi = r % P.nwx;
  const int b = r / P.nwx;
  ref_id = b * (P.nwx * P.nwy * P.nwz) + wxi * (P.nwy * P.nwz) + wyi * P.nwz;
  if (lane >= P.wx * P.wy) return -1;
  const int ly = lane / P.wx, lx = lane % P.wx;
  const int x = wxi * P.wx - P.sx + lx;
  const int y = wyi * P.wy - P.sy + ly;
  if (x < 0 || x >= P.X || y < 0 || y >= P.Y) return -1;
  return map[(b * P.Y + y) * P.X + x];
}

__global__ __launch_bounds__(256) void k_win_count(WinParams P, const int* __restrict__ map, int n_win,
                                                   int* __restrict__ win_cnt) {
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  for (int w = blockIdx.x * 4 + wib; w < n_win; w += gridDim.x * 4) {
    int ref;
    int t = win_token(P, map, w, lane, ref);
    unsigned long long m = __ballot(t >= 0);
    if (lane == 0) win_cnt[w] = __popcll(m);
  }
}

__device__ inline int win_level(const WinParams& P, int cnt) {
  for (int l = 0; l < P.nlev; ++l)
    if (cnt >= P.lo[l] && cnt < P.hi[l]) return l;
  return -1;
}

struct WinLoad {
  WinParams P;
  const int* win_cnt;
  __device__ U128 operator()(long long w) const {
    int c = win_cnt[w];
    if (c <= 0) return U128{0ull, 0ull};
    int l = win_level(P, c);
    if (l < 0) return U128{0ull, 0ull};
    return U128{1ull << (21 * l), (unsigned long long)c << (21 * l)};
  }
};
struct WinStore {
  int* win_dense;   // dense index of the window inside its level
  int* win_tokpre;  // token prefix inside its level
  __device__ void operator()(long long w, U128 ex, U128 v) const {
    if (v.a) {
      int l = (v.a >> 21) ? ((v.a >> 42) ? 2 : 1) : 0;
      win_dense[w] = (int)((ex.a >> (21 * l)) & 0x1FFFFFull);
      win_tokpre[w] = (int)((ex.b >> (21 * l)) & 0x1FFFFFull);
    }
  }
};

// counts layout (int32[8]): [0..2] windows per level, [3..5] tokens per level, [6] total windows, [7] total tokens
__global__ void k_win_totals(const U128* total, int* counts) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int nw = 0, nt = 0;
    for (int l = 0; l < 3; ++l) {
      int a = (int)((total->a >> (21 * l)) & 0x1FFFFFull);
      int b = (int)((total->b >> (21 * l)) & 0x1FFFFFull);
      counts[l] = a;
      counts[3 + l] = b;
      nw += a;
      nt += b;
    }
    counts[6] = nw;
    counts[7] = nt;
  }
}

__global__ __launch_bounds__(256) void k_win_fill(WinParams P, const int* __restrict__ map, int n_win,
                                                  const int* __restrict__ win_cnt, const int* __restrict__ win_dense,
                                                  const int* __restrict__ win_tokpre, const int* __restrict__ counts,
                                                  int* __restrict__ tok_win, int* __restrict__ tok_level,
                                                  int* __restrict__ tok_slot, int* __restrict__ tok_pos,
                                                  int* __restrict__ csr_tok, int* __restrict__ win_start,
                                                  int* __restrict__ win_len) {
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  for (int w = blockIdx.x * 4 + wib; w < n_win; w += gridDim.x * 4) {
    const int cnt = win_cnt[w];
    if (cnt <= 0) continue;
    const int l = win_level(P, cnt);
    if (l < 0) continue;
    int ref;
    const int t = win_token(P, map, w, lane, ref);
    const unsigned long long m = __ballot(t >= 0);
    int wbase = 0, tbase = 0;
    for (int q = 0; q < l; ++q) {
      wbase += counts[q];
      tbase += counts[3 + q];
    }
    const int d = win_dense[w];
    const int start = tbase + win_tokpre[w];
    if (lane == 0) {
      win_start[wbase + d] = start;
      win_len[wbase + d] = cnt;
    }
    if (t >= 0) {
      const int r = __popcll(m & ((1ull << lane) - 1ull));
      csr_tok[start + r] = t;
      tok_win[t] = ref;
      tok_level[t] = l;
      tok_slot[t] = d * P.T[l] + r;
      tok_pos[t] = lane;
    }
  }
}

extern "C" size_t gdmae_window_workspace_bytes(int B, int Y, int X, int wx, int wy) {
  long long nwx = (X + wx - 1) / wx + 1, nwy = (Y + wy - 1) / wy + 1;
  long long n = B * nwx * nwy;
  return gd_align(sizeof(int) * n) * 3 + gd_align(sizeof(U128) * (gd_scan_ws_elems(n) + 2)) + 4096;
}

// drop_lo/hi/T: per level drop_range and max_tokens of DROP_INFO (spt_backbone.py:32-51)
extern "C" int gdmae_window_partition(const int* map, int B, int Y, int X, int wx, int wy, int shifted, int nlev,
                                      const int* drop_lo, const int* drop_hi, const int* max_tokens, int* tok_win,
                                      int* tok_level, int* tok_slot, int* tok_pos, int* csr_tok, int* win_start,
                                      int* win_len, int* counts, void* workspace, size_t workspace_bytes, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  GD_REQUIRE(wx * wy <= GD_WAVE && wx > 0 && wy > 0, "window must fit one wavefront (wx*wy <= 64)");
  GD_REQUIRE(nlev >= 1 && nlev <= 3, "1..3 drop levels");
  WinParams P;
  P.B = B;
  P.Y = Y;
  P.X = X;
  P.wx = wx;
  P.wy = wy;
  P.sx = shifted ? wx / 2 : wx;   // sst_utils.py:19-22: the un-shifted pass adds a full window
  P.sy = shifted ? wy / 2 : wy;
  P.nwx = (X + wx - 1) / wx + 1;  // ceil(g/w) + 1
  P.nwy = (Y + wy - 1) / wy + 1;
  P.nwz = 2;                      // ceil(1/1) + 1 for the single-layer pillar grid
  P.nlev = nlev;
  for (int l = 0; l < 3; ++l) {
    P.lo[l] = l < nlev ? drop_lo[l] : 0;
    P.hi[l] = l < nlev ? drop_hi[l] : 0;
    P.T[l] = l < nlev ? max_tokens[l] : 0;
  }
  const long long n_win = (long long)B * P.nwx * P.nwy;
  GD_REQUIRE(n_win < (1 << 21), "window grid too large for the packed scan");
  GD_REQUIRE(workspace_bytes >= gdmae_window_workspace_bytes(B, Y, X, wx, wy), "window workspace too small");
  GdArena A(workspace, workspace_bytes);
  int* win_cnt = A.take<int>(n_win);
  int* win_dense = A.take<int>(n_win);
  int* win_tokpre = A.take<int>(n_win);
  U128* scan_ws = A.take<U128>(gd_scan_ws_elems(n_win) + 2);
  U128* total = scan_ws + gd_scan_ws_elems(n_win);
  int grid = gd_div_up(n_win, 4);
  if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(k_win_count, dim3(grid), dim3(256), 0, st, P, map, (int)n_win, win_cnt);
  GD_LAUNCH_CHECK();
  int rc = gd_device_scan<U128>(n_win, WinLoad{P, win_cnt}, WinStore{win_dense, win_tokpre}, total, scan_ws, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_win_totals, dim3(1), dim3(64), 0, st, total, counts);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_win_fill, dim3(grid), dim3(256), 0, st, P, map, (int)n_win, win_cnt, win_dense, win_tokpre,
                     counts, tok_win, tok_level, tok_slot, tok_pos, csr_tok, win_start, win_len);
  GD_LAUNCH_CHECK();
  return 0;
}


// ==========================================================================================
// Geometry-plan variants (plan.hip): the same index structures with single-launch scans (gd_device_scan_lb, ZEROED states
// provided by the caller) and fewer, merged launches.  Outputs are bit-identical to the entry points above.
// ==========================================================================================

// stage-1 tokens by a scan over the CELLS (pillars are in ascending cell order, so the token order is the same): the map is
// written for every cell - no memset - and the pillar list is not touched
struct VisCellLoad {
  const float* mask;
  const int* cell2pillar;
  __device__ int operator()(long long c) const {
    const int p = cell2pillar[c];
    const float m = mask[p < 0 ? 0 : p];          // unconditional: the loads of a thread's 16 cells go out together
    return (int)(p >= 0) & (int)(m == 0.f);
  }
};
struct VisCellStore {
  const int* cell2pillar;
  int* tok_pillar;
  int* tok_cell;
  int* map;
  __device__ void operator()(long long c, int ex, int v) const {
    map[c] = v ? ex : -1;
    if (v) {
      tok_pillar[ex] = cell2pillar[c];
      tok_cell[ex] = (int)c;
    }
  }
};
size_t gd_plan_scan_state_bytes(long long n) { return gd_scan_lb_state_bytes<int>(n); }
size_t gd_plan_win_state_bytes(long long n) { return gd_scan_lb_state_bytes<U128>(n); }

int gd_plan_visible_tokens(const float* mask, const int* cell2pillar, long long n_cells, int* tok_pillar, int* tok_cell, int* map,
                           int* n_tok, void* lb_state, hipStream_t st) {
  return gd_device_scan_lb<int>(n_cells, VisCellLoad{mask, cell2pillar}, VisCellStore{cell2pillar, tok_pillar, tok_cell, map}, GdNoTotal{},
                                n_tok, lb_state, st);
}

// strided conv k3 s2 p1 output set by a scan over the OUTPUT cells: a cell is active iff one of its 9 input taps is (no flag
// array, no marking pass)
struct DownLoad {
  const int* map_in;
  Dims di, dn;
  __device__ int operator()(long long c) const {
    const int ci = (int)c;                      // 32-bit divisions
    const int ox = ci % dn.X;
    const int r = ci / dn.X;
    const int oy = r % dn.Y, b = r / dn.Y;
    // nine independent lookups (clamped addresses, validity applied afterwards): all loads of a thread's items are in flight
    // together instead of one dependent branch per tap
    const int* m = map_in + (long long)b * di.Y * di.X;
    int any = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy - 1 + ky;
      const int cy = iy < 0 ? 0 : (iy >= di.Y ? di.Y - 1 : iy);
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox - 1 + kx;
        const int cx = ix < 0 ? 0 : (ix >= di.X ? di.X - 1 : ix);
        const int v = m[(long long)cy * di.X + cx];
        any |= (int)(v >= 0 && iy == cy && ix == cx);
      }
    }
    return any;
  }
};
int gd_plan_downsample(const int* map_in, int B, int Yi, int Xi, int* tok_cell_out, int* map_out, int* n_out, void* lb_state,
                       hipStream_t st) {
  Dims di{B, Yi, Xi};
  Dims dn{B, (Yi + 2 - 3) / 2 + 1, (Xi + 2 - 3) / 2 + 1};
  const long long cells = (long long)B * dn.Y * dn.X;
  return gd_device_scan_lb<int>(cells, DownLoad{map_in, di, dn}, FlagStore{tok_cell_out, map_out}, GdNoTotal{}, n_out, lb_state, st);
}

// all index tables of a stage in ONE launch (blockIdx.y = job): rulebooks (mode 0 / 1 / 2 of k_rulebook, optionally with the
// tap-reversed copy nbr_rev[t][8 - k] = nbr[t][k] - the transposed submanifold rulebook) and the full-resolution sites under the
// tokens of a strided stage (mode 3: out[t * s * s + dy * s + dx] = ((b * Y s + y s + dy) * X s + x s + dx))
struct PlanJob {
  const int* n_tok;
  const int* tok_cell;
  Dims dt, dm;
  const int* map;
  int mode, s;
  int* nbr;
  int* nbr_rev;
};
// a strided stage appends up to four jobs (down, down^T, subm, upsampled sites): 4 x GDMAE_PLAN_MAX_STAGES
constexpr int kPlanMaxJobs = 16;
struct PlanJobs {
  PlanJob j[kPlanMaxJobs];
  int count;
};
__global__ __launch_bounds__(256) void k_plan_jobs(PlanJobs J) {
  const PlanJob& q = J.j[blockIdx.y];
  const int n = *q.n_tok;
  if (q.mode == 3) {
    const int ss = q.s * q.s;
    const long long total = (long long)n * ss;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
      const int t = (int)(i / ss), e = (int)(i % ss);
      const int c = q.tok_cell[t];
      const int x = c % q.dt.X, r = c / q.dt.X, y = r % q.dt.Y, b = r / q.dt.Y;
      q.nbr[i] = (b * (q.dt.Y * q.s) + y * q.s + e / q.s) * (q.dt.X * q.s) + x * q.s + e % q.s;
    }
    return;
  }
  const long long total = (long long)n * 9;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(i / 9), k = (int)(i % 9);
    const int ky = k / 3, kx = k % 3;
    const int c = q.tok_cell[t];
    const int x = c % q.dt.X, r = c / q.dt.X, y = r % q.dt.Y, b = r / q.dt.Y;
    int my, mx;
    bool ok = true;
    if (q.mode == 0) {
      my = y + ky - 1;
      mx = x + kx - 1;
    } else if (q.mode == 1) {
      my = 2 * y - 1 + ky;
      mx = 2 * x - 1 + kx;
    } else {
      const int ty = y + 1 - ky, tx = x + 1 - kx;
      ok = !(ty & 1) && !(tx & 1) && ty >= 0 && tx >= 0;
      my = ty >> 1;
      mx = tx >> 1;
    }
    ok = ok && my >= 0 && my < q.dm.Y && mx >= 0 && mx < q.dm.X;
    const int v = ok ? q.map[(b * q.dm.Y + my) * q.dm.X + mx] : -1;
    q.nbr[i] = v;
    if (q.nbr_rev) q.nbr_rev[(long long)t * 9 + (8 - k)] = v;
  }
}
int gd_plan_jobs(const PlanJobs& J, long long cap_max, hipStream_t st) {
  if (J.count <= 0) return 0;
  int grid = gd_div_up((cap_max > 0 ? cap_max : 1) * 9, 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(k_plan_jobs, dim3(grid, J.count), dim3(256), 0, st, J);
  GD_LAUNCH_CHECK();
  return 0;
}

// the window partitions of ALL stages of a plan (both shifts each = up to GD_WIN_JOBS jobs, blockIdx.y = job): one count launch, one
// scan launch (totals written by each scan's last tile), one fill launch - the partition of a stage only needs that stage's cell map
constexpr int GD_WIN_JOBS = 8;
struct WinSet {
  WinParams P[GD_WIN_JOBS];
  const int* map[GD_WIN_JOBS];
  int n_win[GD_WIN_JOBS];
  int* win_cnt[GD_WIN_JOBS];
  int* win_dense[GD_WIN_JOBS];
  int* win_tokpre[GD_WIN_JOBS];
  int* counts[GD_WIN_JOBS];
  int *tok_win[GD_WIN_JOBS], *tok_level[GD_WIN_JOBS], *tok_slot[GD_WIN_JOBS], *tok_pos[GD_WIN_JOBS], *csr_tok[GD_WIN_JOBS],
      *win_start[GD_WIN_JOBS], *win_len[GD_WIN_JOBS];
};
__global__ __launch_bounds__(256) void k_win_count2(WinSet Q) {
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  const int sh = blockIdx.y;
  const int n_win = Q.n_win[sh];
  const int* __restrict__ map = Q.map[sh];
  for (int w = blockIdx.x * 4 + wib; w < n_win; w += gridDim.x * 4) {
    int ref;
    const int t = win_token(Q.P[sh], map, w, lane, ref);
    const unsigned long long m = __ballot(t >= 0);
    if (lane == 0) Q.win_cnt[sh][w] = __popcll(m);
  }
}
struct WinTotal {
  int* counts;
  __device__ void operator()(U128 total) const {
    int nw = 0, nt = 0;
    for (int l = 0; l < 3; ++l) {
      const int a = (int)((total.a >> (21 * l)) & 0x1FFFFFull);
      const int b = (int)((total.b >> (21 * l)) & 0x1FFFFFull);
      counts[l] = a;
      counts[3 + l] = b;
      nw += a;
      nt += b;
    }
    counts[6] = nw;
    counts[7] = nt;
  }
};
__global__ __launch_bounds__(256) void k_win_fill2(WinSet Q) {
  const int lane = threadIdx.x & (GD_WAVE - 1);
  const int wib = threadIdx.x / GD_WAVE;
  const int sh = blockIdx.y;
  const int n_win = Q.n_win[sh];
  const int* __restrict__ map = Q.map[sh];
  const WinParams& P = Q.P[sh];
  const int* counts = Q.counts[sh];
  for (int w = blockIdx.x * 4 + wib; w < n_win; w += gridDim.x * 4) {
    const int cnt = Q.win_cnt[sh][w];
    if (cnt <= 0) continue;
    const int l = win_level(P, cnt);
    if (l < 0) continue;
    int ref;
    const int t = win_token(P, map, w, lane, ref);
    const unsigned long long m = __ballot(t >= 0);
    int wbase = 0, tbase = 0;
    for (int q = 0; q < l; ++q) {
      wbase += counts[q];
      tbase += counts[3 + q];
    }
    const int d = Q.win_dense[sh][w];
    const int start = tbase + Q.win_tokpre[sh][w];
    if (lane == 0) {
      Q.win_start[sh][wbase + d] = start;
      Q.win_len[sh][wbase + d] = cnt;
    }
    if (t >= 0) {
      const int r = __popcll(m & ((1ull << lane) - 1ull));
      Q.csr_tok[sh][start + r] = t;
      Q.tok_win[sh][t] = ref;
      Q.tok_level[sh][t] = l;
      Q.tok_slot[sh][t] = d * P.T[l] + r;
      Q.tok_pos[sh][t] = lane;
    }
  }
}
size_t gd_plan_windows_ws_bytes(int B, int Y, int X, int wx, int wy) {
  const long long n = (long long)B * ((X + wx - 1) / wx + 1) * ((Y + wy - 1) / wy + 1);
  return 2 * 3 * gd_align(sizeof(int) * n);
}
long long gd_plan_n_windows(int B, int Y, int X, int wx, int wy) { return (long long)B * ((X + wx - 1) / wx + 1) * ((Y + wy - 1) / wy + 1); }
// per stage: out[sh][7]: tok_win, tok_level, tok_slot, tok_pos, csr_tok, win_start, win_len; counts[sh]: int[8]; lb_state: two ZEROED
// gd_plan_win_state_bytes(n_win) states; workspace: gd_plan_windows_ws_bytes
struct GdWinStage {
  const int* map;
  int B, Y, X, wx, wy, nlev;
  const int *drop_lo, *drop_hi, *max_tokens;
  int* out[2][7];
  int* counts[2];
  void* workspace;
  void* lb_state;
};
int gd_plan_windows_all(const GdWinStage* stages, int n_stages, hipStream_t st) {
  GD_REQUIRE(n_stages >= 1 && 2 * n_stages <= GD_WIN_JOBS, "window partitions: at most four stages per launch");
  WinSet Q;
  GdScanBatch<U128, WinLoad, WinStore, WinTotal, GD_WIN_JOBS> Sc;
  long long n_max = 1;
  for (int i = 0; i < n_stages; ++i) {
    const GdWinStage& g = stages[i];
    GD_REQUIRE(g.wx * g.wy <= GD_WAVE && g.wx > 0 && g.wy > 0, "window must fit one wavefront (wx*wy <= 64)");
    GD_REQUIRE(g.nlev >= 1 && g.nlev <= 3, "1..3 drop levels");
    const long long n_win = gd_plan_n_windows(g.B, g.Y, g.X, g.wx, g.wy);
    GD_REQUIRE(n_win < (1 << 21), "window grid too large for the packed scan");
    n_max = n_win > n_max ? n_win : n_max;
    GdArena A(g.workspace, gd_plan_windows_ws_bytes(g.B, g.Y, g.X, g.wx, g.wy));
    for (int sh = 0; sh < 2; ++sh) {
      const int j = 2 * i + sh;
      WinParams& P = Q.P[j];
      P.B = g.B; P.Y = g.Y; P.X = g.X; P.wx = g.wx; P.wy = g.wy;
      P.sx = sh ? g.wx / 2 : g.wx;   // sst_utils.py:19-22: the un-shifted pass adds a full window
      P.sy = sh ? g.wy / 2 : g.wy;
      P.nwx = (g.X + g.wx - 1) / g.wx + 1;
      P.nwy = (g.Y + g.wy - 1) / g.wy + 1;
      P.nwz = 2;
      P.nlev = g.nlev;
      for (int l = 0; l < 3; ++l) {
        P.lo[l] = l < g.nlev ? g.drop_lo[l] : 0;
        P.hi[l] = l < g.nlev ? g.drop_hi[l] : 0;
        P.T[l] = l < g.nlev ? g.max_tokens[l] : 0;
      }
      Q.map[j] = g.map;
      Q.n_win[j] = (int)n_win;
      Q.win_cnt[j] = A.take<int>(n_win);
      Q.win_dense[j] = A.take<int>(n_win);
      Q.win_tokpre[j] = A.take<int>(n_win);
      Q.counts[j] = g.counts[sh];
      Q.tok_win[j] = g.out[sh][0]; Q.tok_level[j] = g.out[sh][1]; Q.tok_slot[j] = g.out[sh][2]; Q.tok_pos[j] = g.out[sh][3];
      Q.csr_tok[j] = g.out[sh][4]; Q.win_start[j] = g.out[sh][5]; Q.win_len[j] = g.out[sh][6];
      Sc.n[j] = n_win;
      Sc.load[j] = WinLoad{P, Q.win_cnt[j]};
      Sc.store[j] = WinStore{Q.win_dense[j], Q.win_tokpre[j]};
      Sc.on_total[j] = WinTotal{g.counts[sh]};
      Sc.S[j] = gd_scan_lb_state<U128>(n_win, (char*)g.lb_state + sh * gd_plan_win_state_bytes(n_win));
    }
  }
  for (int j = 2 * n_stages; j < GD_WIN_JOBS; ++j) {       // unused slots: copies of job 0 (never launched)
    Q.P[j] = Q.P[0]; Q.map[j] = Q.map[0]; Q.n_win[j] = 0;
    Q.win_cnt[j] = Q.win_cnt[0]; Q.win_dense[j] = Q.win_dense[0]; Q.win_tokpre[j] = Q.win_tokpre[0]; Q.counts[j] = Q.counts[0];
    Q.tok_win[j] = Q.tok_win[0]; Q.tok_level[j] = Q.tok_level[0]; Q.tok_slot[j] = Q.tok_slot[0]; Q.tok_pos[j] = Q.tok_pos[0];
    Q.csr_tok[j] = Q.csr_tok[0]; Q.win_start[j] = Q.win_start[0]; Q.win_len[j] = Q.win_len[0];
    Sc.n[j] = 0; Sc.load[j] = Sc.load[0]; Sc.store[j] = Sc.store[0]; Sc.on_total[j] = Sc.on_total[0]; Sc.S[j] = Sc.S[0];
  }
  int grid = gd_div_up(n_max, 4);
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(k_win_count2, dim3(grid, 2 * n_stages), dim3(256), 0, st, Q);
  GD_LAUNCH_CHECK();
  int rc = gd_device_scan_lb_batch(Sc, 2 * n_stages, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_win_fill2, dim3(grid, 2 * n_stages), dim3(256), 0, st, Q);
  GD_LAUNCH_CHECK();
  return 0;
}
