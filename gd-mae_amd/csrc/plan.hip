// The whole geometry plan of a pre-training step as ONE C-ABI call (SURVEY §8 rows a1-a3, a5, a6 index side, a7-a10, the
// rank part of a17 and the active-tile set of a16): voxelization, pillar-major point rows, random masking, the token set /
// cell map of every stage, all sparse-conv rulebooks (incl. the transposed ones the backward gathers through), the
// full-resolution sites under every strided token, both window partitions of every stage and the decoder's active tiles.
//
// Why one call: built op by op from the interpreter the plan was 96 launches, ~130 device allocations and ~25 framework
// ops per batch - 3-4 ms of HOST time per step and 1.3 ms of GPU time, most of it launch overhead of 5-30 us kernels
// (tools/plan_standalone.py).  Here every buffer is carved out of ONE caller-provided arena (layout =
// gdmae_geometry_plan_layout, a pure function of the capacities), every device-wide scan is a single launch (decoupled
// look-back, common.h gd_device_scan_lb; the one-thread "finalize" launches became the scans' grand-total hooks), a strided
// stage's output set is found by a scan over the OUTPUT cells (no flag array / marking pass), a stage's index tables are one
// launch, its two window partitions share their launches, the pillar-major rows are written by the ranking kernels, and all
// data-dependent counts land in ONE int32 array (a single D2H copy).  ~32 launches, no host sync, bit-identical outputs to the
// per-operator entry points (tests/test_hip_parity.py::test_prefetched_plan_is_identical_to_inline_plan).
#include "../../include/gdmae_hip.h"
#include "common.h"
#include <string.h>

// voxelize.hip / partition.hip / conv_tiles.hip
int gd_voxelize_impl(const float* points, long long n_points, int n_cols, const float* lo, const float* vs, const int* grid_xyz,
                     int batch_size, float* points_out, long long* point_coords, long long* inverse, int* inverse32,
                     long long* voxel_coords, int* pillar_cell, int* pillar_pt_off, int* pillar_pts, int* point_rank,
                     int* sample_pillar_off, float* pillar_mean, int* cell2pillar_out, int* counts, void* workspace, size_t workspace_bytes,
                     void* lb_state, float* points_pm, int* row_pillar, hipStream_t st);
size_t gd_voxelize_lb_state_bytes(long long n_points, long long cells);
size_t gd_plan_scan_state_bytes(long long n);
size_t gd_plan_win_state_bytes(long long n);
int gd_plan_visible_tokens(const float* mask, const int* cell2pillar, long long n_cells, int* tok_pillar, int* tok_cell, int* map,
                           int* n_tok, void* lb_state, hipStream_t st);
int gd_plan_downsample(const int* map_in, int B, int Yi, int Xi, int* tok_cell_out, int* map_out, int* n_out, void* lb_state,
                       hipStream_t st);
struct Dims {
  int B, Y, X;
};
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
int gd_plan_jobs(const PlanJobs& J, long long cap_max, hipStream_t st);
size_t gd_plan_windows_ws_bytes(int B, int Y, int X, int wx, int wy);
long long gd_plan_n_windows(int B, int Y, int X, int wx, int wy);
struct GdWinStage {
  const int* map;
  int B, Y, X, wx, wy, nlev;
  const int *drop_lo, *drop_hi, *max_tokens;
  int* out[2][7];
  int* counts[2];
  void* workspace;
  void* lb_state;
};
int gd_plan_windows_all(const GdWinStage* stages, int n_stages, hipStream_t st);
size_t gd_decoder_tiles_lb_state_bytes(long long nt);
int gd_decoder_tiles_lb(const int* const* maps, const int* strides, int k, int B, int H, int W, int* tile_slot, int* tile_list, int* n_act,
                        int* flag, void* lb_state, hipStream_t st);
int gd_decoder_site_rulebook(const int* site, const int* n_dev, int sites_per_tok, long long cap_sites, const int* tile_slot, int H, int W,
                             int* nbr, hipStream_t st);

namespace {

struct StageGeo {
  int Y, X;            // token grid of the stage
  long long cap;       // token capacity
  long long cap_in;    // capacity of the set it is derived from (strided stages)
  int up_s;            // full-resolution sites per token side (0: the stage does not tile the pillar grid evenly)
  long long n_win;
};

struct Layout {
  gdmae_plan_buffer* table;
  int max_entries, n;
  size_t off;
  bool ok;
  long long add(const char* name, size_t bytes) {
    const size_t o = off;
    if (table) {
      if (n >= max_entries) {
        ok = false;
        return -1;
      }
      strncpy(table[n].name, name, sizeof(table[n].name) - 1);
      table[n].name[sizeof(table[n].name) - 1] = 0;
      table[n].offset = (long long)o;
      table[n].bytes = (long long)bytes;
    }
    ++n;
    off += gd_align(bytes);
    return (long long)o;
  }
};

struct Offsets {
  long long lb_state, counts, points, point_coords, inverse, inverse32, voxel_coords, pillar_cell, pt_off, pillar_pts, point_rank,
      sample_off, pillar_mean, cell2pillar, points_pm, row_pillar, vox_ws, mask, len_keep, tok_pillar;
  long long vis_tok_cell, vis_map;
  struct {
    long long tok_cell, map, nbr_subm, nbr_subm_t, nbr_down, nbr_down_t, up_sites, win_ws;
    long long w[2][7];
  } s[GDMAE_PLAN_MAX_STAGES];
  long long dec_slot, dec_list, dec_flag, dec_nbr[3];
  size_t lb_bytes, vox_ws_bytes;
  int n_counts;
  bool dec;
  StageGeo geo[GDMAE_PLAN_MAX_STAGES];
  long long cap_pts, m_cap, cells;
};

int check(const gdmae_plan_params* p) {
  GD_REQUIRE(p != nullptr, "geometry plan: null parameters");
  GD_REQUIRE(p->n_points >= 0 && p->n_points < (1ll << 31) && p->n_cols >= 4 && p->n_cols <= 65 && p->batch_size >= 1,
             "geometry plan: bad point buffer");
  GD_REQUIRE(p->grid[0] >= 1 && p->grid[1] >= 1 && p->grid[2] == 1, "geometry plan: single-layer pillar grids (spt_backbone_mae.py:94)");
  GD_REQUIRE(p->n_stages >= 1 && p->n_stages <= GDMAE_PLAN_MAX_STAGES, "geometry plan: 1..4 stages");
  for (int i = 0; i < p->n_stages; ++i) {
    GD_REQUIRE(p->stride[i] == 1 || p->stride[i] == 2, "geometry plan: conv_down stride 1 or 2 (k3 s2 p1)");
    GD_REQUIRE(i == 0 || p->stride[i] == 2, "geometry plan: only the first stage may keep the resolution");
    GD_REQUIRE(p->win_x[i] >= 1 && p->win_y[i] >= 1 && p->win_x[i] * p->win_y[i] <= 64, "geometry plan: window must fit one wavefront");
    GD_REQUIRE(p->n_levels[i] >= 1 && p->n_levels[i] <= 3, "geometry plan: 1..3 drop levels");
  }
  GD_REQUIRE(p->n_dec >= 0 && p->n_dec <= 3, "geometry plan: at most 3 decoder source stages");
  for (int g = 0; g < p->n_dec; ++g) GD_REQUIRE(p->dec_sources[g] >= 0 && p->dec_sources[g] < p->n_stages, "geometry plan: decoder source");
  return 0;
}

// the layout is a pure function of the parameters (capacities, not data): Python caches the table per shape
int layout(const gdmae_plan_params* p, gdmae_plan_buffer* table, int max_entries, Offsets& O, size_t* total, int* n_entries) {
  if (int rc = check(p)) return rc;
  Layout L{table, max_entries, 0, 0, true};
  const int B = p->batch_size, gx = p->grid[0], gy = p->grid[1], F = p->n_cols - 1;
  const long long n0 = p->cap_points > p->n_points ? p->cap_points : p->n_points;   // capacities, not data
  O.cells = (long long)B * gx * gy;
  GD_REQUIRE(O.cells < (1ll << 31), "geometry plan: B*Y*X must fit int32");
  O.cap_pts = n0 > 0 ? n0 : 1;
  O.m_cap = n0 < O.cells ? n0 : O.cells;
  if (O.m_cap < 1) O.m_cap = 1;
  const long long cap = O.cap_pts;
  // ---- stage geometry
  int Y = gy, X = gx;
  long long c = O.m_cap;
  for (int i = 0; i < p->n_stages; ++i) {
    StageGeo& g = O.geo[i];
    g.cap_in = c;
    if (p->stride[i] == 2) {
      Y = (Y - 1) / 2 + 1;
      X = (X - 1) / 2 + 1;
      c = 4 * c < (long long)B * Y * X ? 4 * c : (long long)B * Y * X;
    }
    g.Y = Y; g.X = X; g.cap = c;
    const int us = gy / Y;
    g.up_s = (us >= 1 && us * Y == gy && us * X == gx) ? us : 0;
    g.n_win = gd_plan_n_windows(B, Y, X, p->win_x[i], p->win_y[i]);
    GD_REQUIRE(g.n_win < (1 << 21), "geometry plan: window grid too large for the packed scan");
  }
  O.dec = p->n_dec > 0;
  for (int g = 0; g < p->n_dec; ++g) {
    const int us = O.geo[p->dec_sources[g]].up_s;
    O.dec = O.dec && (us == 1 || us == 2 || us == 4 || us == 8);
  }
  const long long nt = (long long)B * ((gy + 7) / 8) * ((gx + 7) / 8);
  // ---- look-back states of every scan, cleared by one memset (kept first)
  size_t lb = gd_voxelize_lb_state_bytes(n0, O.cells) + gd_plan_scan_state_bytes(O.cells);
  for (int i = 0; i < p->n_stages; ++i) {
    if (p->stride[i] == 2) lb += gd_plan_scan_state_bytes((long long)B * O.geo[i].Y * O.geo[i].X);
    lb += 2 * gd_plan_win_state_bytes(O.geo[i].n_win);
  }
  if (O.dec) lb += gd_decoder_tiles_lb_state_bytes(nt);
  O.lb_bytes = lb;
  O.lb_state = L.add("lb_state", lb);
  O.n_counts = 2 + p->n_stages + 16 * p->n_stages + 1 + 1;     // N, M | tokens per stage | 8 per (stage, shift) | active tiles | visible pillars
  O.counts = L.add("counts", sizeof(int) * O.n_counts);
  // ---- voxelization (capacity = number of input points)
  O.points = L.add("points", sizeof(float) * cap * p->n_cols);
  O.point_coords = L.add("point_coords", sizeof(long long) * cap * 4);
  O.inverse = L.add("inverse", sizeof(long long) * cap);
  O.inverse32 = L.add("inverse32", sizeof(int) * cap);
  O.voxel_coords = L.add("voxel_coords", sizeof(long long) * cap * 4);
  O.pillar_cell = L.add("pillar_cell", sizeof(int) * cap);
  O.pt_off = L.add("pt_off", sizeof(int) * (cap + 1));
  O.pillar_pts = L.add("pillar_pts", sizeof(int) * cap);
  O.point_rank = L.add("point_rank", sizeof(int) * cap);
  O.sample_off = L.add("sample_off", sizeof(int) * (B + 1));
  O.pillar_mean = L.add("pillar_mean", sizeof(float) * cap * F);
  O.cell2pillar = L.add("cell2pillar", sizeof(int) * O.cells);
  O.points_pm = L.add("points_pm", sizeof(float) * cap * p->n_cols);
  O.row_pillar = L.add("row_pillar", sizeof(int) * cap);
  O.vox_ws_bytes = gdmae_voxelize_workspace_bytes(n0, B, gx, gy, 1);
  O.vox_ws = L.add("vox_ws", O.vox_ws_bytes);
  // ---- masking, visible pillars
  O.mask = L.add("mask", sizeof(float) * O.m_cap);
  O.len_keep = L.add("len_keep", sizeof(int) * B);
  O.tok_pillar = L.add("tok_pillar", sizeof(int) * O.m_cap);
  O.vis_tok_cell = O.vis_map = -1;
  if (p->stride[0] == 2) {
    O.vis_tok_cell = L.add("vis.tok_cell", sizeof(int) * O.m_cap);
    O.vis_map = L.add("vis.map", sizeof(int) * O.cells);
  }
  // ---- stages
  char nm[40];
  static const char* const kWin[7] = {"tok_win", "tok_level", "tok_slot", "tok_pos", "csr_tok", "win_start", "win_len"};
  for (int i = 0; i < p->n_stages; ++i) {
    const StageGeo& g = O.geo[i];
    auto name = [&](const char* s) { snprintf(nm, sizeof(nm), "s%d.%s", i, s); return (const char*)nm; };
    O.s[i].tok_cell = L.add(name("tok_cell"), sizeof(int) * g.cap);
    O.s[i].map = L.add(name("map"), sizeof(int) * (long long)B * g.Y * g.X);
    O.s[i].nbr_subm = L.add(name("nbr_subm"), sizeof(int) * g.cap * 9);
    O.s[i].nbr_subm_t = L.add(name("nbr_subm_t"), sizeof(int) * g.cap * 9);
    O.s[i].nbr_down = O.s[i].nbr_down_t = O.s[i].up_sites = -1;
    if (p->stride[i] == 2) {
      O.s[i].nbr_down = L.add(name("nbr_down"), sizeof(int) * g.cap * 9);
      O.s[i].nbr_down_t = L.add(name("nbr_down_t"), sizeof(int) * g.cap_in * 9);
    }
    if (g.up_s > 1) O.s[i].up_sites = L.add(name("up_sites"), sizeof(int) * g.cap * g.up_s * g.up_s);
    O.s[i].win_ws = L.add(name("win_ws"), gd_plan_windows_ws_bytes(B, g.Y, g.X, p->win_x[i], p->win_y[i]));
    const long long wcap = g.n_win < g.cap ? g.n_win : g.cap;
    for (int sh = 0; sh < 2; ++sh)
      for (int k = 0; k < 7; ++k) {
        snprintf(nm, sizeof(nm), "s%d.w%d.%s", i, sh, kWin[k]);
        O.s[i].w[sh][k] = L.add(nm, sizeof(int) * (k < 5 ? g.cap : wcap));
      }
  }
  O.dec_slot = O.dec_list = O.dec_flag = -1;
  if (O.dec) {
    O.dec_slot = L.add("dec.tile_slot", sizeof(int) * nt);
    O.dec_list = L.add("dec.tile_list", sizeof(int) * nt);
    O.dec_flag = L.add("dec.flag", sizeof(int) * nt);
    // rulebooks of the conv_out backward: tile-compact row of every (active site, tap) per source stage
    for (int g = 0; g < p->n_dec; ++g) {
      const StageGeo& sg = O.geo[p->dec_sources[g]];
      snprintf(nm, sizeof(nm), "dec.nbr%d", g);
      O.dec_nbr[g] = L.add(nm, sizeof(int) * sg.cap * sg.up_s * sg.up_s * 9);
    }
  }
  GD_REQUIRE(L.ok, "geometry plan: buffer table too small");
  if (total) *total = L.off;
  if (n_entries) *n_entries = L.n;
  return 0;
}

}  // namespace

extern "C" int gdmae_geometry_plan_layout(const gdmae_plan_params* p, gdmae_plan_buffer* table, int max_entries, int* n_entries,
                                          size_t* total_bytes) {
  Offsets O;
  return layout(p, table, max_entries, O, total_bytes, n_entries);
}

extern "C" int gdmae_geometry_plan(const gdmae_plan_params* p, const float* points, const float* noise, void* arena, size_t arena_bytes,
                                   void* stream) {
  Offsets O;
  size_t total = 0;
  if (int rc = layout(p, nullptr, 0, O, &total, nullptr)) return rc;
  GD_REQUIRE(arena != nullptr && arena_bytes >= total, "geometry plan: arena too small (gdmae_geometry_plan_layout)");
  GD_REQUIRE(!p->masked || noise != nullptr, "geometry plan: masking needs one noise value per pillar (capacity min(points, cells))");
  hipStream_t st = (hipStream_t)stream;
  char* A = (char*)arena;
  auto I = [&](long long off) { return off < 0 ? (int*)nullptr : (int*)(A + off); };
  const int B = p->batch_size, gx = p->grid[0], gy = p->grid[1], ns = p->n_stages;
  int* counts = I(O.counts);
  int* n_tok = counts + 2;                       // per stage
  int* win_counts = counts + 2 + ns;             // 8 per (stage, shift)
  int* n_act = counts + 2 + ns + 16 * ns;
  int* n_vis = n_act + 1;
  // one clear for: every look-back state and the counts
  GD_CHECK(hipMemsetAsync(A + O.lb_state, 0, (size_t)(O.counts - O.lb_state) + gd_align(sizeof(int) * O.n_counts), st));
  char* lb = A + O.lb_state;
  // ---- a1-a3 (+ canonical point order, pillar means, pillar-major rows)
  {
    int rc = gd_voxelize_impl(points, p->n_points, p->n_cols, p->lo, p->vs, p->grid, B, (float*)(A + O.points), (long long*)(A + O.point_coords),
                              (long long*)(A + O.inverse), I(O.inverse32), (long long*)(A + O.voxel_coords), I(O.pillar_cell), I(O.pt_off),
                              I(O.pillar_pts), I(O.point_rank), I(O.sample_off), (float*)(A + O.pillar_mean), I(O.cell2pillar), counts,
                              A + O.vox_ws, O.vox_ws_bytes, lb, p->want_pm ? (float*)(A + O.points_pm) : nullptr,
                              p->want_pm ? I(O.row_pillar) : nullptr, st);
    if (rc) return rc;
    lb += gd_voxelize_lb_state_bytes(p->n_points, O.cells);
  }
  // ---- a5 masking
  float* mask = (float*)(A + O.mask);
  if (p->masked) {
    int rc = gdmae_random_mask(noise, I(O.sample_off), B, p->keep_frac, mask, I(O.len_keep), stream);
    if (rc) return rc;
  } else {
    GD_CHECK(hipMemsetAsync(mask, 0, sizeof(float) * O.m_cap, st));
  }
  // ---- visible pillars = input set of the first stage
  const bool first_strided = p->stride[0] == 2;
  int* cur_cell = first_strided ? I(O.vis_tok_cell) : I(O.s[0].tok_cell);
  int* cur_map = first_strided ? I(O.vis_map) : I(O.s[0].map);
  int* cur_n = first_strided ? n_vis : n_tok;
  {
    int rc = gd_plan_visible_tokens(mask, I(O.cell2pillar), O.cells, I(O.tok_pillar), cur_cell, cur_map, cur_n, lb, st);
    if (rc) return rc;
    lb += gd_plan_scan_state_bytes(O.cells);
  }
  // ---- the stages.  The token sets are a chain (a strided stage's set = a scan over its output cells of the previous map); every
  //      other table of a stage only needs the maps, so after the chain ALL index tables of ALL stages are one launch and the two
  //      window partitions of ALL stages three (count, scans, fill) - the plan is a serial chain of short launches on its own
  //      stream whose time the step pays almost in full
  int Yp = gy, Xp = gx;                          // grid of the current set
  PlanJobs J;
  J.count = 0;
  long long cap_max = 1;
  GdWinStage W[4];
  GD_REQUIRE(ns <= 4, "geometry plan: at most four stages");
  for (int i = 0; i < ns; ++i) {
    const StageGeo& g = O.geo[i];
    GD_REQUIRE(J.count + 4 <= kPlanMaxJobs, "geometry plan: job table full");
    if (g.cap > cap_max) cap_max = g.cap;
    if (p->stride[i] == 2) {
      int rc = gd_plan_downsample(cur_map, B, Yp, Xp, I(O.s[i].tok_cell), I(O.s[i].map), n_tok + i, lb, st);
      if (rc) return rc;
      lb += gd_plan_scan_state_bytes((long long)B * g.Y * g.X);
      // strided forward rulebook (output token <- 9 input taps) and its transpose (input token -> outputs it feeds)
      J.j[J.count++] = PlanJob{n_tok + i, I(O.s[i].tok_cell), Dims{B, g.Y, g.X}, Dims{B, Yp, Xp}, cur_map, 1, 0, I(O.s[i].nbr_down), nullptr};
      J.j[J.count++] = PlanJob{cur_n, cur_cell, Dims{B, Yp, Xp}, Dims{B, g.Y, g.X}, I(O.s[i].map), 2, 0, I(O.s[i].nbr_down_t), nullptr};
      if (g.cap_in > cap_max) cap_max = g.cap_in;
      cur_cell = I(O.s[i].tok_cell);
      cur_map = I(O.s[i].map);
      cur_n = n_tok + i;
      Yp = g.Y;
      Xp = g.X;
    }
    J.j[J.count++] = PlanJob{cur_n, cur_cell, Dims{B, g.Y, g.X}, Dims{B, g.Y, g.X}, cur_map, 0, 0, I(O.s[i].nbr_subm), I(O.s[i].nbr_subm_t)};
    if (g.up_s > 1) {
      J.j[J.count++] = PlanJob{cur_n, cur_cell, Dims{B, g.Y, g.X}, Dims{B, g.Y, g.X}, nullptr, 3, g.up_s, I(O.s[i].up_sites), nullptr};
      if (g.cap * g.up_s * g.up_s / 9 + 1 > cap_max) cap_max = g.cap * g.up_s * g.up_s / 9 + 1;
    }
    GdWinStage& w = W[i];
    w.map = cur_map;
    w.B = B; w.Y = g.Y; w.X = g.X; w.wx = p->win_x[i]; w.wy = p->win_y[i]; w.nlev = p->n_levels[i];
    w.drop_lo = p->drop_lo[i]; w.drop_hi = p->drop_hi[i]; w.max_tokens = p->max_tokens[i];
    w.counts[0] = win_counts + 16 * i;
    w.counts[1] = win_counts + 16 * i + 8;
    for (int sh = 0; sh < 2; ++sh)
      for (int k = 0; k < 7; ++k) w.out[sh][k] = I(O.s[i].w[sh][k]);
    w.workspace = A + O.s[i].win_ws;
    w.lb_state = lb;
    lb += 2 * gd_plan_win_state_bytes(g.n_win);
  }
  {
    int rc = gd_plan_jobs(J, cap_max, st);
    if (rc) return rc;
    rc = gd_plan_windows_all(W, ns, st);
    if (rc) return rc;
  }
  // ---- active tiles of the decoder's 3x3 convolution
  if (O.dec) {
    const int* maps[3];
    int ups[3];
    for (int g = 0; g < p->n_dec; ++g) {
      maps[g] = I(O.s[p->dec_sources[g]].map);
      ups[g] = O.geo[p->dec_sources[g]].up_s;
    }
    int rc = gd_decoder_tiles_lb(maps, ups, p->n_dec, B, gy, gx, I(O.dec_slot), I(O.dec_list), n_act, I(O.dec_flag), lb, st);
    if (rc) return rc;
    for (int g = 0; g < p->n_dec; ++g) {
      const int si = p->dec_sources[g];
      const StageGeo& sg = O.geo[si];
      const int* sites = sg.up_s > 1 ? I(O.s[si].up_sites) : I(O.s[si].tok_cell);
      rc = gd_decoder_site_rulebook(sites, n_tok + si, sg.up_s * sg.up_s, sg.cap * sg.up_s * sg.up_s, I(O.dec_slot), gy, gx, I(O.dec_nbr[g]), st);
      if (rc) return rc;
    }
  }
  return 0;
}
