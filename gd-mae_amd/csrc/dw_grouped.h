// Grouped weight-gradient launch (dw_grouped.hip): job table shared with the layer executor.
#pragma once
#include <hip/hip_runtime.h>

#define GD_DW_MAX_JOBS 9

struct GdDwJob {
  const void* G;        // (rows, M) bf16 row-major: output gradient of the linear layer
  const void* X;        // (rows, N) bf16 row-major: its input
  int M, N;             // multiples of 128
  float* part;          // (S, M, N) fp32 partial products
  float* colpart;       // (S, M) fp32 column sums of G per slice, or null
  int tile0;            // first 128 x 128 tile of this job in the layer's tile list (filled by gd_dw_grouped)
  // optional gather of the X rows (sparse-convolution weight gradients: X row of operand row r = X[xidx[r * xidx_stride]], a
  // negative index = a zero row); x_f32: the X rows are fp32 (N columns) and rounded to bf16 on load
  const int* xidx;
  int xidx_stride;
  int x_f32;
  // optional: G rows narrower than M (a gradient with fewer than 128 channels, e.g. the 48-wide prediction head): row pitch g_ld
  // elements, columns >= g_cols read as zero (0 = M for both)
  int g_ld, g_cols;
};
struct GdDwGroup {
  GdDwJob job[GD_DW_MAX_JOBS];
  int n_jobs;
  int tiles_total;
  int S;
  long long rows_per_slice;
  long long n_valid;    // rows >= n_valid are treated as zero
  int guard_rows = 0;   // 1: the operands are allocated for n_valid rows only - row loads past the end repeat row n_valid - 1 (and are
                        // cleared like every row >= n_valid); 0: the buffers extend to the padded row count
  // filled by gd_dw_grouped_s: 0 = 128 x 128 tiles (k_dw_grouped), 1 / 2 = 128 x 256 pair tiles (k_dw_grouped2: one staged G chunk
  // serves two X sub-tiles - 1: adjacent column blocks of one job, every N a multiple of 256; 2: the same column block of two
  // consecutive jobs that share G, X and the shape, i.e. two taps of a gathered launch)
  int pair_mode = 0;
};

bool gd_dw_group_supported(long long n_pad, int d, int ff);
int gd_dw_group_slices(long long n_pad, int tiles_total);
int gd_dw_group_slices_for(long long n_pad, int tiles_total, int max_wgs);
int gd_dw_pick(long long n, int tiles_total, int max_wgs, long long* n_pad);
int gd_dw_grouped(hipStream_t st, GdDwGroup& A, long long n_pad, long long n_valid);
int gd_dw_grouped_s(hipStream_t st, GdDwGroup& A, long long n_pad, long long n_valid, int S);
