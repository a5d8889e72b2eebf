/* gdmae_hip.h - C ABI of libgdmae_hip.so: the MI355X (gfx950) native implementation of the GD-MAE
 * pre-training hot path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the parameter is documented "host";
 *  - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); every launch goes to it,
 *    nothing synchronises with the host, nothing allocates: all scratch is caller provided;
 *  - return value 0 = ok, non-zero = failure (hipError_t value or -1 for a violated precondition);
 *    gdmae_last_error() returns a thread-local description.  The library never calls exit()
 *    (the reference's wrappers do: pcdet/ops/sst_ops/src/sst_ops.cpp:7-19);
 *  - element counts that are data dependent (pillars, tokens, windows) are produced ON THE DEVICE in
 *    small int32 `counts` arrays; the host reads them once per step after the whole geometry plan has
 *    been enqueued (one D2H copy instead of the reference's dozens of .item() syncs);
 *  - "canonical order" = ascending original index; it replaces the atomic arrival order
 *    of the reference kernels (pcdet/ops/sst_ops/src/sst_ops_gpu.cu:14-28) and makes results deterministic.
 *
 * The reference's FFI for this path is the pybind11 module pcdet.ops.sst_ops.sst_ops_cuda
 * (pcdet/ops/sst_ops/src/sst_ops_api.cpp:6-9: ingroup_inds_wrapper, group_inner_inds_wrapper) plus three
 * un-vendored GPU libraries reached from Python (spconv, torch_scatter, pytorch3d).  Each entry point
 * below names the reference interface it replaces.  INTEGRATION.md shows the binding a maintainer adds.
 */
#ifndef GDMAE_HIP_H
#define GDMAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- housekeeping --------------------------------------------------------------------------- */
int gdmae_abi_version(void);
const char* gdmae_target_arch(void); /* "gfx950" */
const char* gdmae_last_error(void);

/* ---- a1-a3, a17(rank): dynamic pillar voxelisation ------------------------------------------ *
 * Replaces: common_utils.get_in_range_mask (pcdet/utils/common_utils.py:66-76), the coord build and
 * coords.unique(dim=0, return_inverse=True) of DynVFE.forward (pcdet/models/backbones_3d/vfe/dyn_vfe.py:62-68),
 * torch_scatter.scatter(reduce='mean') (dyn_vfe.py:81) and the in-pillar rank that
 * group_inner_inds_kernel derives from atomics (sst_ops_gpu.cu:22-28).
 *
 * points        (n_points, n_cols) fp32 rows [batch_idx, x, y, z, f...]; n_cols = 1 + F.
 * lo, vs        HOST float[3]: point_cloud_range[:3], voxel_size.      grid_xyz  HOST int[3] = (X, Y, Z).
 * Outputs (capacity in elements; N = kept points, M = pillars, both <= n_points):
 *   points_out (n_points*n_cols)  kept rows in original order          point_coords (n_points*4) int64 [b,z,y,x]
 *   inverse (n_points) int64, inverse32 (n_points) int32  pillar id of each kept point
 *   voxel_coords (n_points*4) int64 [b,z,y,x], lexicographically ascending (= torch.unique(dim=0) order)
 *   pillar_cell (n_points) int32  linear cell key ((b*Z+z)*Y+y)*X+x of each pillar
 *   pillar_pt_off (n_points+1) int32 CSR offsets; pillar_pts (n_points) int32 kept-point ids grouped by
 *   pillar, ascending inside a pillar; point_rank (n_points) int32 rank of each kept point in its pillar
 *   sample_pillar_off (batch_size+1) int32 first pillar of every sample
 *   pillar_mean (n_points*F) fp32 per-pillar mean of [x,y,z,f...] accumulated in canonical order
 *   cell2pillar (B*Z*Y*X) int32 dense map cell -> pillar id or -1 (may be NULL: kept in the workspace)
 *   counts int32[2] = {N, M}
 * workspace: gdmae_voxelize_workspace_bytes(...) bytes. */
size_t gdmae_voxelize_workspace_bytes(long long n_points, int batch_size, int gx, int gy, int gz);
int gdmae_voxelize(const float* points, long long n_points, int n_cols, const float* lo, const float* vs,
                   const int* grid_xyz, int batch_size, float* points_out, long long* point_coords,
                   long long* inverse, int* inverse32, long long* voxel_coords, int* pillar_cell,
                   int* pillar_pt_off, int* pillar_pts, int* point_rank, int* sample_pillar_off,
                   float* pillar_mean, int* cell2pillar, int* counts, void* workspace, size_t workspace_bytes,
                   void* stream);

/* ---- a4: point decoration, segmented max ---------------------------------------------------- *
 * gdmae_decorate_points replaces dyn_vfe.py:85-105 ([xyz - centre, xyz+feat, xyz - mean], 6+F channels).
 * gdmae_segment_max[_bwd] replaces torch_scatter.scatter_max (dyn_vfe.py:109): out (M,C), arg (M,C) =
 * kept-point id of the maximum (lowest id on ties); backward routes dout to the arg-max row only. */
int gdmae_decorate_points(const float* points, const long long* point_coords, const int* inverse32,
                          const float* pillar_mean, long long N, int n_cols, const float* lo, const float* vs,
                          float* out, void* stream);
/* First DynVFE point layer as one call per direction (dyn_vfe.py:74-109 + network_utils.py:7-21: decoration,
 * Linear(6+F -> 64, no bias), BatchNorm1d(train), ReLU).  The (N, 64) pre-activation is never stored: it is
 * recomputed from the points in MFMA accumulators for the statistics, the output, the backward statistics and the
 * weight gradient.  W (64, 6+F) fp32; out (N, 64) fp32 (out_bf16 = 0), bf16 (1) or fp16 (2: the rows gdmae_vfe_max_layer_*_f16
 * read), g (N, 64) bf16 or fp32; stats / ab / mv as gdmae_bn_fold;
 * dgamma / dbeta / dW written, or accumulated into when `accumulate`.  Row i of out / g is row i of `points`.
 * coords_per_pillar != 0: the rows are the pillar-major ones of gdmae_pillar_major_rows (rows of a pillar contiguous,
 * as gdmae_vfe_max_layer_* wants them): points = points_pm, inverse32 = row_pillar, point_coords = the (M, 4)
 * voxel_coords table read through the pillar id.
 * gdmae_pillar_major_rows: points_pm[q] = points[pillar_pts[q]], row_pillar[q] = inverse32[pillar_pts[q]] for q < *n_dev
 * (device-side count, e.g. gdmae_voxelize counts[0]; buffers sized for `capacity` rows). */
int gdmae_pillar_major_rows(const float* points, int n_cols, const int* pillar_pts, const int* inverse32, const int* n_dev,
                            long long capacity, float* points_pm, int* row_pillar, void* stream);
size_t gdmae_vfe_point_layer_workspace_bytes(int n_cols);
int gdmae_vfe_point_layer_fwd(const float* points, const long long* point_coords, const int* inverse32,
                              const float* pillar_mean, int coords_per_pillar, long long N, int n_cols, const float* lo,
                              const float* vs,
                              const float* W, int C, const float* gamma, const float* beta, double eps, double momentum,
                              float* running_mean, float* running_var, long long* num_batches, double* stats, float* ab,
                              float* mv, void* out, int out_bf16, void* workspace, void* stream);
int gdmae_vfe_point_layer_bwd(const float* points, const long long* point_coords, const int* inverse32,
                              const float* pillar_mean, int coords_per_pillar, long long N, int n_cols, const float* lo,
                              const float* vs,
                              const float* W, int C, const float* gamma, const double* stats, const float* ab,
                              const void* g, int g_bf16, float* dgamma, float* dbeta, float* dW, int accumulate,
                              void* workspace, void* stream);
/* Second DynVFE point layer + per-pillar maximum as one call per direction, bf16 throughput mode (dyn_vfe.py:107-112:
 * Linear(64 -> 128, no bias), BatchNorm1d(train), ReLU, torch_scatter.scatter_max).  The (N, 128) pre-activation is
 * recomputed from y1 in MFMA accumulators by every kernel.  y1 (N, 64) bf16 with rows in pillar-major order and
 * row_pillar (N) from gdmae_pillar_major_rows + gdmae_vfe_point_layer_fwd(coords_per_pillar = 1); pillar_pt_off (M + 1) = first row of each
 * pillar; W (128, 64) bf16; out (M, 128) fp32, arg = row of the maximum (first row on ties = lowest point id); g = gradient of out; gm = scratch of
 * M * 128 floats; dy1 (N, 64) bf16 written; dgamma / dbeta / dW (128, 64) fp32 written, or accumulated into. */
size_t gdmae_vfe_max_layer_workspace_bytes(void);
int gdmae_vfe_max_layer_fwd(const void* y1, long long N, const void* W, const int* pillar_pt_off, const int* row_pillar,
                            int M, const float* gamma, const float* beta, double eps, double momentum,
                            float* running_mean, float* running_var, long long* num_batches, double* stats, float* ab,
                            float* mv, float* out, int* arg, void* workspace, void* stream);
int gdmae_vfe_max_layer_bwd(const void* y1, long long N, const void* W, const int* row_pillar, int M, const float* gamma,
                            const double* stats, const float* ab, const float* out, const int* arg, const float* g, void* gm,
                            void* dy1, float* dgamma, float* dbeta, float* dW, int accumulate,
                            void* workspace, void* stream);
/* Round 6, the same layer on fp16 rows: y1 (N, 64) holds fp16 values (gdmae_vfe_point_layer_fwd with out_bf16 = 2) and W is the fp32
 * (128, 64) master matrix - rounded to fp16 for the pre-activation y1 W^T (v_mfma_f32_32x32x16_f16) and to bf16 for the gradient product
 * dy1 = dx W; dy1 stays bf16.  The pillar maximum passes ONE point's value on, so the rounding of y1 and of W does not average over a
 * pillar's points: fp16 (11 significand bits; the rows are BatchNorm + ReLU outputs) instead of bf16 (8) at the same byte count and
 * matrix-core rate.  Same dyn_vfe.py:107-112 / scatter_max semantics as above. */
int gdmae_vfe_max_layer_fwd_f16(const void* y1, long long N, const float* W, const int* pillar_pt_off, const int* row_pillar,
                                int M, const float* gamma, const float* beta, double eps, double momentum,
                                float* running_mean, float* running_var, long long* num_batches, double* stats, float* ab,
                                float* mv, float* out, int* arg, void* workspace, void* stream);
int gdmae_vfe_max_layer_bwd_f16(const void* y1, long long N, const float* W, const int* row_pillar, int M, const float* gamma,
                                const double* stats, const float* ab, const float* out, const int* arg, const float* g, void* gm,
                                void* dy1, float* dgamma, float* dbeta, float* dW, int accumulate,
                                void* workspace, void* stream);
int gdmae_segment_max(const float* x, const int* pillar_pt_off, const int* pillar_pts, int M, int C, float* out,
                      int* arg, void* stream);
int gdmae_segment_max_bwd(const float* dout, const int* arg, const int* inverse32, long long N, int C, float* dx,
                          void* stream);

/* ---- a5: MAE random masking ------------------------------------------------------------------ *
 * Replaces common_utils.random_masking (common_utils.py:49-63) and the per-sample loop with a host sync
 * per sample (pcdet/models/backbones_3d/spt_backbone_mae.py:96-100).  noise: one fp32 per pillar;
 * keep_frac = (1 - RATIO) evaluated by the caller in fp64; len_keep = (int)(L * keep_frac) per sample.
 * mask_out (M) fp32: 0 = visible (the len_keep smallest noise values, ties -> lower index), 1 = masked. */
int gdmae_random_mask(const float* noise, const int* sample_pillar_off, int batch_size, double keep_frac,
                      float* mask_out, int* len_keep_out, void* stream);

/* ---- a6 (index side): token sets and sparse-conv rulebooks ----------------------------------- *
 * Replace spconv's hash-table rulebook construction for SubMConv2d(k3) / SparseConv2d(k3,s2,p1)
 * (call sites pcdet/utils/spconv_utils.py:41-43; spt_backbone.py:206,217) and the boolean-mask
 * compaction features[mask == 0] (spt_backbone_mae.py:102-107).
 * A token set = ascending linear cell keys tok_cell[] ((b*Y+y)*X+x) + dense map cell -> token (-1 empty).
 * gdmae_visible_tokens: stage-1 tokens = pillars with mask == 0.   vox_counts = counts of gdmae_voxelize.
 * gdmae_downsample_tokens: output active set of a k3 s2 p1 sparse conv (site o active iff an active
 *   input 2*o-1+k exists), ordered by linear key at the output resolution (Yo = (Yi-1)/2+1).
 * gdmae_rulebook: nbr[t*9+k] = token feeding t through tap k=ky*3+kx or -1;
 *   mode 0 submanifold, 1 strided forward (map = input map), 2 strided transposed (map = output map). */
int gdmae_visible_tokens(const float* mask, const int* pillar_cell, const int* vox_counts, long long m_cap,
                         long long n_cells, int* tok_pillar, int* tok_cell, int* map, int* n_tok, void* scan_ws,
                         void* stream);
int gdmae_downsample_tokens(const int* n_in, const int* tok_cell_in, long long cap_in, int B, int Yi, int Xi,
                            int* tok_cell_out, int* map_out, int* n_out, int* flag_ws, void* scan_ws, void* stream);
int gdmae_rulebook(const int* n_tok, const int* tok_cell, long long cap, int B, int Yt, int Xt, int Ym, int Xm,
                   const int* map, int mode, int* nbr, void* stream);

/* ---- a7-a10: shifted-window partition --------------------------------------------------------- *
 * Replaces get_window_coors (pcdet/models/model_utils/sst_utils.py:6-47), get_inner_win_inds ->
 * ingroup_inds_kernel (sst_ops_gpu.cu:14-20), drop_single_shift (spt_backbone.py:32-51),
 * make_continuous_inds / get_flat2win_inds (sst_utils.py:50-104).
 * Per token (capacity = tokens): tok_win = batch_win_inds, tok_level = drop level, tok_slot = flat2window
 * index inside the level (dense_window * max_tokens + canonical rank), tok_pos = in-window cell y*wx+x.
 * Window lists ordered by (level, dense index): win_start/win_len into csr_tok (tokens of a window,
 * canonical order).  counts int32[8]: [0..2] windows per level, [3..5] tokens per level, [6],[7] totals. */
size_t gdmae_window_workspace_bytes(int B, int Y, int X, int wx, int wy);
int gdmae_window_partition(const int* map, int B, int Y, int X, int wx, int wy, int shifted, int nlev,
                           const int* drop_lo, const int* drop_hi, const int* max_tokens, int* tok_win,
                           int* tok_level, int* tok_slot, int* tok_pos, int* csr_tok, int* win_start, int* win_len,
                           int* counts, void* workspace, size_t workspace_bytes, void* stream);

/* ---- a6/a16 data movement: row gather / scatter ----------------------------------------------- *
 * gdmae_gather_rows: out[s,:] = idx[s] >= 0 ? src[idx[s],:] : 0  (rulebook im2col with n_slots = tokens*9,
 *   gather of decoder features at pillar sites: spt_backbone_mae.py:141-143).
 * gdmae_scatter_rows: dst[idx[r],:] = src[r,:] (SparseConvTensor.dense(): spt_backbone_mae.py:128).
 * row_bytes must be a multiple of 16; dtype agnostic. */
int gdmae_gather_rows(const void* src, const int* idx, long long n_slots, int row_bytes, void* out, void* stream);
int gdmae_scatter_rows(const void* src, const int* idx, long long n_rows, int row_bytes, void* dst, void* stream);

/* Column-slice variants (the table - src of the gather / dst of the scatter - has rows of table_row_bytes and
 * the moved slice starts at table_col_bytes): write / read one 128-channel source of the decoder's 384-channel
 * concatenated map in place, instead of torch.cat (spt_backbone_mae.py:132). */
int gdmae_gather_rows_strided(const void* src, const int* idx, long long n_slots, int row_bytes, int table_row_bytes,
                              int table_col_bytes, void* out, void* stream);
int gdmae_scatter_rows_strided(const void* src, const int* idx, long long n_rows, int row_bytes, int table_row_bytes,
                               int table_col_bytes, void* dst, void* stream);
/* gdmae_colstats: out double[2*C] = per-column {sum, sum of squares} of a row-major (R, C) fp32 (is_bf16 = 0)
 * or bf16 (1) matrix; deterministic.  BatchNorm2d batch statistics of channels-last dense maps
 * (nn.BatchNorm2d in spt_backbone_mae.py:40,47) and column sums of dense gradients. */
size_t gdmae_colstats_workspace_bytes(int C);
int gdmae_colstats(const void* x, long long R, int C, int is_bf16, double* out, void* workspace, void* stream);

/* ---- a16 (token side of the sparse-aware decoder) ----------------------------------------------- *
 * Replace ConvTranspose2d -> BatchNorm2d -> ReLU -> cat (spt_backbone_mae.py:30-44,125-132) on the active-site
 * rows P (n, C) (fp32 or bf16), C <= 256; a, b, c0, c1 are per-channel fp32 vectors from the BatchNorm algebra.
 *   gdmae_rows_affine_relu_scatter: Z[site[r], col0:col0+C] = relu(a*P[r]+b)   (Z rows of z_row_elems elements)
 *   gdmae_rows_bwd_stats: out double[3C] = column sums of {dh, dh*P, g},  g = dZ[site[r], slice], dh = g*(aP+b>0)
 *   gdmae_rows_bwd: dP[r] = a*dh + c0 + c1*P[r] */
/* Small algebra of the decoder-head backward (hand-derived backward of conv_out + BatchNorm2d, spt_backbone_mae.py:46-52):
 * from the 9 border-region sums of dY it produces S (9, C2) f64 = sum of dY each tap can reach, tot (Cin) f64 = column
 * sums of dZ over all sites, dWk (9, C2, Cin) fp32 = background part of the weight gradient, Wd (9, C2, Cin) = the conv
 * weight (C2, Cin, 3, 3) re-laid per tap in the compute dtype.  reg (16, C2) from gdmae_border_sums; tap_region (9, 9) and
 * cnt (9) f64 constants of the map geometry; R = number of sites. */
size_t gdmae_decoder_region_workspace_bytes(int C2, int Cin);
int gdmae_decoder_region_algebra(const double* stats2, const float* ab2, const double* st2, const float* k01,
                                 const double* reg, const double* tap_region, const double* cnt, double R, int C2, int Cin,
                                 const float* conv_w, const void* bgz, int cdt_bf16, double* S, double* tot, float* dWk,
                                 void* Wd, void* workspace, void* stream);
/* Z (R, C) = row vector v in every row (C * elem_bytes a multiple of 16): background of the dense decoder map
 * (replaces bg.expand(R, C).contiguous(), spt_backbone_mae.py:125-133 densify of the non-active sites). */
int gdmae_fill_rows(const void* v, long long R, int C, int elem_bytes, void* Z, void* stream);
int gdmae_rows_affine_relu_scatter(const void* P, int p_bf16, const int* site, long long n, int C, const float* a,
                                   const float* b, void* Z, int z_bf16, int z_row_elems, int col0, void* stream);
/* ... minus a per-channel constant sub (C) of Z's dtype, rounded like the two-step sequence (the decoder backward's Z - background
 * rows); C, the row pitch and col0 must be multiples of 8. */
int gdmae_rows_affine_relu_sub(const void* P, int p_bf16, const int* site, long long n, int C, const float* a, const float* b,
                               const void* sub, void* Z, int z_bf16, int z_row_elems, int col0, void* stream);
/* ... plus an identity shortcut: Z[r] = relu(a P[r] + b) + R[r] for rows R, Z (n, C) of the same dtype, rounded like the two-step
 * sequence - BatchNorm2d + ReLU + `y + x` of a dense channels-last Conv-BN-ReLU block (sst_bev_backbone.py:36-40, fine-tune row
 * f1; a channels-last (B, C, Y, X) map is the row-major (B Y X, C) matrix). */
int gdmae_rows_affine_relu_add(const void* P, int p_bf16, long long n, int C, const float* a, const float* b, const void* R, void* Z,
                               int z_bf16, void* stream);
size_t gdmae_rows_bwd_stats_workspace_bytes(int C);
int gdmae_rows_bwd_stats(const void* P, int p_bf16, const int* site, long long n, int C, const float* a, const float* b,
                         const void* dZ, int z_bf16, int z_row_elems, int col0, double* out, void* workspace, void* stream);
int gdmae_rows_bwd(const void* P, int p_bf16, const int* site, long long n, int C, const float* a, const float* b,
                   const float* c0, const float* c1, const void* dZ, int z_bf16, int z_row_elems, int col0, void* dP,
                   int dp_bf16, void* stream);

/* BatchNorm(train) bookkeeping, one launch each (network_utils.py:7-21 / spt_backbone_mae.py:30-52 BatchNorm1d/2d):
 *   gdmae_bn_fold: column statistics of x (R, C) over `count` samples (count > R when x holds only the non-zero rows
 *     of an implicit dense map) -> stats double[2C] = {mean, rstd}, ab float[2C] = {a = gamma*rstd, b = beta - a*mean},
 *     mv float[2C] = {mean, biased var}; if running_mean != NULL also the nn.BatchNorm running-statistics update
 *     (momentum, unbiased variance) and ++*num_batches.  workspace: gdmae_colstats_workspace_bytes(C).
 *   gdmae_bn_bwd_coeffs: st double[n_st*C] = column sums {dh, dh*x, (g)} of the row backward -> dgamma, dbeta
 *     (written, or added when accumulate != 0) and c01 float[2C] = {c0, c1} of dx = a*dh + c0 + c1*x.  tot (C,
 *     optional, n_st == 3): column sums of the incoming gradient over ALL sites (background share of dbeta). */
int gdmae_bn_fold(const void* x, long long R, int C, int is_bf16, double count, const float* gamma, const float* beta,
                  double eps, double momentum, float* running_mean, float* running_var, long long* num_batches,
                  double* stats, float* ab, float* mv, void* workspace, void* stream);
/* gdmae_bn_fold from partial rows part (nblk, 2, C) fp32 = {column sums, column sums of squares} per row block, left by a producer's
 * epilogue (gdmae_conv3x3_dense_stats): the same outputs without the pass over x */
int gdmae_bn_fold_partials(const float* part, int nblk, int C, double count, const float* gamma, const float* beta, double eps,
                           double momentum, float* running_mean, float* running_var, long long* num_batches, double* stats, float* ab,
                           float* mv, void* stream);
int gdmae_bn_bwd_coeffs(const double* st, int n_st, const double* stats, const float* ab, const float* gamma, int C,
                        double count, const double* tot, float* dgamma, float* dbeta, int accumulate, float* c01,
                        void* stream);
/* ... from the fp32 partial rows (nblk, n_st, C) that gdmae_rows_bwd_stats leaves in its workspace when called with out == NULL
 * (nblk = gdmae_rows_bwd_stats_rows(n)): one launch instead of a reduce + a coefficient launch. */
int gdmae_rows_bwd_stats_rows(long long n);
int gdmae_bn_bwd_coeffs_rows(const float* part, int nblk, int n_st, const double* stats, const float* ab, const float* gamma, int C,
                             double count, const double* tot, float* dgamma, float* dbeta, int accumulate, float* c01,
                             void* stream);

/* Backward of the decoder's dense 3x3 conv_out (spt_backbone_mae.py:46-52) restricted to the sites that need it:
 * out[t, k, :] = dY[site[t] - k] for the 9 taps k = (ky+1)*3 + (kx+1) (zero outside the H x W map), where the
 * output gradient dY[u] = k0 + k1 * Y[u] + (cell2pillar[u] >= 0 ? rows[cell2pillar[u]] : 0) is never materialised.
 * Y (B*H*W, C) fp32 or bf16 (out has the same type), k0/k1 (C) fp32, rows (M, C) fp32, C % 8 == 0.
 * tile_slot != NULL: Y is the tile-compact map of gdmae_conv3x3_tiles_fwd (rows of the active 8x8 tiles + the 9
 * border-class constant rows ybg for every other site); NULL: the plain dense map. */
int gdmae_conv3x3_grad_taps(const void* Y, int y_bf16, const int* tile_slot, const void* ybg, const float* k0,
                            const float* k1, const float* rows, const int* cell2pillar, const int* site, long long n, int H,
                            int W, int C, void* out, void* stream);

/* Border-region sums for the closed-form background share of that backward.  Regions 1..8 = row 0, row H-1,
 * column 0, column W-1, corners (0,0), (0,W-1), (H-1,0), (H-1,W-1) of every H x W map:
 * out[r-1][c] = sum of Y over the sites of region r; out[8+r-1][c] = sum of rows[p] over the pillars in region r. */
size_t gdmae_border_sums_workspace_bytes(int B, int C);
int gdmae_border_sums(const void* Y, int y_bf16, const int* tile_slot, const void* ybg, const float* rows,
                      const int* pillar_cell, int M, int B, int H, int W, int C, double* out, void* workspace, void* stream);

/* ---- decoder conv_out (Conv2d 3x3 pad 1, spt_backbone_mae.py:46-52,125-133) on the ACTIVE TILES of the BEV map --------
 * Replaces F.conv2d (MIOpen) + the dense 384-channel input map + the statistics pass over the dense output.
 * The input map is implicit: source stage g (128 channels, upsampling stride s_g) holds relu(a_g * P_g[row] + b_g) at the
 * sites covered by one of its tokens (P_g = ConvTranspose2d(k = s) output rows (token, dy, dx), bf16), relu(b_g) at every
 * other site.  Output: tile-compact map Yc (n_act * 64, 128) bf16 (8x8-site tiles whose one-site halo touches an active
 * site; tile_slot / tile_list from gdmae_decoder_tiles), every other site = one of the 9 border-class constants ybg.
 *   gdmae_decoder_tiles      geometry only (maps / strides: host arrays over the k <= 3 source stages); n_act device int
 *   gdmae_conv3x3_tiles_pack conv_w (128, 128 k, 3, 3) fp32 -> MFMA-fragment-ordered bf16 weights Wp, background row bgz
 *                            (128 k) bf16, class constants ybg (9, 128) bf16; b = host array of the k shift vectors
 *   gdmae_conv3x3_tiles_fwd  the bf16-MFMA implicit GEMM + fused BatchNorm2d(train) statistics over all B*H*W sites:
 *                            stats f64[256] = mean | rstd, ab f32[256] = scale | shift, mv f32[256] = mean | biased var,
 *                            running statistics updated when running_mean != NULL
 *   gdmae_tiles_gather_rows  out[i] = Y row of cell[i] = (b*H + y)*W + x   (elem_bytes 2 or 4)
 *   gdmae_tiles_to_dense     plain (B*H*W, C) copy (only for callers that want the reference's dense spatial_features) */
size_t gdmae_decoder_tiles_workspace_bytes(int B, int H, int W);
int gdmae_decoder_tiles(const int* const* maps, const int* strides, int k, int B, int H, int W, int* tile_slot, int* tile_list,
                        int* n_act, void* workspace, void* stream);
size_t gdmae_conv3x3_tiles_packed_bytes(int k);
int gdmae_conv3x3_tiles_pack(const float* conv_w, int C2, int Cin, const float* const* b, int k, void* Wp, void* bgz, void* ybg,
                             void* stream);
size_t gdmae_conv3x3_tiles_workspace_bytes(int n_act);
int gdmae_conv3x3_tiles_fwd(const void* const* P, const int* const* maps, const float* const* a, const float* const* b,
                            const int* strides, int k, const void* Wp, const void* ybg, const int* tile_list, int n_act, int B,
                            int H, int W, void* Yc, const float* gamma, const float* beta, double eps, double momentum,
                            float* running_mean, float* running_var, long long* num_batches, double* stats, float* ab,
                            float* mv, void* workspace, void* stream);
int gdmae_tiles_gather_rows(const void* Yc, const int* tile_slot, const void* ybg, const int* cell, long long n, int H, int W,
                            int C, int elem_bytes, void* out, void* stream);
int gdmae_tiles_to_dense(const void* Yc, const int* tile_slot, const void* ybg, int B, int H, int W, int C, int elem_bytes,
                         void* out, void* stream);
/* The reference's dense spatial_features (spt_backbone_mae.py:130-138) from the tile-compact bf16 conv output in one pass:
 * out (B*H*W, C) fp32 = relu(a[c] * y + b[c]) (folded BatchNorm2d), y from the active tiles or the border-class constants. */
int gdmae_tiles_to_dense_affine_relu(const void* Yc, const int* tile_slot, const void* ybg, int B, int H, int W, int C, const float* a,
                                     const float* b, float* out, void* stream);

/* The three row kernels above also serve BatchNorm1d + ReLU of the DynVFE point MLP (site = NULL: identity rows,
 * Z/dZ = a plain (n, C) matrix with z_row_elems = C, col0 = 0).  Fused DynVFE tail (dyn_vfe.py:107-109):
 *   gdmae_segment_max_affine: out[p,c] = max_{i in pillar p} relu(a_c x[i,c] + b_c), arg = arg-max point id
 *   gdmae_segmax_bwd_stats:   sums double[2C] = column sums of {dh, dh*x}, dh = dout*[arg==i]*[out>0]
 *   gdmae_segmax_bn_bwd:      dx[i,c] = a_c dh[i,c] + c0_c + c1_c x[i,c]  (BatchNorm chain rule, all N points) */
int gdmae_segment_max_affine(const void* x, int x_bf16, const int* pillar_pt_off, const int* pillar_pts, int M, int C,
                             const float* a, const float* b, float* out, int* arg, void* stream);
int gdmae_segmax_bwd_stats(const void* x, int x_bf16, const float* out, const int* arg, const float* dout, long long M,
                           int C, const float* a /* optional: the forward's affine, lets x = (out-b)/a */,
                           const float* b, double* sums, void* workspace, void* stream);
int gdmae_segmax_bn_bwd(const void* x, int x_bf16, const float* out, const int* arg, const float* dout,
                        const int* inverse32, long long N, int C, const float* a, const float* c0, const float* c1,
                        void* dx, int dx_bf16, void* stream);

/* ---- a11, a13: windowed cosine attention ------------------------------------------------------ *
 * Replaces flat2window_v2/window2flat_v2 (sst_utils.py:107-180), WindowAttention.forward
 * (pcdet/models/model_utils/sst_basic_block.py:22-54) and _scaled_cosine_attention
 * (pcdet/models/model_utils/cosine_msa.py:114-176) for ONE occupancy level (T = 16/32/64 padded tokens).
 * qk (Ms, 2d): projected queries [0,d) and keys [d,2d); v (Ms, d); out (Ms, d) written at token rows;
 * io_bf16 selects fp32 (0) or bf16 (1) rows in HBM - arithmetic is fp32 in registers either way.
 * Backward: dqk (Ms,2d), dv (Ms,d), dtau_part: n_win*H floats (partials of d loss / d clamp(tau)).
 * bf16 rows run on the bf16 matrix cores at every level: v_mfma_f32_16x16x{16,32}_bf16 with one wavefront per
 * window x 4 heads at T = 16 (attention_t16.hip), v_mfma_f32_32x32x16_bf16 at T = 32 (one wavefront per window x head)
 * and T = 64 (two wavefronts per window x head) (attention_t32.hip); logits come from the raw bf16 rows (exact products,
 * fp32 accumulation) and are normalised afterwards.  fp32 rows: exact fp32 v_mfma_f32_32x32x2_f32 for T >= 32
 * (attention_mfma.hip), lane-per-query VALU kernel for T = 16.  gdmae_set_attention_impl: 0 = that default, 1 = VALU
 * kernels everywhere, 2 = fp32 MFMA (T >= 32) / VALU (T = 16) also for bf16 rows (A/B tests). */
int gdmae_set_attention_impl(int impl);
int gdmae_window_attention_fwd(const void* qk, const void* v, void* out, int io_bf16, const int* csr_tok,
                               const int* win_start, const int* win_len, int n_win, int T, int d, int H,
                               const float* tau, float tau_min, void* stream);
int gdmae_window_attention_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, int io_bf16,
                               float* dtau_part, const int* csr_tok, const int* win_start, const int* win_len,
                               int n_win, int T, int d, int H, const float* tau, float tau_min, void* stream);
/* All occupancy levels of one shift in one call (what the layer executor issues): the windows of level l are
 * win_start / win_len [sum_{k<l} n_win[k], ...), max_tokens[l] its padded token count; dtau_part holds
 * sum_l n_win[l] * H partial slots, level after level.  bf16 rows with levels of 16 / 32 / 64 tokens go out as ONE launch per direction
 * (csrc/attention_coop.hip), everything else level by level through the functions above.
 * lse (forward, optional): (n_tok, H) fp32, receives log2 sum_k exp(logit) of every row of the T = 32 / 64 levels; the backward takes it
 * back together with the forward's output rows `out` (both optional there: without them the backward re-derives the softmax statistics on
 * the one-wavefront-per-(window, head) kernels). */
/* 1 when the forward below, called with these arguments under the current gdmae_set_attention_impl, writes `lse`: only then may `out` /
 * `lse` be handed to the backward (max_tokens: HOST array of n_levels ints) */
int gdmae_window_attention_levels_writes_lse(int io_bf16, int n_levels, const int* max_tokens, int d, int H);
int gdmae_window_attention_levels_fwd(const void* qk, const void* v, void* out, int io_bf16, const int* csr_tok,
                                      const int* win_start, const int* win_len, int n_levels, const int* n_win,
                                      const int* max_tokens, int d, int H, const float* tau, float tau_min, float* lse, void* stream);
int gdmae_window_attention_levels_bwd(const void* qk, const void* v, const void* dout, void* dqk, void* dv, int io_bf16,
                                      float* dtau_part, const int* csr_tok, const int* win_start, const int* win_len,
                                      int n_levels, const int* n_win, const int* max_tokens, int d, int H, const float* tau,
                                      float tau_min, const void* out, const float* lse, void* stream);
/* Measurement hook (bench.py roofline leg): HIP-event brackets around the two all-levels entries on the stream they launch on.
 * gdmae_attention_timing(1) starts collecting (dropping earlier records), (0) stops; gdmae_attention_timing_read returns the
 * summed milliseconds and the number of calls of the forward (which = 0) / backward (which = 1) entry. */
int gdmae_attention_timing(int on);
int gdmae_attention_timing_read(int which, double* total_ms, long long* calls);
/* Generalisation (round 3): measurement slots for every instrumented kernel family of the PRODUCT path - the brackets sit
 * inside the native layer / stage executors, so what is timed is exactly what the training step launches.  Slots
 * (gdmae_kernel_timing_name): 0 k_win_attn_fwd, 1 k_win_attn_bwd (all-levels entries), 2 k_tok_gemm (every fused token GEMM
 * launch incl. k_tok_gemm_multi), 3 k_dw_grouped, 11 k_layer_tail, ...  gdmae_kernel_timing(1) starts collecting (dropping
 * earlier records), (0) stops; gdmae_kernel_timing_read returns summed milliseconds, call count and the summed ALGORITHMIC
 * bytes / flops of the bracketed launches (operands read once + results written once, stated next to each bracket). */
int gdmae_kernel_timing(int on);
int gdmae_kernel_timing_slots(void);
const char* gdmae_kernel_timing_name(int slot);
int gdmae_kernel_timing_read(int slot, double* total_ms, long long* calls, double* bytes, double* flops);
/* For the fused layer launches (csrc/layer_fused.hip) `bytes` counts the bf16 operand / result rows and the weight images only;
 * what those launches move besides (fp32 statistics rows, per-workgroup partial rows, fp32 rows at the stage boundary, the
 * y + pos copy of a layer output) is summed here. */
int gdmae_kernel_timing_read_side(int slot, double* side_bytes);
/* out[0] = sum(term) / sum(weights), out[1] = 1 / sum(weights) (both 0 when no weight is positive): the weighted mean
 * that finishes pytorch3d.loss.chamfer_distance (spt_backbone_mae.py:83-89), one single-workgroup launch. */
int gdmae_weighted_mean_finish(const float* term, const float* weights, long long n, float* out, void* stream);
int gdmae_sum_partials(const float* part, long long n, float scale, float* out, int accumulate, void* stream);
int gdmae_sum_partials_gated(const float* part, long long n, float scale, float* out, const float* gate, float gate_min,
                             void* stream); /* out = gate[0] >= gate_min ? scale*sum : 0 (clamp(tau) gradient) */

/* ---- a14: fused residual add + LayerNorm ------------------------------------------------------- *
 * Replaces `src = src + src2; src = self.normN(src)` of EncoderLayer.forward
 * (pcdet/models/model_utils/sst_basic_block.py:77-84; nn.LayerNorm(d), eps 1e-5): y = LN(a + b) * gamma + beta,
 * a fp32 (n,d), b fp32 or bf16 (n,d), d in {64,128,256}; stats (n,2) = per-row mean, rstd (saved for backward).
 * Backward: dx (n,d) = gradient w.r.t. (a + b) (goes to both), dgamma_dbeta (2d) = {dgamma, dbeta}. */
size_t gdmae_add_layernorm_workspace_bytes(int d);
int gdmae_add_layernorm_fwd(const float* a, const void* b, int b_is_bf16, const float* gamma, const float* beta,
                            long long n, int d, float eps, float* y, float* stats, void* y_bf16 /* optional copy */,
                            void* stream);
/* dy2 (optional, fp32/bf16): second upstream gradient summed on load; dx_bf16 (optional): bf16 copy of dx;
 * sums (3d) = {dgamma, dbeta, column sums of dx (= bias gradient of the GEMM that produced b)}. */
int gdmae_add_layernorm_bwd(const float* a, const void* b, int b_is_bf16, const float* gamma, const float* stats,
                            const float* dy, const void* dy2, int dy2_bf16, long long n, int d, float* dx, void* dx_bf16,
                            float* sums, void* workspace, void* stream);
/* gdmae_prep_tokens: x_out = x, xpos_out = x + pos_table[tok_pos] in the GEMM input dtype (out_bf16; with fp32 only
 * xpos_out is written) - the q/k input of WindowAttention.forward (sst_basic_block.py:44-49).
 * gdmae_add3: out(fp32) = a(fp32) + b + c, b/c optional fp32 or bf16 (gradient accumulation of the residual stream). */
int gdmae_prep_tokens(const float* x, const float* pos_table, const int* tok_pos, long long n, int d, void* x_out,
                      void* xpos_out, int out_bf16, void* stream);
/* out (total bf16 elements) = sum of k <= 8 bf16 buffers, fp32 accumulation, one rounding: the gradient of a map with several
 * consumers in one pass (gdmae_hip.ops.FanOut; reference center_head.py:26-45 feeds one map to every branch of SeparateHead). */
int gdmae_sum_bf16(const void* const* src, int k, long long total, void* out, void* stream);
int gdmae_add3(const float* a, const void* b, int b_bf16, const void* c, int c_bf16, long long total, float* out, void* stream);
/* the same with the result in fp32 (out_bf16 = 0) or bf16 (1): the block residual feat + out of SSTBlockV1 (spt_backbone.py:158)
 * goes to the next sparse convolution, which rounds its input rows to bf16 anyway - written in bf16 the 9 gathers of every row
 * move half the bytes */
int gdmae_add3_to(const float* a, const void* b, int b_bf16, const void* c, int c_bf16, long long total, void* out, int out_bf16,
                  void* stream);

/* ---- library GEMMs (hipBLASLt, one cached algorithm per shape bucket) ------------------------------ *
 * The token / point / site GEMMs of the path outside the encoder-layer executor (nn.Linear of DynVFE, the
 * spconv im2col products, the deconvolution token GEMM, their gradients); row-major operands.
 *   gdmae_gemm: C (M,N) = op(A) op(B) + bias(N); A is (M,K), or (K,M) with trans_a; B is (K,N), or (N,K) with
 *     trans_b; A/B bf16 (ab_bf16) or fp32; C has the operand type, or fp32 if c_f32; bias (optional) in C's type.
 *   gdmae_gemm_tn_splitk: C (m,n) fp32 (+)= A^T B for A (K,m), B (K,n) with the long K dimension (20 k ... 1.4 M
 *     rows) split into equal slices = one batched GEMM + a fixed-order reduction (weight gradients).
 * workspace: gdmae_gemm_workspace_bytes() / gdmae_gemm_tn_splitk_workspace_bytes(K, m, n). */
/* calls[0] = library GEMM calls so far, calls[1] = algorithm plans created so far (first use of a shape bucket = candidate timing
 * with stream synchronisation): a loop has reached its steady state once calls[1] stops growing. */
int gdmae_gemm_stats(long long* calls);
/* Algorithm selection of the hipBLASLt front end (bf16 operands; the op-by-op reference paths of the tests): 0 = the heuristic's first
 * algorithm (deterministic across processes and call orders), 1 = time the 16 best at the first use of a shape (default), 2 = time
 * every supporting algorithm, -1 = the GDMAE_GEMM_TUNE environment default.  Clears the plan cache. */
int gdmae_gemm_tuning(int mode);
size_t gdmae_gemm_workspace_bytes(void);
int gdmae_gemm(const void* A, const void* B, void* C, long long M, long long N, long long K, int trans_a, int trans_b,
               int ab_bf16, int c_f32, const void* bias, void* workspace, void* stream);
size_t gdmae_gemm_tn_splitk_workspace_bytes(long long K, int m, int n);
int gdmae_gemm_tn_splitk(const void* A, const void* B, float* C, long long K, int m, int n, int ab_bf16, int accumulate,
                         void* workspace, void* stream);

/* Hand-written bf16-MFMA "TN" product for the token-contracted weight gradients (dw_grouped.hip; backward of the nn.Linear
 * layers of sst_basic_block.py:57-84 / cosine_msa.py): dW (M, N) fp32 = G^T X for G (rows, M), X (rows, N) bf16 row-major,
 * dbias (M) fp32 = column sums of G (optional, may be NULL).  M, N multiples of 128, rows a multiple of 512 (callers
 * zero-pad).  Deterministic (row slices summed in a fixed order).  The encoder-layer executor issues the five weight
 * gradients of a layer through the same kernel as ONE launch. */
size_t gdmae_dw_gemm_workspace_bytes(long long rows, int M, int N);
int gdmae_dw_gemm(const void* G, const void* X, long long rows, int M, int N, float* dW, float* dbias, void* workspace,
                  void* stream);

/* ---- a6 as one call: sparse conv (k3) -> BatchNorm1d(train) -> ReLU block -------------------------- *
 * post_act_block (spconv_utils.py:37-56; conv_down / conv_out of SSTBlockV1, spt_backbone.py:206,217,256-263).
 * x (n_in, cin) rows in the compute dtype (bf16 != 0: bf16, else fp32) or fp32 with x_f32 (cast folded into the
 * gather); nbr (n_out, 9) / nbr_t (n_in, 9): rulebook and transposed rulebook (gdmae_rulebook); W (cout, 9*cin) in the
 * compute dtype (spconv-2.x layout (cout, 3, 3, cin)).  cols / y / stats / ab / mv are written by the forward and read
 * by the backward.  g: upstream gradient (n_out, cout), compute dtype or fp32 (g_f32).  dW / dgamma / dbeta are
 * ACCUMULATED; dx may be NULL.  running_* may be NULL (no running-statistics update). */
typedef struct gdmae_conv_block_args {
  long long n_in, n_out;
  int cin, cout, bf16, x_f32, g_f32;
  float eps, momentum;
  const void* x;
  const int* nbr;
  const int* nbr_t;
  const void* W;
  const float *gamma, *beta;
  float *running_mean, *running_var;
  long long* num_batches;
  void* cols;      /* (n_out, 9*cin) compute dtype */
  void* y;         /* (n_out, cout) conv output before BatchNorm */
  double* stats;   /* [2*cout] mean | rstd */
  float* ab;       /* [2*cout] folded affine */
  float* mv;       /* [2*cout] mean | biased var */
  void* out;       /* (n_out, cout) compute dtype                  [forward]  */
  const void* g;   /*                                              [backward] */
  void* dx;        /* (n_in, cin) compute dtype                    [backward] */
  float *dW, *dgamma, *dbeta;
  void* scratch;   /* gdmae_conv_block_scratch_bytes */
  /* round 3, bf16 with 128 / 256 channels: packed weight images (gdmae_spconv_pack_jobs + gdmae_tok_gemm_pack) select the
   * im2col-free path - the convolution and its input gradient are implicit GEMMs over the rulebook (gdmae_spconv), the weight
   * gradient one grouped TN launch that gathers the input rows on load; `cols` is then neither written nor read (may be NULL) */
  const void* packed_fwd;   /* 9 x (cout, cin) images */
  const void* packed_bwd;   /* 9 x (cin, cout) images (per-tap transposed weights) */
  int out_f32;              /* forward: write `out` in fp32 although the block computes in bf16 (the consumer is the fp32 residual
                               stream of an encoder stage: saves its cast pass) */
} gdmae_conv_block_args;
size_t gdmae_conv_block_scratch_bytes(long long n_in, long long n_out, int cin, int cout, int bf16);
/* Sparse convolution as an implicit GEMM over a rulebook (csrc/spconv.hip): Y (n, cout) bf16 = sum_tap W_tap X[nbr[:, tap]], X bf16
 * or fp32 (x_f32) rows, nbr (n, 9) with negative entries for missing taps; cin, cout in {128, 256}.  packed: 9 per-tap
 * (cout, cin) bf16 images in MFMA-fragment order (gdmae_spconv_packed_bytes), produced by gdmae_tok_gemm_pack from the job table
 * gdmae_spconv_pack_jobs writes (9 x 6 int64; transposed = 1: the (cin, cout) images of the input-gradient convolution, to be
 * used with the transposed rulebook and cin / cout swapped).  Replaces spconv's gather-GEMM-scatter (spconv_utils.py:37-56). */
size_t gdmae_spconv_packed_bytes(int cin, int cout);
int gdmae_spconv_pack_jobs(const float* W /* (cout, 3, 3, cin) fp32 */, int cin, int cout, int transposed, void* packed, long long* jobs);
int gdmae_spconv(const void* X, int x_f32, const int* nbr, const void* packed, long long n, int cin, int cout, void* Y,
                 int timing_slot /* 0: the sparse-conv forward slot of gdmae_kernel_timing */, void* stream);
/* The same launch with the BatchNorm statistics of Y as its epilogue: part (ceil(n / gdmae_spconv_stat_rows(cin, cout, x_f32)), 2, cout)
 * fp32 = per-workgroup column sums of Y and Y^2 (of the bf16-rounded values), to be folded by gdmae_bn_fold_partials - what
 * gdmae_conv_block_fwd does instead of a statistics pass over Y (post_act_block's BatchNorm1d, spconv_utils.py:37-56). */
int gdmae_spconv_stat_rows(int cin, int cout, int x_f32);
int gdmae_spconv_stats(const void* X, int x_f32, const int* nbr, const void* packed, long long n, int cin, int cout, void* Y, float* part,
                       void* stream);
/* ---- 3x3 conv_out backward of the generative decoder without the tap matrix (round 3; spt_backbone_mae.py:46-52 backward) ---- *
 * gdmae_decoder_dy: dYc (n_act * 64, C) bf16 = k0 + k1 * Yc + rows[pillar of the site] on the active tiles (Yc / tile_list of
 *   gdmae_conv3x3_tiles_fwd / gdmae_decoder_tiles; rows (M, C) fp32 = the sparse part of the output gradient; zero at the
 *   out-of-map sites of edge tiles) - the conv output gradient, materialised ONCE instead of nine-fold (gdmae_conv3x3_grad_taps).
 * gdmae_decoder_site_rulebook: nbr[t, k] = row of dYc holding site[t] - (ky - 1, kx - 1), k = 3 ky + kx, -1 outside the map;
 *   site = full-resolution cells ((b H + y) W + x) of a source stage's active sites, sites_per_tok of them per token, token count on
 *   the device (n_dev).  Geometry only: gdmae_geometry_plan builds it as "dec.nbr<g>".
 * gdmae_tap_dw: out[k][n][m_off + m] += sum_t G[t][m] X[nbr[t][k]][n] for the 9 taps as ONE grouped TN launch + a fixed-order
 *   reduce (G (n_pad = gdmae_tap_dw_rows(n, M, N) >= n rows, M) bf16 contiguous; X rows (.., N) bf16 gathered through nbr (n, 9));
 *   with gdmae_spconv over the same rulebook for the input gradient this replaces the taps + two library GEMMs per stage. */
int gdmae_decoder_dy(const void* Yc, const int* tile_list, int n_act, const float* k0, const float* k1, const float* rows,
                     const int* cell2pillar, int H, int W, int C, void* dYc, void* stream);
int gdmae_decoder_site_rulebook(const int* site, const int* n_dev, int sites_per_tok, long long cap_sites, const int* tile_slot,
                                int H, int W, int* nbr, void* stream);
long long gdmae_tap_dw_rows(long long n, int M, int N);      /* n_pad: rows G must be allocated for */
size_t gdmae_tap_dw_workspace_bytes(long long n, int M, int N);
int gdmae_tap_dw(const void* G, long long n, long long n_pad, int M, const void* X, const int* nbr, int N, float* out, int ld_out,
                 int m_off, void* workspace, void* stream);
int gdmae_conv_block_fwd(const gdmae_conv_block_args* args /* host */, void* stream);
int gdmae_conv_block_bwd(const gdmae_conv_block_args* args /* host */, void* stream);

/* ---- a13/a14 as one call: native executor of a whole encoder layer ------------------------------ *
 * EncoderLayer.forward (sst_basic_block.py:77-84; WindowAttention :22-54; cosine_msa.py): q = k = x + pos, v = x,
 * in-projection, windowed cosine attention, out-projection, LN(x + attn), FFN(GELU erf), LN(x + ffn) - forward or
 * backward enqueued by ONE call (hipBLASLt GEMMs with cached algorithms + the kernels above).  All pointers are
 * device pointers unless noted; weights and biases are in the GEMM dtype (bf16 != 0: bf16, else fp32), LayerNorm
 * parameters, tau, x, y, dy, dx and every parameter gradient are fp32.  Parameter gradients are ACCUMULATED (+=).
 * `saved` (gdmae_encoder_layer_bytes: saved_bytes) is written by the forward and read by the backward; `scratch`
 * needs fwd_scratch_bytes / bwd_scratch_bytes. */
typedef struct gdmae_layer_args {
  long long n;                 /* tokens */
  int d, ff, nhead, bf16;
  float eps, tau_min;
  int n_levels;                /* window-size levels of the partition (<= 4) */
  int n_win[4], max_tokens[4];
  const int* tok_pos;          /* (n) row of pos_table per token */
  const int* csr_tok;          /* tokens ordered by (level, window) */
  const int* win_start;
  const int* win_len;          /* per window, levels concatenated */
  const float* pos_table;      /* (window cells, d) */
  const void *Win, *bin, *Wo, *bo, *W1, *b1, *W2, *b2;
  const float *g1, *be1, *g2, *be2, *tau;
  const float* x;              /* (n, d) layer input */
  float* y;                    /* (n, d) layer output            [forward]  */
  const float* dy;             /* (n, d)                         [backward] */
  float* dx;                   /* (n, d)                         [backward] */
  float *dWin, *dbin, *dtau, *dWo, *dbo, *dW1, *db1, *dW2, *db2, *dg1, *dbe1, *dg2, *dbe2;   /* [backward] */
  void* saved;
  void* scratch;
  /* bf16 mode, optional: the layer's weights packed in MFMA-fragment order by gdmae_tok_gemm_pack (layout:
   * gdmae_layer_packed_bytes / gdmae_layer_pack_jobs).  Non-NULL: the token GEMMs that are not weight gradients run on
   * the library's own fused kernels (gdmae_tok_gemm: bias + GELU, residual + LayerNorm, GELU backward in the epilogue)
   * instead of hipBLASLt + separate row kernels. */
  const void* packed;
  /* Fused stage path only (gdmae_encoder_stage_fused() == 1), optional: the block residual around the stage (SSTBlockV1.forward,
   * spt_backbone.py:219-264: the stage output is added to the stage input before conv_out) folded into the stage's own launches,
   * all rows bf16.  Forward: layers[0].x_bf16 = 1 (x holds bf16 rows) and layers[n_layers-1].res_out = (n, d) bf16 receives
   * x + y instead of y.  Backward: layers[n_layers-1].dres = (n, d) bf16 gradient of res_out (read instead of dy; it is also the
   * gradient of the skip path) and layers[0].dx_bf16 = (n, d) bf16 receives dres + the gradient through the stage instead of dx. */
  int x_bf16;
  void* res_out;
  const void* dres;
  void* dx_bf16;
} gdmae_layer_args;
/* Packed weight image of one layer (bf16 mode): forward operands [Win(q,k rows) | Win(v rows) | Wo | W1 | W2] followed by
 * the transposed operands of the input-gradient products [W2^T | W1^T | Wo^T | Win(q,k)^T | Win(v)^T].
 * gdmae_layer_pack_jobs fills a HOST table of gdmae_layer_pack_job_count(d, ff) x 6 int64 {src, dst, M, K, ld, flags} for gdmae_tok_gemm_pack
 * (copy it to the device; one launch packs any number of layers). */
size_t gdmae_layer_packed_bytes(int d, int ff);
int gdmae_layer_pack_jobs(const float* Win, const float* Wo, const float* W1, const float* W2, int d, int ff, void* packed,
                          long long* jobs_host /* 6 x gdmae_layer_pack_job_count(d, ff) */);
/* Jobs gdmae_layer_pack_jobs writes: the ten images above plus, for d in {128, 256} and ff = 2 d, the weight STREAM of the
 * in-register forward launch (csrc/layer_v3.hip: Wo, then per 128-channel hidden chunk W1[chunk rows] and W2[:, chunk columns] as
 * 16 x 32 matrix-core fragments in the order the launch consumes them; EncoderLayer's linears, sst_basic_block.py:57-84). */
int gdmae_layer_pack_job_count(int d, int ff);
int gdmae_tok_gemm_pack(const long long* jobs_dev, int n_jobs, void* stream);
/* Y = epilogue(X Wp^T + bias): X (n_pad, K) bf16 rows (n_pad % 64 == 0), Wp = packed (N, K) weights, bias (N) bf16 or
 * NULL; (K, N) in {128, 256} x {128, 256}, (256, 512), (512, 256).  epilogue 0: out0 = . (bf16); 1: out0 = h = . and
 * out1 = gelu_erf(h); 2: out0 = . * gelu'(aux) (aux = h), and out1 = gelu_erf(aux) when out1 is given; 3: y = LayerNorm(res + bf16(.)) fp32 (first n rows), stats
 * (n, 2) = mean | rstd, optional bf16 copies y_bf16 = y and ypos_bf16 = y + pos_table[tok_pos[row]]. */
int gdmae_tok_gemm(const void* X, const void* Wp, const void* bias, long long n, long long n_pad, int K, int N, int epilogue,
                   void* out0, void* out1, const void* aux, const float* res, const float* gamma, const float* beta, float eps,
                   float* y, float* stats, void* y_bf16, const float* pos_table, const int* tok_pos, void* ypos_bf16,
                   void* stream);

/* The packed in-projection of cosine_msa.py / sst_basic_block.py:22-54 (q = k = x + pos, v = x) as ONE launch of three plain
 * products over shared row tiles:  qk (n_pad, 2d) = Xpos Wqk^T + bias3[0:2d],  v (n_pad, d) = X Wv^T + bias3[2d:3d];
 * Wp_qk = packed (2d, d) image of in_proj_weight[0:2d], Wp_v = packed (d, d) image of in_proj_weight[2d:3d], bias3 (3d) bf16
 * or NULL, d in {128, 256}.  Bit-identical to two gdmae_tok_gemm calls with epilogue 0. */
int gdmae_tok_gemm_qkv(const void* Xpos, const void* X, const void* Wp_qk, const void* Wp_v, const void* bias3, long long n_pad,
                       int d, void* qk, void* v, void* stream);

/* The feed-forward block of a layer as ONE launch (tok_gemm.hip k_tok_ffn; sst_basic_block.py:79-84 linear1 -> GELU -> linear2 ->
 * residual + LayerNorm 2):  h (n_pad, 2 d) bf16 = X W1^T + b1 is stored (the backward differentiates the GELU at it), gelu(h)
 * stays in LDS as the operand tile of the second product (gdmae_tok_gemm epilogue 2 re-creates it through out1 for the weight
 * gradient), then every output of epilogue 3: y, stats, y_bf16 / ypos_bf16 / f_out (optional).  W1p = packed (2 d, d) image,
 * W2p = packed (d, 2 d) image, d in {128, 256}.  Bit-identical to epilogue 1 followed by epilogue 3. */
int gdmae_tok_gemm_ffn(const void* X, const void* W1p, const void* b1, const void* W2p, const void* b2, long long n, long long n_pad,
                       int d, void* h, const float* res, const float* gamma, const float* beta, float eps, float* y, float* stats,
                       void* y_bf16, const float* pos_table, const int* tok_pos, void* ypos_bf16, void* f_out, void* stream);

/* Input-gradient token GEMM fused with the backward of the post-norm it feeds (tok_gemm.hip, epilogue LN_BWD; backward of
 * sst_basic_block.py:57-84 `src = norm(src + dropout(src2))`):  g = dy + [dy2] + bf16(X Wp^T) is the gradient of
 * LN(ln_a + ln_b); dx (n, N) fp32 and its optional bf16 copy dx_bf16 (n_pad, N) are the LayerNorm backward of g with the
 * saved (mean, rstd) rows `stats`; part (n_pad / gdmae_tok_gemm_ln_bwd_rows(N), 3, N) fp32 receives one partial row of
 * dgamma / dbeta / column sums of dx per workgroup (summed by the caller in a fixed order).  dy2 (bf16) may be NULL. */
int gdmae_tok_gemm_ln_bwd_rows(int N);
int gdmae_tok_gemm_ln_bwd(const void* X, const void* Wp, long long n, long long n_pad, int K, int N, const float* dy,
                          const void* dy2_bf16, const float* ln_a, const void* ln_b_bf16, const float* stats,
                          const float* gamma, float* dx, void* dx_bf16, float* part, void* stream);
int gdmae_encoder_layer_bytes(long long n, int d, int ff, int nhead, int bf16, size_t* saved_bytes,
                              size_t* fwd_scratch_bytes, size_t* bwd_scratch_bytes);   /* depend on n only through ceil(n / 2048) */
int gdmae_encoder_layer_fwd(const gdmae_layer_args* args /* host */, void* stream);
int gdmae_encoder_layer_bwd(const gdmae_layer_args* args /* host */, void* stream);
/* n_layers consecutive layers of one stage in one call (BasicShiftBlockV2 x NUM_BLOCKS, sst_basic_block.py:100-114):
 * layers[i+1].x == layers[i].y, same n / d / ff / dtype.  In bf16 mode the q/k and v inputs of layers 1.. are written
 * by the previous layer's second LayerNorm, and in the backward (all layers sharing ONE scratch buffer) the three
 * pieces of a layer's input gradient are summed on load by the previous layer instead of being added and re-read.
 * Backward: layers[n_layers-1].dy = upstream gradient, layers[0].dx = gradient of the stage input. */
int gdmae_encoder_stage_fwd(const gdmae_layer_args* layers /* host array */, int n_layers, void* stream);
int gdmae_encoder_stage_bwd(const gdmae_layer_args* layers /* host array */, int n_layers, void* stream);
/* Which launch sequence the stage entry points use for bf16 rows with packed weights (d in {128, 256}, ff = 2 d):
 *   1  "layer around its bytes" (csrc/layer_fused.hip): per layer and direction THREE launches around the attention
 *      (forward: out-projection + LayerNorm 1 + feed-forward block + LayerNorm 2; backward: feed-forward block + LayerNorm 1 +
 *      out-projection, and in-projection + LayerNorm 2 of the layer below), bf16 residual stream inside the stage: 60 d bytes per
 *      token and layer instead of 106 d.  Only layers[n_layers-1].y (fp32) is written; the y of the other layers is not.
 *   0  one launch per product with fused row epilogues and an fp32 residual stream (csrc/tok_gemm.hip), every layers[i].y written.
 *  -1  the default: 1 unless the environment variable GDMAE_LAYER_V2 is 0. */
int gdmae_encoder_set_layer_path(int path);
/* 1 when gdmae_encoder_stage_fwd / _bwd would take the fused path (1 above) for these layers, else 0 */
int gdmae_encoder_stage_fused(const gdmae_layer_args* layers /* host array */, int n_layers);

/* ---- a16: the decoder's ConvTranspose2d(k = s, stride s, no bias) blocks on token rows (csrc/rows_gemm.hip) ----------------- *
 * Reference: spt_backbone_mae.py:30-45 `decoder_deblocks` (applied to the densified maps at :125-131).  With kernel = stride the
 * outputs of an input site do not overlap: P (n, s*s*cout) = X (n, cin) Wm, Wm[ci][(dy*s+dx, c)] = weight[ci][c][dy][dx]; row
 * (token, dy*s+dx) of P viewed as (n*s*s, cout) is the deconvolution output at the full-resolution site of (token, dy, dx).
 * cin in {128, 256}, cout = 128, s in {1, 2, 4}; X / P / dP / dX bf16 rows (n is NOT padded: loads are guarded), weight fp32
 * (cin, cout, s, s).  gdmae_deconv_rows_pack writes both MFMA-fragment-ordered images (gdmae_deconv_rows_packed_bytes each) in
 * one launch; _bwd_weight ACCUMULATES X^T dP into `dW` in the weight's own layout (fixed summation order). */
size_t gdmae_deconv_rows_packed_bytes(int cin, int cout, int s);
int gdmae_deconv_rows_pack(const float* weight, int cin, int cout, int s, void* packed_fwd, void* packed_bwd, void* stream);
int gdmae_deconv_rows_fwd(const void* X, long long n, int cin, int cout, int s, const void* packed_fwd, void* P, void* stream);
int gdmae_deconv_rows_bwd_input(const void* dP, long long n, int cin, int cout, int s, const void* packed_bwd, void* dX, void* stream);
size_t gdmae_deconv_rows_dw_workspace_bytes(long long n, int cin, int cout, int s);
int gdmae_deconv_rows_bwd_weight(const void* X, const void* dP, long long n, int cin, int cout, int s, float* dW, void* workspace,
                                 void* stream);

/* The prediction head nn.Linear(128 -> n_out) on the fp32 decoder rows of all pillars (spt_backbone_mae.py:52,74 `decoder_pred`;
 * n_out = 48 = 16 points x 3; 8 <= n_out <= 64, n_out % 8 == 0).  The forward is fp32-accurate on the bf16 matrix cores: X and W are
 * split into their bf16 rounding and the bf16 rounding of the remainder, X W^T = Xh Wh + Xl Wh + Xh Wl in fp32 accumulators (the Chamfer
 * loss squares these offsets: bf16-rounded outputs bias it by 1 - 4e-4 relative); Y_f32 (n, n_out) receives the accumulated values,
 * Y (n, n_out) bf16 their rounding.  Backward on bf16 operands (X_bf16 = bf16(X), bf16(dY)): dX (n, 128) fp32 (optional), dW (n_out, 128)
 * and db (n_out) fp32 ACCUMULATED in a fixed order.  packed: gdmae_pred_head_packed_bytes(), refreshed by gdmae_pred_head_pack whenever the weights change. */
size_t gdmae_pred_head_packed_bytes(void);
int gdmae_pred_head_pack(const float* weight, const float* bias, int n_in, int n_out, void* packed, void* stream);
int gdmae_pred_head_fwd(const float* X, long long n, int n_out, const void* packed, void* Y, void* X_bf16 /* (n, 128) bf16 out: the rounded
                        operand rows, operand of the weight gradient */, float* Y_f32 /* optional (n, n_out) fp32 result; Y itself may then be NULL */,
                        void* stream);
size_t gdmae_pred_head_bwd_workspace_bytes(long long n);
int gdmae_pred_head_bwd(const void* dY, int dy_f32 /* dY holds fp32 rows: rounded here, the rounded rows written to dY_bf16 */,
                        void* dY_bf16 /* (n, n_out) bf16 scratch, dy_f32 only */,
                        const float* scale_a, const float* scale_b /* optional device scalars (dy_f32 only): dY is multiplied by
                        scale_a[0] * scale_b[0] on load - the loss's upstream gradient and 1 / sum of weights of the Chamfer mean */,
                        const void* X_bf16, long long n, int n_out,
                        const void* packed, float* dX, float* dW, float* db, void* workspace, void* stream);

/* ---- f1: dense 3x3 convolution of channels-last bf16 maps (fine-tune detector) ----------------- *
 * Conv2d(cin, cout, 3, stride 1, padding = dilation in {1, 2}) on (B, H, W, C) bf16 maps on the bf16 matrix cores: forward, input
 * gradient (the same launch on the transposed, tap-flipped image) and weight gradient.  Replaces F.conv2d -> MIOpen at
 * pcdet/models/backbones_2d/sst_bev_backbone.py:14-19,34-40 (SSTBEVBackbone), pcdet/models/dense_heads/center_head.py:20-35,96-104
 * (SeparateHead, shared_conv) and pcdet/models/backbones_3d/spt_backbone.py:282-303 (conv_out of the dense decoder).
 * Channel counts of a LAUNCH (cin_l, cout_l) are multiples of 32: a layer's cin / cout padded with zero weights (pack) and zero
 * channels (maps); dilation 2 needs cin_l % 64 == 0 and cout_l % 64 == 0.  csrc/conv_dense.hip. */
size_t gdmae_conv3x3_dense_packed_bytes(int cin, int cout);
/* weight (cout, cin, 3, 3) fp32 -> fragment-ordered bf16 image; transposed = 1: the image of the input-gradient launch
 * (cout_pad -> cin_pad channels).  Refresh whenever the weight changes. */
int gdmae_conv3x3_dense_pack(const float* weight, int cin, int cout, int dil, int transposed, void* packed, void* stream);
/* Y (B, H, W, cout_l) bf16 = conv(X (B, H, W, cin_l) bf16) + bias (cout_l fp32, optional) */
int gdmae_conv3x3_dense(const void* X, int B, int H, int W, int cin_l, int cout_l, int dil, const void* packed, const float* bias,
                        void* Y, void* stream);
/* ... + addend (B, H, W, cout_l) bf16 added to the rounded result in the store pass: the gradient an identity shortcut carries to a
 * block's input joins the convolution's input gradient (sst_bev_backbone.py:36-40) */
int gdmae_conv3x3_dense_add(const void* X, int B, int H, int W, int cin_l, int cout_l, int dil, const void* packed, const float* bias,
                            const void* addend, void* Y, void* stream);
/* ... + the statistics of the BatchNorm that follows the convolution as its epilogue: stat_rows (gdmae_conv3x3_dense_stat_rows() = 256,
 * 2, cout_l) fp32 partial rows {sum, sum of squares} per channel of the ROUNDED outputs over all B H W sites, summed in a fixed order;
 * gdmae_bn_fold_partials turns them into the folded affine (no pass over Y).  workspace: gdmae_conv3x3_dense_stats_workspace_bytes. */
size_t gdmae_conv3x3_dense_stats_workspace_bytes(int B, int H, int W, int cout_l);
int gdmae_conv3x3_dense_stat_rows(void);
int gdmae_conv3x3_dense_stats(const void* X, int B, int H, int W, int cin_l, int cout_l, int dil, const void* packed, const float* bias,
                              void* Y, float* stat_rows, void* workspace, void* stream);
/* ... with fp32 output rows, optionally ADDED to Y's previous content: six launches on the three-piece bf16 splits of both operands
 * (x = p0 + p1 + p2 to 2^-25; the pairs with i + j <= 2) give the convolution to fp32 accuracy - the fp32 parity mode's decoder conv_out
 * (spt_backbone_mae.py:46-50) and the fp32 fine-tune convolutions, gdmae_hip/dense.py Conv3x3DenseF32 */
int gdmae_conv3x3_dense_f32out(const void* X, int B, int H, int W, int cin_l, int cout_l, int dil, const void* packed, const float* bias,
                               float* Y, int accumulate, void* stream);
size_t gdmae_conv3x3_dense_dw_workspace_bytes(int B, int H, int W, int cin_l, int cout_l);
/* dW (cout, cin, 3, 3) fp32 ACCUMULATED (fixed summation order) from X (B, H, W, cin_l) and dY (B, H, W, cout_l), cin_l % 64 == 0,
 * cout_l % 32 == 0 */
int gdmae_conv3x3_dense_bwd_weight(const void* X, const void* dY, int B, int H, int W, int cin_l, int cout_l, int cin, int cout, int dil,
                                   float* dW, void* workspace, void* stream);

/* ---- a17-a19: reconstruction targets and Chamfer loss ----------------------------------------- *
 * gdmae_group_gt_points replaces sst_ops_cuda.group_inner_inds_wrapper (sst_ops_api.cpp:8;
 * sst_ops_gpu.cu:22-39) + points[group_inds] + get_voxel_centers (common_utils.py:130-145):
 *   gt_points (M,K,3) = xyz of the first min(cnt,K) points of the pillar (canonical order, cyclic pad)
 *   minus the pillar centre; gt_index (M,K) int32 optional.
 * gdmae_chamfer replaces pytorch3d.loss.chamfer_distance(pred, gt, weights) (spt_backbone_mae.py:88):
 *   term[m] = w_m ((1/P1) sum_i min_j |x_i-y_j|^2 + (1/P2) sum_j min_i |x_i-y_j|^2), dpred = d term / d pred.
 *   loss = sum(term) / sum(w). */
int gdmae_group_gt_points(const float* points, int n_cols, const int* pillar_pt_off, const int* pillar_pts,
                          const long long* voxel_coords, int M, int K, const float* lo, const float* vs,
                          float* gt_points, int* gt_index, void* stream);
int gdmae_chamfer(const float* pred, const float* gt, const float* weights, int M, int P1, int P2, float* term,
                  float* dpred, void* stream);

/* ---- f2 (next row): on-GPU input pipeline ------------------------------------------------------- *
 * World flip / rotation / scaling (data_augmentor.py:54-143, common_utils.py:99-121), xy range mask
 * (data_processor.py:77-88, common_utils.py:124-127) and collate with the batch index in column 0 (dataset.py:181-186)
 * of the B raw frames of a batch in one pass.  raw (n_raw, F) fp32, frames back to back; frame_off (B+1) int32 device;
 * frame_params (B, 8) fp32 device = [flip_x, flip_y, cos, sin, scale, 0, 0, 0] per frame (drawn on the host like the
 * reference draws them); xy_range host {xmin, ymin, xmax, ymax}.  out (n_raw, 1+F) receives the kept points in
 * input order; kept_off (B+1) int32 device: first output row of every non-empty frame (-1 for an empty one),
 * kept_off[B] = number of rows written.  shuffle_points is a row permutation applied afterwards (gdmae_gather_rows). */
size_t gdmae_augment_collate_workspace_bytes(long long n_raw);
int gdmae_augment_collate(const float* raw, long long n_raw, int F, const int* frame_off, int B, const float* frame_params,
                          const float* xy_range /* host */, float* out, int* kept_off, void* workspace, void* stream);

/* ---- the reference's own native op API for this path, for arbitrary group ids ------------------- *
 * Drop-in equivalents of pybind module pcdet.ops.sst_ops.sst_ops_cuda (pcdet/ops/sst_ops/src/sst_ops_api.cpp:6-9):
 *   int ingroup_inds_wrapper(at::Tensor group_inds, at::Tensor out_inds)          (sst_ops.cpp:21-33)
 *   int group_inner_inds_wrapper(at::Tensor inverse_inds, at::Tensor group_inds)  (sst_ops.cpp:35-48)
 * gdmae_ingroup_inds: out_inds[i] = number of EARLIER elements with the same group id (ids in [0,n_groups)).
 * gdmae_group_inner_inds: group_inds (M,K) int64 = first min(cnt,K) member indices of every group in
 *   ascending order, slot k >= cnt filled with slot k % cnt, empty groups -1.
 * Differences from the reference wrappers: canonical order instead of atomic arrival order; the caller
 * supplies n_groups (no max().item() sync) and the workspace (no cudaMalloc/cudaFree per call); launches
 * go to `stream` (not the legacy default stream); errors are returned (no exit(-1)). */
size_t gdmae_group_workspace_bytes(long long n, long long n_groups);
int gdmae_ingroup_inds(const long long* group_inds, long long n, long long n_groups, long long* out_inds,
                       void* workspace, size_t workspace_bytes, void* stream);
int gdmae_group_inner_inds(const long long* inverse_inds, long long n, long long M, int K, long long* group_inds,
                           void* workspace, size_t workspace_bytes, void* stream);

/* ---- the whole geometry plan as ONE call (round 3) ---------------------------------------------------- *
 * Voxelization (gdmae_voxelize + gdmae_pillar_major_rows), random masking, the token set / cell map of every stage, all
 * sparse-conv rulebooks incl. the transposed ones, the full-resolution sites under the tokens of strided stages, both window
 * partitions of every stage and the decoder's active tiles - the index side of SURVEY rows a1-a3, a5-a10, a16, a17 that the
 * reference spreads over dyn_vfe.py:62-81, common_utils.py:49-63, spconv's rulebook builder, sst_utils.py:6-104,
 * spt_backbone.py:32-104 and sst_ops_gpu.cu - enqueued by a single call (~32 launches, no host sync, no allocation).
 * Every buffer lives in ONE caller-provided arena; gdmae_geometry_plan_layout lists (name, offset, bytes) of each buffer and
 * the arena size for given CAPACITIES (it does not depend on the data, cache it per shape).  Names: the gdmae_voxelize outputs
 * ("points", "point_coords", "inverse", "inverse32", "voxel_coords", "pillar_cell", "pt_off", "pillar_pts", "point_rank",
 * "sample_off", "pillar_mean", "cell2pillar"), "points_pm" / "row_pillar", "mask", "tok_pillar", per stage i "s<i>.tok_cell",
 * "s<i>.map", "s<i>.nbr_subm", "s<i>.nbr_subm_t", "s<i>.nbr_down", "s<i>.nbr_down_t", "s<i>.up_sites", per shift k
 * "s<i>.w<k>.{tok_win,tok_level,tok_slot,tok_pos,csr_tok,win_start,win_len}", "dec.tile_slot" / "dec.tile_list", and
 * "counts" = int32 [N points kept, M pillars | tokens of stage 0.. | 8 per (stage, shift): windows per level (3), tokens per
 * level (3), windows, tokens | active tiles | visible pillars] - the only thing the host has to read back.
 * noise: (min(n_points, B*Y*X)) fp32 masking noise, one value per pillar in pillar order (null when !masked).
 * Outputs are bit-identical to the per-operator entry points above. */
#define GDMAE_PLAN_MAX_STAGES 4
typedef struct gdmae_plan_params {
  long long n_points;          /* rows of `points` (capacity of every per-point / per-pillar buffer) */
  int n_cols, batch_size;
  float lo[3], vs[3];
  int grid[3];                 /* X, Y, Z (Z = 1) */
  int n_stages;
  int stride[GDMAE_PLAN_MAX_STAGES];               /* conv_down stride of the stage: 1 (first stage only) or 2 (k3 s2 p1) */
  int win_x[GDMAE_PLAN_MAX_STAGES], win_y[GDMAE_PLAN_MAX_STAGES];
  int n_levels[GDMAE_PLAN_MAX_STAGES];
  int drop_lo[GDMAE_PLAN_MAX_STAGES][3], drop_hi[GDMAE_PLAN_MAX_STAGES][3], max_tokens[GDMAE_PLAN_MAX_STAGES][3];
  int masked;                  /* 0: every pillar is a token (fine-tune backbone) */
  double keep_frac;            /* 1 - mask ratio (python double) */
  int n_dec;                   /* decoder source stages (0: no tile set) */
  int dec_sources[GDMAE_PLAN_MAX_STAGES];
  int want_pm;                 /* pillar-major point rows */
  long long cap_points;        /* capacity the LAYOUT is sized for (0: n_points).  A caller whose batches differ in point count from
                                  step to step asks for the layout of a rounded-up capacity once and then passes the exact n_points
                                  (<= cap_points) with every gdmae_geometry_plan call on that layout */
} gdmae_plan_params;
typedef struct gdmae_plan_buffer {
  char name[40];
  long long offset, bytes;
} gdmae_plan_buffer;
int gdmae_geometry_plan_layout(const gdmae_plan_params* p, gdmae_plan_buffer* table, int max_entries, int* n_entries,
                               size_t* total_bytes);
int gdmae_geometry_plan(const gdmae_plan_params* p, const float* points, const float* noise, void* arena, size_t arena_bytes,
                        void* stream);

/* ---- f1 (next row): CenterHead target assignment --------------------------------------------------- *
 * Replaces the per-box Python / CPU loop of CenterHead.assign_targets (pcdet/models/dense_heads/center_head.py:106-221;
 * centernet_utils.py:9-70 gaussian_radius / gaussian2D / draw_gaussian_to_heatmap) for one head: boxes of the head's
 * classes are compacted per sample in input order; per box the clamped centre, the Gaussian radius (reference fp32
 * operation order), inds / mask / regression targets [dx, dy, z, log dims, cos, sin, extras]; the heat map is the
 * per-class maximum of the Gaussian patches (atomic max on non-negative floats: order independent).
 * gt_boxes (B, n_max, box_dim >= 8) fp32 device, class in the last column (0 = padding); class_map (n_class_total + 1)
 * device int: global class id -> 1-based id inside this head / 0.  Outputs are zero-filled by the call. */
size_t gdmae_center_head_targets_workspace_bytes(int B, int num_max_objs);
int gdmae_center_head_targets(const float* gt_boxes, int B, int n_max, int box_dim, const int* class_map, int n_class_total,
                              int n_cls, const float* pc_range /* host [x0, y0] */, const float* voxel_size /* host [vx, vy] */,
                              float feature_map_stride, int fw, int fh, int num_max_objs, double gaussian_overlap, int min_radius,
                              float* heatmap, float* ret_boxes, long long* inds, long long* mask, void* workspace, void* stream);
/* CornerNet focal loss of CenterHead on the clamped sigmoid of the heat-map logits (loss_utils.py:273-312, center_head.py:236-238) as one
 * pass per direction.  logits: element (b, c, y, x) at strides_byxc[0] b + [1] y + [2] x + [3] c (elements; bf16 or fp32: the column slice
 * of the channels-last map the head convolution wrote), gt (B, C, H, W) fp32.  fwd: prob (optional, (B, C, H, W) fp32) = the clamped
 * sigmoid, partials = gdmae_focal_loss_rows() x 3 floats of scratch, out4 = {loss, positive sum, negative sum, #pos}.  bwd: dlogits
 * (B, H, W, C) channels-last, fp32 or bf16 = grad_out[0] * d loss / d logits. */
int gdmae_focal_loss_rows(void);
int gdmae_focal_loss_fwd(const void* logits, int logits_bf16, const long long* strides_byxc, const float* gt, int B, int C, int H, int W,
                         float* prob, float* partials, float* out4, void* stream);
int gdmae_focal_loss_bwd(const void* logits, int logits_bf16, const long long* strides_byxc, const float* gt, int B, int C, int H, int W,
                         const float* out4, const float* grad_out, void* dlogits, int dlogits_bf16, void* stream);
/* The same + iou_boxes (B, num_max_objs, 7) fp32 = the ground-truth box of every assigned slot (zero elsewhere): the target of the
 * IoU-aware head (center_head.py:118,161; tools/cfgs/waymo_models/gd_mae_iou.yaml:228); null = not wanted. */
int gdmae_center_head_targets_iou(const float* gt_boxes, int B, int n_max, int box_dim, const int* class_map, int n_class_total,
                                  int n_cls, const float* pc_range, const float* voxel_size, float feature_map_stride, int fw, int fh,
                                  int num_max_objs, double gaussian_overlap, int min_radius, float* heatmap, float* ret_boxes,
                                  float* iou_boxes, long long* inds, long long* mask, void* workspace, void* stream);
/* Box decoding of a head's K best heat-map cells, one launch (replaces centernet_utils.py:163-260 decode_bbox_from_heatmap and
 * its _topk / _transpose_and_gather_feat chain; IoU normalisation of center_head.py:296-299).  cell (B, K) int64: flat index
 * over (class, y, x) of the head's heat map; score (B, K); maps (B, c, H, W) fp32, dim = LOG sizes, rot = [cos, sin], vel / iou
 * optional (null).  boxes (B, K, 7 | 9) [x, y, z, dx, dy, dz, heading (, vx, vy)], labels (B, K) class inside the head, ious
 * (B, K) = clamp((iou + 1) / 2, 0, 1) or 1, valid (B, K) = inside post_center_limit_range [and score > score_thresh]. */
int gdmae_center_head_decode(const long long* cell, const float* score, const float* center, const float* center_z,
                             const float* dim, const float* rot, const float* vel, const float* iou, int B, int K, int H, int W,
                             const float* pc_range /* host [x0, y0] */, const float* voxel_size /* host [vx, vy] */,
                             float feature_map_stride, const float* post_center_limit_range /* host [6] */, float score_thresh,
                             int use_score_thresh, float* boxes, int* labels, float* ious, unsigned char* valid, void* stream);

/* ---- f4 (next row): rotated BEV IoU and NMS for evaluation ------------------------------------------ *
 * Replace the iou3d_nms CUDA extension (pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:236-414; iou3d_nms.cpp): boxes
 * (n, 7) fp32 device [x, y, z, dx, dy, dz, heading].  The overlap is the reference's: intersection polygon from proper
 * edge crossings + corners inside the other box with a 1e-2 margin, ordered by angle, shoelace area (fp32).
 * gdmae_boxes_bev_pairs: out (n, m) = BEV overlap area (mode 0) or BEV IoU (mode 1).
 * gdmae_nms_bev: boxes sorted by descending score -> keep (n) int64 kept indices in order, n_keep device int; rotated 1 =
 * rotated IoU (nms_gpu), 0 = axis-aligned footprints (nms_normal_gpu).  The suppression masks are scanned on the device. */
int gdmae_boxes_bev_pairs(const float* boxes_a, int n, const float* boxes_b, int m, int mode, float* out, void* stream);
size_t gdmae_nms_workspace_bytes(int n);
int gdmae_nms_bev(const float* boxes, int n, float thresh, int rotated, long long* keep, int* n_keep, void* workspace,
                  void* stream);

/* ---- a21: fused optimizer step over one flat buffer ------------------------------------------- *
 * Replaces clip_grad_norm_ (tools/train_utils/train_utils.py:52) and OptimWrapper.step
 * (tools/train_utils/optimization/fastai_optim.py:135-152: p *= 1 - wd*lr, then torch Adam). */
int gdmae_grad_sq_norm(const float* grad, long long n, float* partials /* >= 1024 */, float* sq_norm_out, void* stream);
/* segments: HOST array of 2 * n_segments element offsets [begin, end) of the optimised ranges of the flat buffers (<= 16;
 * parameters outside them only take part in the clip norm); grad_scale multiplies gradient and norm first (1 / world size
 * after a SUM all-reduce, so that no separate averaging pass over the 32 MB buffer is needed). */
int gdmae_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const long long* segments,
                    int n_segments, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    float max_norm, float grad_scale, const float* sq_norm, void* stream);
/* the same step, additionally writing the bf16 copy of every UPDATED element to param_bf16 (same element offsets; NULL = none): the
 * shadow the bf16 GEMMs read comes out of the optimizer launch instead of a cast pass over the flat buffer */
int gdmae_adam_step_shadow(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const long long* segments,
                           int n_segments, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                           float max_norm, float grad_scale, const float* sq_norm, void* param_bf16, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDMAE_HIP_H */
