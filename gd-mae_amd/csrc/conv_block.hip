// Native executor of one sparse-convolution block  SparseConv2d / SubMConv2d (k3) -> BatchNorm1d(train) -> ReLU
// (reference post_act_block, pcdet/utils/spconv_utils.py:37-56; used as conv_down / conv_out of every SSTBlockV1,
// spt_backbone.py:206,217,256-263): forward or backward enqueued by ONE C-ABI call instead of ~12 interpreter-issued ops
// per direction (the forward of the training step is host-bound).  Same arithmetic as
// gdmae_hip.ops.SparseConv3x3 + gdmae_hip.vfe.BNReLURows: rulebook gather (im2col) -> one GEMM -> column statistics ->
// folded affine + ReLU rows; backward = BatchNorm chain rule on column sums, split-K weight gradient, transposed
// rulebook gather + GEMM for the input gradient.  Parameter gradients are ACCUMULATED into the caller's fp32 buffers.
#include "../../include/gdmae_hip.h"
#include "common.h"
#include "dw_grouped.h"
#include "gemm.h"
#include <stdlib.h>

// spconv.hip: implicit-GEMM sparse convolution over the rulebook (bf16, 128 / 256 channels)
bool gd_spconv_supported(int cin, int cout);
int gd_spconv(hipStream_t st, const void* X, int x_f32, const int* nbr, const void* Wp, long long n, int cin, int cout, void* Y, int slot,
              float* part, const float* ride_part = nullptr, int ride_S = 0, int ride_cout = 0, int ride_cin = 0,
              float* ride_dW = nullptr);
int gd_spconv_rows(int cin, int cout, int x_f32);
int gd_bn_fold_from_partials(hipStream_t st, const float* part, int nblk, int C, double count, const float* gamma,
                             const float* beta, double eps, double momentum, float* running_mean, float* running_var,
                             long long* num_batches, double* stats, float* ab, float* mv);

namespace {

__device__ inline unsigned short cb_f2bf(float f) { return gd_to_bf16(f); }

// out[s, :] = bf16(src[idx[s], :]) (0 for idx < 0): im2col gather of fp32 token rows with the cast folded in
__global__ __launch_bounds__(256) void k_gather_rows_f32_bf16(const float* __restrict__ src, const int* __restrict__ idx,
                                                              long long n_slots, int C, unsigned short* __restrict__ out) {
  const int cv = C >> 3;
  const long long total = n_slots * cv;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long s = i / cv;
    const int c = (int)(i % cv) << 3;
    const int j = idx[s];
    uint4 q = make_uint4(0, 0, 0, 0);
    if (j >= 0) {
      const float4 a = *reinterpret_cast<const float4*>(src + (long long)j * C + c);
      const float4 b = *reinterpret_cast<const float4*>(src + (long long)j * C + c + 4);
      q.x = cb_f2bf(a.x) | ((unsigned)cb_f2bf(a.y) << 16);
      q.y = cb_f2bf(a.z) | ((unsigned)cb_f2bf(a.w) << 16);
      q.z = cb_f2bf(b.x) | ((unsigned)cb_f2bf(b.y) << 16);
      q.w = cb_f2bf(b.z) | ((unsigned)cb_f2bf(b.w) << 16);
    }
    *reinterpret_cast<uint4*>(out + s * C + c) = q;
  }
}

// Wt[(k, o), i] = W[o, k, i]  for W (cout, 9, cin): the B operand of the input-gradient GEMM
template <typename T>
__global__ __launch_bounds__(256) void k_permute_w(const T* __restrict__ W, int cout, int cin, T* __restrict__ Wt) {
  const int total = cout * 9 * cin;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int i = e % cin, ko = e / cin, o = ko % cout, k = ko / cout;
    Wt[e] = W[((long long)o * 9 + k) * cin + i];
  }
}

struct Scratch {
  char *gemm_ws, *cs_ws, *st, *rs_ws, *c01, *dy, *gcols, *wt, *sk_ws, *dw_part;
  size_t bytes;
};
// row slices of the gathered weight-gradient launch (~2+ workgroups per CU; GDMAE_SPCONV_DW_WGS overrides the bound) and the
// padded row count of dy that goes with them
int conv_dw_pick(long long n_out, int tiles, long long* n_pad) {
  static const int wgs = getenv("GDMAE_SPCONV_DW_WGS") ? atoi(getenv("GDMAE_SPCONV_DW_WGS")) : 640;
  return gd_dw_pick(n_out, tiles, wgs, n_pad);
}
// implicit: the im2col-free path (spconv.hip + the gathered grouped weight gradient): no gathered matrices, no library workspaces;
// dy is padded to the slice grid of the weight-gradient kernel, which also gets its partial tiles here
Scratch layout(void* base, long long n_in, long long n_out, int cin, int cout, int es, bool implicit) {
  Scratch s;
  size_t off = 0;
  auto take = [&](size_t b) { char* p = (char*)base + off; off += gd_align(b); return p; };
  s.gemm_ws = take(implicit ? 0 : GD_LT_WORKSPACE);
  s.cs_ws = take(gdmae_colstats_workspace_bytes(cout));
  s.st = take((size_t)3 * cout * sizeof(double));
  s.rs_ws = take(gdmae_rows_bwd_stats_workspace_bytes(cout));
  s.c01 = take((size_t)2 * cout * sizeof(float));
  long long n_pad = n_out;
  const int S_dw = implicit ? conv_dw_pick(n_out, 9 * (cout / 128) * (cin / 128), &n_pad) : 0;
  s.dy = take((size_t)n_pad * cout * es);
  s.gcols = take(implicit ? 0 : (size_t)n_in * 9 * cout * es);
  s.wt = take(implicit ? 0 : (size_t)9 * cout * cin * es);
  s.sk_ws = take(implicit ? 0 : gdmae_gemm_tn_splitk_workspace_bytes(n_out, cout, 9 * cin));
  s.dw_part = nullptr;
  if (implicit) {
    s.dw_part = take((size_t)S_dw * 9 * cout * cin * sizeof(float));
  }
  s.bytes = off;
  return s;
}
bool use_implicit(const gdmae_conv_block_args* a) {
  static const int off = getenv("GDMAE_SPCONV") ? atoi(getenv("GDMAE_SPCONV")) == 0 : 0;
  return !off && a->bf16 && a->packed_fwd != nullptr && a->packed_bwd != nullptr && gd_spconv_supported(a->cin, a->cout) &&
         gd_spconv_supported(a->cout, a->cin);
}

// dW[(o * 9 + k) * cin + i] += sum_s part[k][s][o][i]: the 9 per-tap partial products of the grouped weight-gradient launch into
// the (cout, 3, 3, cin) layout, fixed order
__global__ __launch_bounds__(256) void k_spconv_dw_reduce(const float* __restrict__ part, int S, int cout, int cin, float* __restrict__ dW) {
  const long long per_tap = (long long)S * cout * cin, mn4 = (long long)cout * cin / 4;
  for (long long e = blockIdx.x * (long long)blockDim.x + threadIdx.x; e < 9 * mn4; e += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(e / mn4);
    const long long r = e % mn4;
    const float4* p = reinterpret_cast<const float4*>(part + k * per_tap) + r;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int s = 0;
    for (; s + 8 <= S; s += 8) {                   // eight slices in flight, added in slice order (bit-identical to the serial loop)
      float4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = p[(long long)(s + j) * mn4];
#pragma unroll
      for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
    }
    for (; s < S; ++s) {
      const float4 v = p[(long long)s * mn4];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const long long o = (r * 4) / cin, i = (r * 4) % cin;
    float4* d = reinterpret_cast<float4*>(dW + (o * 9 + k) * cin + i);
    float4 w = *d;
    w.x += acc.x; w.y += acc.y; w.z += acc.z; w.w += acc.w;
    *d = w;
  }
}

#define CB_TRY(x)             \
  do {                        \
    int rc_ = (x);            \
    if (rc_ != 0) return rc_; \
  } while (0)

}  // namespace

extern "C" size_t gdmae_conv_block_scratch_bytes(long long n_in, long long n_out, int cin, int cout, int bf16) {
  size_t b = layout(nullptr, n_in, n_out, cin, cout, bf16 ? 2 : 4, false).bytes;
  if (bf16 && gd_spconv_supported(cin, cout) && gd_spconv_supported(cout, cin)) {
    const size_t bi = layout(nullptr, n_in, n_out, cin, cout, 2, true).bytes;
    b = bi > b ? bi : b;
  }
  return b;
}

extern "C" int gdmae_conv_block_fwd(const gdmae_conv_block_args* a, void* stream) {
  GD_REQUIRE(a->cin % 8 == 0 && a->cout % 8 == 0 && a->cout <= 256, "conv block: channels must be multiples of 8, cout <= 256");
  if (a->n_out <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int es = a->bf16 ? 2 : 4;
  const bool implicit = use_implicit(a);
  Scratch s = layout(a->scratch, a->n_in, a->n_out, a->cin, a->cout, es, implicit);
  const long long slots = a->n_out * 9;
  bool stats_fused = false;
  if (implicit) {
    // the gathered rows go straight into the MFMA operand tile: no im2col matrix, no library GEMM
    // ... and the BatchNorm statistics are its epilogue (per-workgroup column sums of the rounded rows) where the partial rows fit
    // the statistics workspace: no pass over y
    const int rpw = gd_spconv_rows(a->cin, a->cout, a->x_f32);
    const long long nblk = rpw > 0 ? (a->n_out + rpw - 1) / rpw : 0;
    static const bool fuse_ok = !(getenv("GDMAE_SPCONV_STATS") && atoi(getenv("GDMAE_SPCONV_STATS")) == 0);      // A/B switch
    stats_fused = fuse_ok && nblk >= 1 && nblk <= 1024;
    CB_TRY(gd_spconv(st, a->x, a->x_f32, a->nbr, a->packed_fwd, a->n_out, a->cin, a->cout, a->y, GD_T_SPCONV_FWD,
                     stats_fused ? (float*)s.cs_ws : nullptr));
    if (stats_fused)
      CB_TRY(gd_bn_fold_from_partials(st, (const float*)s.cs_ws, (int)nblk, a->cout, (double)a->n_out, a->gamma, a->beta, a->eps, a->momentum,
                                      a->running_mean, a->running_var, a->num_batches, a->stats, a->ab, a->mv));
  } else if (a->bf16 && a->x_f32) {
    long long g = (slots * (a->cin / 8) + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(k_gather_rows_f32_bf16, dim3((int)g), dim3(256), 0, st, (const float*)a->x, a->nbr, slots, a->cin,
                       (unsigned short*)a->cols);
    GD_LAUNCH_CHECK();
  } else {
    CB_TRY(gdmae_gather_rows(a->x, a->nbr, slots, a->cin * es, a->cols, stream));
  }
  if (!implicit)
    CB_TRY(gdmae_gemm(a->cols, a->W, a->y, a->n_out, a->cout, 9ll * a->cin, 0, 1, a->bf16, 0, nullptr, s.gemm_ws, stream));
  if (!stats_fused)
    CB_TRY(gdmae_bn_fold(a->y, a->n_out, a->cout, a->bf16, (double)a->n_out, a->gamma, a->beta, a->eps, a->momentum, a->running_mean,
                         a->running_var, a->num_batches, a->stats, a->ab, a->mv, s.cs_ws, stream));
  CB_TRY(gdmae_rows_affine_relu_scatter(a->y, a->bf16, nullptr, a->n_out, a->cout, a->ab, a->ab + a->cout, a->out,
                                        a->out_f32 ? 0 : a->bf16, a->cout, 0, stream));
  return 0;
}

extern "C" int gdmae_conv_block_bwd(const gdmae_conv_block_args* a, void* stream) {
  GD_REQUIRE(a->cin % 8 == 0 && a->cout % 8 == 0 && a->cout <= 256, "conv block: channels must be multiples of 8, cout <= 256");
  if (a->n_out <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const int es = a->bf16 ? 2 : 4;
  const bool implicit = use_implicit(a);
  Scratch s = layout(a->scratch, a->n_in, a->n_out, a->cin, a->cout, es, implicit);
  const int C = a->cout;
  // ---- BatchNorm1d + ReLU backward on the rows of y
  CB_TRY(gdmae_rows_bwd_stats(a->y, a->bf16, nullptr, a->n_out, C, a->ab, a->ab + C, a->g, a->g_f32 ? 0 : a->bf16, C, 0, nullptr,
                              s.rs_ws, stream));
  CB_TRY(gdmae_bn_bwd_coeffs_rows((const float*)s.rs_ws, gdmae_rows_bwd_stats_rows(a->n_out), 3, a->stats, a->ab, a->gamma, C,
                                  (double)a->n_out, nullptr, a->dgamma, a->dbeta, 1, (float*)s.c01, stream));
  CB_TRY(gdmae_rows_bwd(a->y, a->bf16, nullptr, a->n_out, C, a->ab, a->ab + C, (const float*)s.c01, (const float*)s.c01 + C, a->g,
                        a->g_f32 ? 0 : a->bf16, C, 0, s.dy, a->bf16, stream));
  if (implicit) {
    // ---- weight gradient: the nine per-tap products dW_k (cout, cin) = dy^T X[nbr[:, k]] as ONE grouped TN launch (the X rows
    //      are gathered through the rulebook column on load; all tiles of a row slice on one XCD), reduced into (cout, 9, cin)
    GdDwGroup Gp;
    Gp.n_jobs = 9;
    long long n_pad = 0;
    const int S = conv_dw_pick(a->n_out, 9 * (C / 128) * (a->cin / 128), &n_pad);
    for (int k = 0; k < 9; ++k) {
      Gp.job[k] = GdDwJob{s.dy, a->x, C, a->cin, (float*)s.dw_part + (size_t)k * S * C * a->cin, nullptr, 0, a->nbr + k, 9, a->x_f32};
    }
    {
      GdTimed timed(GD_T_SPCONV_BWD, st, (double)a->n_out * (2.0 * C + 9.0 * a->cin * (a->x_f32 ? 4 : 2) + 36.0) + 36.0 * S * C * a->cin,
                    2.0 * a->n_out * 9.0 * C * a->cin);
      CB_TRY(gd_dw_grouped_s(st, Gp, n_pad, a->n_out, S));
    }
    GD_REQUIRE(Gp.S == S, "conv block: slice count");
    // ---- input gradient: the same implicit GEMM over the transposed rulebook with the per-tap transposed weights; the reduce of the
    //      weight gradient's partial tiles rides along as its first workgroups (GDMAE_DW_REDUCE_RIDES=0: a launch of its own)
    static const bool rides = !(getenv("GDMAE_DW_REDUCE_RIDES") && atoi(getenv("GDMAE_DW_REDUCE_RIDES")) == 0);
    bool reduced = false;
    if (a->dx && rides) {
      const int rc = gd_spconv(st, s.dy, 0, a->nbr_t, a->packed_bwd, a->n_in, C, a->cin, a->dx, GD_T_SPCONV_BWD, nullptr, (const float*)s.dw_part, S,
                               C, a->cin, a->dW);
      if (rc != -2) {
        CB_TRY(rc);
        reduced = true;
      }
    }
    if (!reduced) {
      hipLaunchKernelGGL(k_spconv_dw_reduce, dim3(gd_div_up(9ll * C * a->cin / 4, 256)), dim3(256), 0, st, (const float*)s.dw_part, S, C, a->cin,
                         a->dW);
      GD_LAUNCH_CHECK();
      if (a->dx) CB_TRY(gd_spconv(st, s.dy, 0, a->nbr_t, a->packed_bwd, a->n_in, C, a->cin, a->dx, GD_T_SPCONV_BWD, nullptr));
    }
    return 0;
  }
  // ---- weight gradient: dW (cout, 9 cin) += dy^T cols
  CB_TRY(gdmae_gemm_tn_splitk(s.dy, a->cols, a->dW, a->n_out, C, 9 * a->cin, a->bf16, 1, s.sk_ws, stream));
  // ---- input gradient: dx (n_in, cin) = gather(dy, nbr_t) (n_in, 9 cout) @ Wt (9 cout, cin)
  if (a->dx) {
    CB_TRY(gdmae_gather_rows(s.dy, a->nbr_t, a->n_in * 9, C * es, s.gcols, stream));
    const int total = C * 9 * a->cin;
    if (a->bf16) hipLaunchKernelGGL((k_permute_w<unsigned short>), dim3((total + 255) / 256), dim3(256), 0, st, (const unsigned short*)a->W, C, a->cin, (unsigned short*)s.wt);
    else hipLaunchKernelGGL((k_permute_w<float>), dim3((total + 255) / 256), dim3(256), 0, st, (const float*)a->W, C, a->cin, (float*)s.wt);
    GD_LAUNCH_CHECK();
    CB_TRY(gdmae_gemm(s.gcols, s.wt, a->dx, a->n_in, a->cin, 9ll * C, 0, 0, a->bf16, 0, nullptr, s.gemm_ws, stream));
  }
  return 0;
}
