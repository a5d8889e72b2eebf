// CenterHead target assignment on the GPU (SURVEY next row f1; reference
// pcdet/models/dense_heads/center_head.py:106-221 `assign_target_of_single_head` / `assign_targets`, and
// pcdet/models/model_utils/centernet_utils.py:9-70 `gaussian_radius` / `gaussian2D` / `draw_gaussian_to_heatmap`).
//
// The reference moves the boxes to the CPU and loops over them in Python: per box a handful of scalar tensor ops, a numpy
// Gaussian patch and an in-place torch.max on a heat-map slice (B x ~50-500 iterations per step, plus the D2H / H2D
// copies).  Here: k_ch_prepare compacts the boxes of a head per sample in input order (ballot scan), computes the integer
// centre, the Gaussian radius (the reference's fp32 operation order, so the truncations agree) and the regression targets;
// k_ch_draw rasterises one box per workgroup with an atomic max on the (non-negative) float bit patterns - a maximum is
// order independent, so the heat map is deterministic.
#include "common.h"

struct ChBox {
  int cx, cy, r, cls;     // integer centre (feature-map cells), Gaussian radius, class index inside the head (0-based) / -1
};

// class_map: (n_class_total + 1) int: global class id (1-based, 0 = padding) -> 1-based id inside this head, 0 = not in this head
__global__ __launch_bounds__(64) void k_ch_prepare(const float* __restrict__ gt, int n_max, int box_dim, const int* __restrict__ class_map,
                                                   int n_class_total, float x0, float y0, float vsx, float vsy, float stride, int fw, int fh,
                                                   int num_max_objs, double overlap_d, int min_radius, ChBox* __restrict__ boxes,
                                                   float* __restrict__ ret_boxes, long long* __restrict__ inds, long long* __restrict__ mask,
                                                   int ret_dim, float* __restrict__ iou_boxes) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* g = gt + (long long)b * n_max * box_dim;
  int base = 0;
  for (int i0 = 0; i0 < n_max; i0 += 64) {
    const int i = i0 + lane;
    int hc = 0;
    if (i < n_max) {
      const int c = (int)g[(long long)i * box_dim + box_dim - 1];       // .long() truncation of the class column
      hc = (c >= 0 && c <= n_class_total) ? class_map[c] : 0;
    }
    const unsigned long long m = __ballot(hc > 0);
    const int k = base + __popcll(m & ((1ull << lane) - 1ull));
    base += __popcll(m);
    if (hc > 0 && k < num_max_objs) {
      const float* bx = g + (long long)i * box_dim;
      const float x = bx[0], y = bx[1], z = bx[2];
      float cxf = ((x - x0) / vsx) / stride, cyf = ((y - y0) / vsy) / stride;
      cxf = fminf(fmaxf(cxf, 0.f), (float)fw - 0.5f);
      cyf = fminf(fmaxf(cyf, 0.f), (float)fh - 0.5f);
      const int cxi = (int)cxf, cyi = (int)cyf;
      const float w = (bx[3] / vsx) / stride, h = (bx[4] / vsy) / stride;      // reference: dx, dy in cells; radius(height = dx, width = dy)
      ChBox rec;
      rec.cx = cxi; rec.cy = cyi; rec.cls = -1; rec.r = 0;
      if (w > 0.f && h > 0.f && cxi >= 0 && cxi <= fw && cyi >= 0 && cyi <= fh) {
        // gaussian_radius(height = w_, width = h_) in the reference's fp32 operation order
        // (python scalars such as 1 - min_overlap are doubles that enter the fp32 tensor arithmetic rounded to fp32)
        const float omo = (float)(1.0 - overlap_d), opo = (float)(1.0 + overlap_d), omn = (float)(overlap_d - 1.0);
        const float hh = w, ww = h;
        const float hw = hh + ww;
        const float c1 = ((ww * hh) * omo) / opo;
        const float r1 = (hw + sqrtf(hw * hw - 4.f * c1)) / 2.f;
        const float b2 = 2.f * hw;
        const float c2 = (omo * ww) * hh;
        const float r2 = (b2 + sqrtf(b2 * b2 - 16.f * c2)) / 2.f;
        const float a3x4 = (float)(4.0 * (4.0 * overlap_d));      // python: 4 * a3 with a3 = 4 * min_overlap, both doubles
        const float b3 = (float)(-2.0 * overlap_d) * hw;
        const float c3 = (omn * ww) * hh;
        const float r3 = (b3 + sqrtf(b3 * b3 - a3x4 * c3)) / 2.f;
        int r = (int)fminf(fminf(r1, r2), r3);
        r = r < min_radius ? min_radius : r;
        rec.r = r;
        rec.cls = hc - 1;
        inds[(long long)b * num_max_objs + k] = (long long)cyi * fw + cxi;
        mask[(long long)b * num_max_objs + k] = 1;
        float* rb = ret_boxes + ((long long)b * num_max_objs + k) * ret_dim;
        rb[0] = cxf - (float)cxi;
        rb[1] = cyf - (float)cyi;
        rb[2] = z;
        rb[3] = logf(bx[3]); rb[4] = logf(bx[4]); rb[5] = logf(bx[5]);
        rb[6] = cosf(bx[6]); rb[7] = sinf(bx[6]);
        for (int e = 8; e < ret_dim; ++e) rb[e] = bx[e - 1];               // extra regression targets (velocity ...)
        if (iou_boxes) {                                                   // IoU-aware head: the box itself (center_head.py:161)
          float* ib = iou_boxes + ((long long)b * num_max_objs + k) * 7;
          for (int e = 0; e < 7; ++e) ib[e] = bx[e];
        }
      }
      boxes[(long long)b * num_max_objs + k] = rec;
    }
  }
  // slots that received no box
  for (int k = base + lane; k < num_max_objs; k += 64) boxes[(long long)b * num_max_objs + k] = ChBox{0, 0, 0, -1};
}

__global__ __launch_bounds__(256) void k_ch_draw(const ChBox* __restrict__ boxes, int num_max_objs, int n_cls, int fw, int fh,
                                                 float* __restrict__ heatmap) {
  const int b = blockIdx.y, k = blockIdx.x;
  const ChBox rec = boxes[(long long)b * num_max_objs + k];
  if (rec.cls < 0) return;
  const int r = rec.r, x = rec.cx, y = rec.cy;
  const int left = x < r ? x : r, right = (fw - x) < (r + 1) ? (fw - x) : (r + 1);
  const int top = y < r ? y : r, bottom = (fh - y) < (r + 1) ? (fh - y) : (r + 1);
  const int wdt = left + right, hgt = top + bottom;
  if (wdt <= 0 || hgt <= 0) return;
  const double sigma = (double)(2 * r + 1) / 6.0;
  const double den = 2.0 * sigma * sigma;
  unsigned* hm = (unsigned*)(heatmap + ((long long)b * n_cls + rec.cls) * fh * fw);
  for (int i = threadIdx.x; i < wdt * hgt; i += blockDim.x) {
    const int dy = i / wdt - top, dx = i % wdt - left;
    double v = exp(-(double)(dx * dx + dy * dy) / den);
    if (v < 2.220446049250313e-16) v = 0.0;          // gaussian2D: h[h < eps * h.max()] = 0 (the patch maximum is 1)
    const float f = (float)v;
    atomicMax(hm + (long long)(y + dy) * fw + (x + dx), __float_as_uint(f));
  }
}

extern "C" size_t gdmae_center_head_targets_workspace_bytes(int B, int num_max_objs) {
  return gd_align((size_t)B * num_max_objs * sizeof(ChBox));
}

// gt_boxes (B, n_max, box_dim) fp32 device [x, y, z, dx, dy, dz, heading, (extras,) class]; class_map device int
// (n_class_total + 1): global class id -> 1-based id inside this head or 0.  Outputs (zero-filled here): heatmap
// (B, n_cls, fh, fw) fp32, ret_boxes (B, num_max_objs, box_dim) fp32, inds / mask (B, num_max_objs) int64.
extern "C" int gdmae_center_head_targets_iou(const float* gt_boxes, int B, int n_max, int box_dim, const int* class_map, int n_class_total,
                                             int n_cls, const float* pc_range, const float* voxel_size, float feature_map_stride, int fw, int fh,
                                             int num_max_objs, double gaussian_overlap, int min_radius, float* heatmap, float* ret_boxes,
                                             float* iou_boxes, long long* inds, long long* mask, void* workspace, void* stream);
extern "C" int gdmae_center_head_targets(const float* gt_boxes, int B, int n_max, int box_dim, const int* class_map, int n_class_total,
                                         int n_cls, const float* pc_range /* host [x0, y0] */, const float* voxel_size /* host [vx, vy] */,
                                         float feature_map_stride, int fw, int fh, int num_max_objs, double gaussian_overlap, int min_radius,
                                         float* heatmap, float* ret_boxes, long long* inds, long long* mask, void* workspace, void* stream) {
  return gdmae_center_head_targets_iou(gt_boxes, B, n_max, box_dim, class_map, n_class_total, n_cls, pc_range, voxel_size, feature_map_stride,
                                       fw, fh, num_max_objs, gaussian_overlap, min_radius, heatmap, ret_boxes, nullptr, inds, mask, workspace,
                                       stream);
}
// the same + iou_boxes (B, num_max_objs, 7) fp32: the ground-truth box of every assigned slot (zero elsewhere), the regression
// target of the IoU-aware head (center_head.py:118,161; null = not wanted)
extern "C" int gdmae_center_head_targets_iou(const float* gt_boxes, int B, int n_max, int box_dim, const int* class_map, int n_class_total,
                                             int n_cls, const float* pc_range, const float* voxel_size, float feature_map_stride, int fw, int fh,
                                             int num_max_objs, double gaussian_overlap, int min_radius, float* heatmap, float* ret_boxes,
                                             float* iou_boxes, long long* inds, long long* mask, void* workspace, void* stream) {
  GD_REQUIRE(B >= 1 && box_dim >= 8 && n_cls >= 1 && fw >= 1 && fh >= 1 && num_max_objs >= 1, "center_head_targets: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  GD_CHECK(hipMemsetAsync(heatmap, 0, (size_t)B * n_cls * fh * fw * sizeof(float), st));
  GD_CHECK(hipMemsetAsync(ret_boxes, 0, (size_t)B * num_max_objs * box_dim * sizeof(float), st));
  if (iou_boxes) GD_CHECK(hipMemsetAsync(iou_boxes, 0, (size_t)B * num_max_objs * 7 * sizeof(float), st));
  GD_CHECK(hipMemsetAsync(inds, 0, (size_t)B * num_max_objs * sizeof(long long), st));
  GD_CHECK(hipMemsetAsync(mask, 0, (size_t)B * num_max_objs * sizeof(long long), st));
  ChBox* boxes = (ChBox*)workspace;
  if (n_max > 0) {
    hipLaunchKernelGGL(k_ch_prepare, dim3(B), dim3(64), 0, st, gt_boxes, n_max, box_dim, class_map, n_class_total, pc_range[0], pc_range[1],
                       voxel_size[0], voxel_size[1], feature_map_stride, fw, fh, num_max_objs, gaussian_overlap, min_radius, boxes, ret_boxes,
                       inds, mask, box_dim, iou_boxes);
    GD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_ch_draw, dim3(num_max_objs < n_max ? num_max_objs : n_max, B), dim3(256), 0, st, (const ChBox*)boxes, num_max_objs,
                       n_cls, fw, fh, heatmap);
    GD_LAUNCH_CHECK();
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Box decoding of a head's K best heat-map cells as ONE launch (evaluation / RoI path; reference
// pcdet/models/model_utils/centernet_utils.py:163-260 `decode_bbox_from_heatmap` with its `_topk` / `_gather_feat` chain,
// and the IoU normalisation of center_head.py:296-299).  The reference gathers every regression map separately through
// (B, H*W, C) transposed copies of the maps; here one thread per candidate reads its 9-12 values straight from the
// (B, C, H, W) maps.  cell: flat index over (class, y, x) of the head's heat map (the K best of ALL classes: the set and the
// order the reference's per-class top-K followed by a top-K over classes x K produces).
// ------------------------------------------------------------------------------------------------------------------
struct ChDecode {
  const long long* cell;     // (B, K)
  const float* score;        // (B, K) sigmoid heat-map value
  const float *center, *center_z, *dim, *rot, *vel, *iou;   // (B, 2 | 1 | 3 | 2 | 2 | 1, H, W); vel / iou optional
  int B, K, H, W;
  float x0, y0, vsx, vsy, stride;
  float lim[6];
  float score_thresh;
  int use_thresh;
  int box_dim;               // 7, or 9 with vel
  float* boxes;              // (B, K, box_dim)
  int* labels;               // (B, K) class index inside the head
  float* ious;               // (B, K) clamp((iou + 1) / 2, 0, 1), or 1 without an iou map
  unsigned char* valid;      // (B, K) inside the post-centre range and above the score threshold
};
__global__ __launch_bounds__(256) void k_ch_decode(ChDecode D) {
  const long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (i >= (long long)D.B * D.K) return;
  const int b = (int)(i / D.K);
  const long long hw = (long long)D.H * D.W;
  const long long c = D.cell[i];
  const int cls = (int)(c / hw);
  const long long site = c % hw;
  const int y = (int)(site / D.W), x = (int)(site % D.W);
  auto at = [&](const float* m, int ch, int k) { return m[((long long)b * ch + k) * hw + site]; };
  float box[9];
  box[0] = (((float)x + at(D.center, 2, 0)) * D.stride) * D.vsx + D.x0;
  box[1] = (((float)y + at(D.center, 2, 1)) * D.stride) * D.vsy + D.y0;
  box[2] = at(D.center_z, 1, 0);
  box[3] = expf(at(D.dim, 3, 0)); box[4] = expf(at(D.dim, 3, 1)); box[5] = expf(at(D.dim, 3, 2));
  box[6] = atan2f(at(D.rot, 2, 1), at(D.rot, 2, 0));                 // rot = [cos, sin]
  if (D.vel) { box[7] = at(D.vel, 2, 0); box[8] = at(D.vel, 2, 1); }
  for (int e = 0; e < D.box_dim; ++e) D.boxes[i * D.box_dim + e] = box[e];
  D.labels[i] = cls;
  float q = 1.f;
  if (D.iou) q = fminf(fmaxf((at(D.iou, 1, 0) + 1.f) * 0.5f, 0.f), 1.f);
  D.ious[i] = q;
  bool ok = true;
  for (int e = 0; e < 3; ++e) ok = ok && box[e] >= D.lim[e] && box[e] <= D.lim[3 + e];
  if (D.use_thresh) ok = ok && D.score[i] > D.score_thresh;
  D.valid[i] = ok ? 1 : 0;
}
extern "C" int gdmae_center_head_decode(const long long* cell, const float* score, const float* center, const float* center_z,
                                        const float* dim, const float* rot, const float* vel, const float* iou, int B, int K, int H, int W,
                                        const float* pc_range /* host [x0, y0] */, const float* voxel_size /* host [vx, vy] */,
                                        float feature_map_stride, const float* post_center_limit_range /* host [6] */, float score_thresh,
                                        int use_score_thresh, float* boxes, int* labels, float* ious, unsigned char* valid, void* stream) {
  GD_REQUIRE(B >= 1 && K >= 1 && H >= 1 && W >= 1, "center_head_decode: bad sizes");
  ChDecode D;
  D.cell = cell; D.score = score; D.center = center; D.center_z = center_z; D.dim = dim; D.rot = rot; D.vel = vel; D.iou = iou;
  D.B = B; D.K = K; D.H = H; D.W = W;
  D.x0 = pc_range[0]; D.y0 = pc_range[1]; D.vsx = voxel_size[0]; D.vsy = voxel_size[1]; D.stride = feature_map_stride;
  for (int e = 0; e < 6; ++e) D.lim[e] = post_center_limit_range[e];
  D.score_thresh = score_thresh; D.use_thresh = use_score_thresh; D.box_dim = vel ? 9 : 7;
  D.boxes = boxes; D.labels = labels; D.ious = ious; D.valid = valid;
  hipLaunchKernelGGL(k_ch_decode, dim3(gd_div_up((long long)B * K, 256)), dim3(256), 0, (hipStream_t)stream, D);
  GD_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// CornerNet focal loss of CenterHead (reference pcdet/utils/loss_utils.py:273-312 on the clamped sigmoid of center_head.py:236-238):
//     p = clamp(sigmoid(x), 1e-4, 1 - 1e-4);  pos = [gt == 1], neg = [gt < 1]
//     loss = -(sum log(p) (1 - p)^2 pos + sum log(1 - p) p^2 (1 - gt)^4 neg) / max(#pos, 1)        (#pos = 0: only the negative sum)
// The reference issues ~20 elementwise passes over the (B, C, H, W) heat map per direction; here one pass per direction: the
// forward reads the logits where the head convolution left them (a column slice of a channels-last map: element strides per (b, y,
// x, c), bf16 or fp32), writes the clamped sigmoid the reference keeps in pred_dict['hm'] and per-workgroup partial sums; the
// backward writes d loss / d logits (scaled by the incoming scalar gradient) as a channels-last (B, H, W, C) tensor.
// ------------------------------------------------------------------------------------------------
namespace {
struct FlArgs {
  const void* x;
  int x_bf16;
  long long sb, sy, sx, sc;       // element strides of the logits
  const float* gt;                // (B, C, H, W) contiguous
  int B, C, H, W;
};
__device__ __forceinline__ float fl_logit(const FlArgs& A, int b, int c, int y, int x) {
  const long long o = b * A.sb + y * A.sy + x * A.sx + c * A.sc;
  return A.x_bf16 ? __uint_as_float(((unsigned)((const unsigned short*)A.x)[o]) << 16) : ((const float*)A.x)[o];
}
__global__ __launch_bounds__(256) void k_focal_fwd(FlArgs A, float* __restrict__ prob, float* __restrict__ part) {
  const long long total = (long long)A.B * A.C * A.H * A.W;
  float pl = 0.f, nl = 0.f, np = 0.f;
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int x = (int)(e % A.W);
    long long t = e / A.W;
    const int y = (int)(t % A.H);
    t /= A.H;
    const int c = (int)(t % A.C), b = (int)(t / A.C);
    const float v = fl_logit(A, b, c, y, x);
    const float s = 1.f / (1.f + expf(-v));
    const float p = fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
    const float g = A.gt[e];
    if (prob) prob[e] = p;
    if (g == 1.f) {
      pl += logf(p) * (1.f - p) * (1.f - p);
      np += 1.f;
    } else if (g < 1.f) {
      const float w = (1.f - g) * (1.f - g);
      nl += logf(1.f - p) * p * p * (w * w);
    }
  }
  __shared__ float sh[3][256];
  sh[0][threadIdx.x] = pl; sh[1][threadIdx.x] = nl; sh[2][threadIdx.x] = np;
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) {
#pragma unroll
      for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + d];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) part[(long long)blockIdx.x * 3 + threadIdx.x] = sh[threadIdx.x][0];
}
// out[4] = {loss, positive sum, negative sum, #pos}
__global__ __launch_bounds__(256) void k_focal_finish(const float* __restrict__ part, int nblk, float* __restrict__ out) {
  __shared__ double sh[3][256];
  double a[3] = {0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < nblk; i += 256)
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] += (double)part[(long long)i * 3 + k];
#pragma unroll
  for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] = a[k];
  __syncthreads();
  for (int d = 128; d > 0; d >>= 1) {
    if ((int)threadIdx.x < d) {
#pragma unroll
      for (int k = 0; k < 3; ++k) sh[k][threadIdx.x] += sh[k][threadIdx.x + d];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double pos = sh[0][0], neg = sh[1][0], n = sh[2][0];
    out[0] = (float)(n == 0.0 ? -neg : -(pos + neg) / (n < 1.0 ? 1.0 : n));
    out[1] = (float)pos;
    out[2] = (float)neg;
    out[3] = (float)n;
  }
}
__global__ __launch_bounds__(256) void k_focal_bwd(FlArgs A, const float* __restrict__ fin, const float* __restrict__ gout, void* __restrict__ dx,
                                                   int dx_bf16) {
  const long long total = (long long)A.B * A.C * A.H * A.W;
  const float n = fin[3];
  const float scale = -gout[0] / (n < 1.f ? 1.f : n);        // #pos = 0: -neg_loss, i.e. the same with a divisor of 1
  for (long long e = blockIdx.x * 256ll + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
    const int c = (int)(e % A.C);                            // output order: channels-last (b, y, x, c)
    long long t = e / A.C;
    const int x = (int)(t % A.W);
    t /= A.W;
    const int y = (int)(t % A.H), b = (int)(t / A.H);
    const float v = fl_logit(A, b, c, y, x);
    const float s = 1.f / (1.f + expf(-v));
    const bool inside = s >= 1e-4f && s <= 1.f - 1e-4f;      // clamp passes the gradient on its closed interval
    const float p = fminf(fmaxf(s, 1e-4f), 1.f - 1e-4f);
    const float g = A.gt[(((long long)b * A.C + c) * A.H + y) * A.W + x];
    float dp = 0.f;
    if (g == 1.f) {
      dp = (1.f - p) * (1.f - p) / p - 2.f * (1.f - p) * logf(p);
    } else if (g < 1.f) {
      const float w = (1.f - g) * (1.f - g);
      dp = (w * w) * (2.f * p * logf(1.f - p) - p * p / (1.f - p));
    }
    const float d = inside ? scale * dp * s * (1.f - s) : 0.f;
    if (dx_bf16) ((unsigned short*)dx)[e] = gd_to_bf16(d);
    else ((float*)dx)[e] = d;
  }
}
}  // namespace
extern "C" int gdmae_focal_loss_rows(void) { return 1024; }
extern "C" int gdmae_focal_loss_fwd(const void* logits, int logits_bf16, const long long* strides_byxc /* host [4]: elements */, const float* gt,
                                    int B, int C, int H, int W, float* prob, float* partials, float* out4, void* stream) {
  GD_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1 && logits && gt && partials && out4, "focal_loss_fwd: bad arguments");
  FlArgs A{logits, logits_bf16, strides_byxc[0], strides_byxc[1], strides_byxc[2], strides_byxc[3], gt, B, C, H, W};
  const long long total = (long long)B * C * H * W;
  const int nblk = (int)(total / 1024 + 1 > 1024 ? 1024 : total / 1024 + 1);
  hipLaunchKernelGGL(k_focal_fwd, dim3(nblk), dim3(256), 0, (hipStream_t)stream, A, prob, partials);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_focal_finish, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)partials, nblk, out4);
  GD_LAUNCH_CHECK();
  return 0;
}
extern "C" int gdmae_focal_loss_bwd(const void* logits, int logits_bf16, const long long* strides_byxc, const float* gt, int B, int C, int H, int W,
                                    const float* out4, const float* grad_out, void* dlogits /* (B, H, W, C) */, int dlogits_bf16, void* stream) {
  GD_REQUIRE(B >= 1 && C >= 1 && H >= 1 && W >= 1 && logits && gt && out4 && grad_out && dlogits, "focal_loss_bwd: bad arguments");
  FlArgs A{logits, logits_bf16, strides_byxc[0], strides_byxc[1], strides_byxc[2], strides_byxc[3], gt, B, C, H, W};
  const long long total = (long long)B * C * H * W;
  long long g = (total + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(k_focal_bwd, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, A, out4, grad_out, dlogits, dlogits_bf16);
  GD_LAUNCH_CHECK();
  return 0;
}
