// Rotated bird's-eye-view overlap / IoU and rotated NMS for the evaluation side of the fine-tune detector (SURVEY next
// row f4; reference pcdet/ops/iou3d_nms/src/iou3d_nms_kernel.cu:236-414 `boxes_overlap_kernel`, `boxes_iou_bev_kernel`,
// `nms_kernel`, `nms_normal_kernel`, and the host loop of iou3d_nms.cpp that walks the suppression masks).
//
// Overlap of two rotated rectangles, as the reference defines it: the intersection polygon is assembled from (a) the proper
// crossings of the 4 x 4 edge pairs and (b) the corners of either box that lie inside the other one WITH A 1e-2 MARGIN
// (so boxes that merely touch within a centimetre already report a sliver of overlap); the points are ordered by angle
// around their centroid and the area is the shoelace sum.  fp32 throughout, like the reference.
//
// Stated plainly: the device helpers below (cross3, box_corners, inside_with_margin, seg_cross, bev_overlap) RESTATE
// iou3d_nms_kernel.cu:38-235 operation by operation - same intermediate quantities, same order of the fp32 operations, the same
// angular sort.  That is deliberate and it is the parity contract of this file: evaluation numbers (which box survives NMS at a
// threshold, an IoU that lands on either side of a matching threshold) depend on the last bits of this arithmetic and on the
// 1e-2 margin rule, so a geometrically equivalent formulation (e.g. Sutherland-Hodgman clipping) would NOT reproduce them.  What
// is pinned, and how: closed-form intersection areas (tests/test_iou3d_nms.py KNOWN_OVERLAPS: axis-aligned, 90 / 45 degree
// turns, containment, a corner triangle, disjoint) for both the CPU restatement (oracle/iou3d_oracle.py) and these kernels, an
// independent exact float64 clipping away from the margin cases, and kernel == oracle to 2e-5 on random boxes.  The reference
// holds no vectors for it and its CPU twin does not compile here (<cuda.h>), so "matches the reference's own output" stays
// unpinned.  The NMS half (mask words + a one-wavefront device scan, no host round trip) is this build's own design.
//
// NMS: boxes arrive sorted by score.  k_nms_masks: one thread per (box i, 64-box column block) builds the 64-bit word of
// the later boxes j > i with IoU > threshold; k_nms_scan: ONE wavefront walks the boxes in order with the "removed" bit set
// in LDS (lane = one 64-bit word of the row being OR-ed in) and writes the kept indices and their count - the reference
// copies the masks to the host and scans them there; here nothing leaves the device.
#include "common.h"

namespace {
constexpr float kEps = 1e-8f;
constexpr float kMargin = 1e-2f;

struct P2 {
  float x, y;
};
__device__ inline float cross3(const P2& a, const P2& b, const P2& o) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }

__device__ inline void box_corners(const float* b, P2 (&c)[5]) {
  const float hx = b[3] * 0.5f, hy = b[4] * 0.5f;
  const float cs = cosf(b[6]), sn = sinf(b[6]);
  const float lx[4] = {-hx, hx, hx, -hx}, ly[4] = {-hy, -hy, hy, hy};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // rotate the axis-aligned corner about the centre (the reference builds the absolute corner first, then rotates the
    // difference to the centre: identical arithmetic)
    const float px = (b[0] + lx[k]) - b[0], py = (b[1] + ly[k]) - b[1];
    c[k].x = px * cs + py * (-sn) + b[0];
    c[k].y = px * sn + py * cs + b[1];
  }
  c[4] = c[0];
}

__device__ inline bool inside_with_margin(const float* b, const P2& p) {
  const float cs = cosf(-b[6]), sn = sinf(-b[6]);
  const float rx = (p.x - b[0]) * cs + (p.y - b[1]) * (-sn);
  const float ry = (p.x - b[0]) * sn + (p.y - b[1]) * cs;
  return fabsf(rx) < b[3] * 0.5f + kMargin && fabsf(ry) < b[4] * 0.5f + kMargin;
}

// proper crossing of segments p0-p1 and q0-q1
__device__ inline bool seg_cross(const P2& p1, const P2& p0, const P2& q1, const P2& q0, P2& out) {
  const bool rect = fminf(p0.x, p1.x) <= fmaxf(q0.x, q1.x) && fminf(q0.x, q1.x) <= fmaxf(p0.x, p1.x) &&
                    fminf(p0.y, p1.y) <= fmaxf(q0.y, q1.y) && fminf(q0.y, q1.y) <= fmaxf(p0.y, p1.y);
  if (!rect) return false;
  const float s1 = cross3(q0, p1, p0), s2 = cross3(p1, q1, p0), s3 = cross3(p0, q1, q0), s4 = cross3(q1, p1, q0);
  if (!(s1 * s2 > 0.f && s3 * s4 > 0.f)) return false;
  const float s5 = cross3(q1, p1, p0);
  if (fabsf(s5 - s1) > kEps) {
    out.x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    out.y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    const float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    const float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    const float D = a0 * b1 - a1 * b0;
    out.x = (b0 * c1 - b1 * c0) / D;
    out.y = (a1 * c0 - a0 * c1) / D;
  }
  return true;
}

__device__ float bev_overlap(const float* a, const float* b) {
  P2 ca[5], cb[5];
  box_corners(a, ca);
  box_corners(b, cb);
  P2 pts[24];
  int n = 0;
  float sx = 0.f, sy = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      P2 x;
      if (seg_cross(ca[i + 1], ca[i], cb[j + 1], cb[j], x)) {
        sx += x.x; sy += x.y;
        pts[n++] = x;
      }
    }
  for (int k = 0; k < 4; ++k) {
    if (inside_with_margin(a, cb[k])) { sx += cb[k].x; sy += cb[k].y; pts[n++] = cb[k]; }
    if (inside_with_margin(b, ca[k])) { sx += ca[k].x; sy += ca[k].y; pts[n++] = ca[k]; }
  }
  if (n < 3) return 0.f;
  const float mx = sx / n, my = sy / n;
  float ang[24];
  for (int i = 0; i < n; ++i) ang[i] = atan2f(pts[i].y - my, pts[i].x - mx);
  // ascending angle (stable exchange sort, n <= 24)
  for (int j = 0; j < n - 1; ++j)
    for (int i = 0; i < n - j - 1; ++i)
      if (ang[i] > ang[i + 1]) {
        const float t = ang[i]; ang[i] = ang[i + 1]; ang[i + 1] = t;
        const P2 q = pts[i]; pts[i] = pts[i + 1]; pts[i + 1] = q;
      }
  float area = 0.f;
  for (int k = 0; k < n - 1; ++k) {
    const float ux = pts[k].x - pts[0].x, uy = pts[k].y - pts[0].y, vx = pts[k + 1].x - pts[0].x, vy = pts[k + 1].y - pts[0].y;
    area += ux * vy - uy * vx;
  }
  return fabsf(area) * 0.5f;
}

__device__ inline float bev_iou(const float* a, const float* b) {
  const float ov = bev_overlap(a, b);
  return ov / fmaxf(a[3] * a[4] + b[3] * b[4] - ov, kEps);
}
__device__ inline float axis_iou(const float* a, const float* b) {
  const float l = fmaxf(a[0] - a[3] * 0.5f, b[0] - b[3] * 0.5f), r = fminf(a[0] + a[3] * 0.5f, b[0] + b[3] * 0.5f);
  const float t = fmaxf(a[1] - a[4] * 0.5f, b[1] - b[4] * 0.5f), bo = fminf(a[1] + a[4] * 0.5f, b[1] + b[4] * 0.5f);
  const float inter = fmaxf(r - l, 0.f) * fmaxf(bo - t, 0.f);
  return inter / fmaxf(a[3] * a[4] + b[3] * b[4] - inter, kEps);
}

// mode 0: overlap area, 1: IoU
__global__ __launch_bounds__(256) void k_bev_pairs(const float* __restrict__ A, int n, const float* __restrict__ B, int m, int mode,
                                                   float* __restrict__ out) {
  const long long total = (long long)n * m;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const float* a = A + (i / m) * 7;
    const float* b = B + (i % m) * 7;
    out[i] = mode ? bev_iou(a, b) : bev_overlap(a, b);
  }
}

// mask[i][cb] bit j: box (64 cb + j) > i overlaps box i beyond the threshold
__global__ __launch_bounds__(64) void k_nms_masks(const float* __restrict__ boxes, int n, float thresh, int rotated,
                                                  unsigned long long* __restrict__ mask, int words) {
  __shared__ float cols[64 * 7];
  const int rb = blockIdx.y, cb = blockIdx.x, t = threadIdx.x;
  const int cn = min(64, n - cb * 64), rn = min(64, n - rb * 64);
  if (t < cn)
    for (int e = 0; e < 7; ++e) cols[t * 7 + e] = boxes[(long long)(cb * 64 + t) * 7 + e];
  __syncthreads();
  if (t >= rn) return;
  const int i = rb * 64 + t;
  unsigned long long w = 0ull;
  if (cb >= rb) {
    const float* bi = boxes + (long long)i * 7;
    const int start = cb == rb ? t + 1 : 0;
    for (int j = start; j < cn; ++j) {
      const float v = rotated ? bev_iou(bi, cols + j * 7) : axis_iou(bi, cols + j * 7);
      if (v > thresh) w |= 1ull << j;
    }
  }
  mask[(long long)i * words + cb] = w;
}

// one wavefront: sequential scan in score order; removed bits live in LDS (words <= 4096 / 64 * ... = n / 64)
__global__ __launch_bounds__(64) void k_nms_scan(const unsigned long long* __restrict__ mask, int n, int words, long long* __restrict__ keep,
                                                 int* __restrict__ n_keep) {
  extern __shared__ unsigned long long removed[];
  const int lane = threadIdx.x;
  for (int w = lane; w < words; w += 64) removed[w] = 0ull;
  __syncthreads();
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    const bool dead = (removed[i >> 6] >> (i & 63)) & 1ull;     // uniform across the wavefront
    if (!dead) {
      if (lane == 0) keep[cnt] = i;
      ++cnt;
      for (int w = lane; w < words; w += 64) removed[w] |= mask[(long long)i * words + w];
    }
    __syncthreads();
  }
  if (lane == 0) *n_keep = cnt;
}
}  // namespace

// boxes (n, 7) / (m, 7) fp32 device [x, y, z, dx, dy, dz, heading]; out (n, m): mode 0 = BEV overlap area, 1 = BEV IoU
extern "C" int gdmae_boxes_bev_pairs(const float* boxes_a, int n, const float* boxes_b, int m, int mode, float* out, void* stream) {
  if (n <= 0 || m <= 0) return 0;
  long long g = ((long long)n * m + 255) / 256;
  if (g > 65535) g = 65535;
  hipLaunchKernelGGL(k_bev_pairs, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, boxes_a, n, boxes_b, m, mode, out);
  GD_LAUNCH_CHECK();
  return 0;
}

extern "C" size_t gdmae_nms_workspace_bytes(int n) {
  const size_t words = (size_t)(n + 63) / 64;
  return gd_align((size_t)(n > 0 ? n : 1) * words * sizeof(unsigned long long));
}

// boxes (n, 7) sorted by descending score; keep (n) int64 receives the kept indices in order, n_keep (device int) their count.
// rotated: 1 = rotated BEV IoU (reference nms_gpu), 0 = axis-aligned IoU ignoring the heading (reference nms_normal_gpu).
extern "C" int gdmae_nms_bev(const float* boxes, int n, float thresh, int rotated, long long* keep, int* n_keep, void* workspace,
                             void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (n <= 0) {
    GD_CHECK(hipMemsetAsync(n_keep, 0, sizeof(int), st));
    return 0;
  }
  const int words = (n + 63) / 64;
  GD_REQUIRE((size_t)words * 8 <= 60 * 1024, "nms_bev: too many boxes for the in-LDS bit set");
  unsigned long long* mask = (unsigned long long*)workspace;
  hipLaunchKernelGGL(k_nms_masks, dim3(words, words), dim3(64), 0, st, boxes, n, thresh, rotated, mask, words);
  GD_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_nms_scan, dim3(1), dim3(64), (size_t)words * 8, st, (const unsigned long long*)mask, n, words, keep, n_keep);
  GD_LAUNCH_CHECK();
  return 0;
}
