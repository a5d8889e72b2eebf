// On-GPU input pipeline for one batch (SURVEY §8 "next" row f2).
//
// Replaces, for the SSL pre-training configuration (tools/cfgs/waymo_models/gd_mae_ssl.yaml:18-41), the per-frame numpy
// chain of the reference's DataLoader workers:
//   random_world_flip / random_world_rotation / random_world_scaling      pcdet/datasets/augmentor/data_augmentor.py:54-143
//     (rotate_points_along_z = points[:, :3] @ [[c, s, 0], [-s, c, 0], [0, 0, 1]] in fp32, common_utils.py:99-121)
//   mask_points_and_boxes_outside_range (x, y inside the closed range)    processor/data_processor.py:77-88,
//                                                                         common_utils.py:124-127
//   collate_batch: batch index prepended as column 0, frames concatenated dataset.py:181-186
// on the raw frames as they arrive from pinned host memory: one pass, 4 * F bytes read and 4 * (1 + F) bytes written
// per kept point, plus a packed scan for the compaction.  The random decisions (flip flags, angle, scale) are drawn
// on the host exactly like the reference does (gdmae_hip/input_pipeline.py) and passed as a small per-frame table;
// shuffle_points (data_processor.py:90-100) is a permutation of the kept rows applied afterwards with the row
// gather kernel (device random keys + sort in production, an explicit permutation in the parity test).
// The order of operations and the fp32 arithmetic follow the reference: flip, then x' = x c - y s, y' = x s + y c
// (evaluated as the reference's matmul does: products then sum, no FMA contraction), then scale, then the range test
// on the transformed coordinates.
#include "common.h"

struct AugParams {   // per frame, fp32: [flip_x, flip_y, cos, sin, scale, pad, pad, pad]
  const float* tab;
  const int* frame_off;   // (B + 1) device
  int B;
};

struct AugLoad {
  const float* raw;
  int F;
  AugParams P;
  float xmin, ymin, xmax, ymax;
  __device__ int frame_of(long long i) const {
    int b = 0;
    while (b + 1 < P.B && i >= P.frame_off[b + 1]) ++b;
    return b;
  }
  __device__ void xform(long long i, int b, float& x, float& y, float& z) const {
    const float* p = raw + i * F;
    const float* t = P.tab + b * 8;
    x = p[0];
    y = p[1];
    z = p[2];
    if (t[0] != 0.f) y = -y;          // flip along x: y -> -y
    if (t[1] != 0.f) x = -x;          // flip along y: x -> -x
    const float c = t[2], s = t[3];
    const float xr = __fadd_rn(__fmul_rn(x, c), __fmul_rn(y, -s));
    const float yr = __fadd_rn(__fmul_rn(x, s), __fmul_rn(y, c));
    x = __fmul_rn(xr, t[4]);
    y = __fmul_rn(yr, t[4]);
    z = __fmul_rn(z, t[4]);
  }
  __device__ int operator()(long long i) const {
    float x, y, z;
    xform(i, frame_of(i), x, y, z);
    return (x >= xmin && x <= xmax && y >= ymin && y <= ymax) ? 1 : 0;
  }
};

struct AugStore {
  AugLoad L;
  float* out;        // (n_kept, 1 + F)
  int* kept_off;     // (B + 1): first output row of every frame
  __device__ void operator()(long long i, int ex, int v) const {
    const int b = L.frame_of(i);
    if (i == L.P.frame_off[b]) kept_off[b] = ex;
    if (v) {
      float x, y, z;
      L.xform(i, b, x, y, z);
      float* o = out + (long long)ex * (L.F + 1);
      o[0] = (float)b;
      o[1] = x;
      o[2] = y;
      o[3] = z;
      const float* p = L.raw + i * L.F;
      for (int c = 3; c < L.F; ++c) o[1 + c] = p[c];
    }
  }
};

extern "C" size_t gdmae_augment_collate_workspace_bytes(long long n_raw) { return (gd_scan_ws_elems(n_raw) + 4) * sizeof(int); }

// raw (n_raw, F) fp32 frames back to back; frame_off (B+1) int32 device; frame_params (B, 8) fp32 device;
// xy_range host {xmin, ymin, xmax, ymax}; out (>= n_raw, 1+F); kept_off (B+1) int32 device, kept_off[B] = rows written.
extern "C" int gdmae_augment_collate(const float* raw, long long n_raw, int F, const int* frame_off, int B,
                                     const float* frame_params, const float* xy_range, float* out, int* kept_off,
                                     void* workspace, void* stream) {
  GD_REQUIRE(F >= 3 && B >= 1, "augment_collate: need xyz columns and at least one frame");
  hipStream_t st = (hipStream_t)stream;
  if (n_raw <= 0) {
    GD_CHECK(hipMemsetAsync(kept_off, 0, sizeof(int) * (B + 1), st));
    return 0;
  }
  AugLoad L{raw, F, AugParams{frame_params, frame_off, B}, xy_range[0], xy_range[1], xy_range[2], xy_range[3]};
  // frames without points never hit `i == frame_off[b]`: pre-fill with -1, fixed up by the caller-visible rule below
  GD_CHECK(hipMemsetAsync(kept_off, 0xFF, sizeof(int) * (B + 1), st));
  return gd_device_scan<int>(n_raw, L, AugStore{L, out, kept_off}, kept_off + B, (int*)workspace, st);
}
