"""f2 measurement: GPU input pipeline (gdmae_augment_collate + shuffle) on a config-B batch (8 x 180 k points x 5
features) vs the numpy restatement of the reference chain on the host.  Prints one JSON line."""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gdmae_hip import input_pipeline as ip  # noqa: E402
from gdmae_hip import lib as L  # noqa: E402
from oracle import input_oracle as io  # noqa: E402

B, N, F = 8, 180000, 5
rng = np.array([-74.88, -74.88, -2, 74.88, 74.88, 4.0], np.float32)
g = np.random.default_rng(0)
frames = [np.concatenate([g.uniform(-80, 80, (N, 2)), g.uniform(-2, 4, (N, 1)), g.uniform(0, 1, (N, 2))], 1).astype(np.float32)
          for _ in range(B)]
np.random.seed(0)
params = [ip.draw_world_params() for _ in range(B)]
pipe = ip.GpuInputPipeline(rng)
for _ in range(3):
    out = pipe(frames, params=params)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    out = pipe(frames, params=params)
torch.cuda.synchronize()
e2e_ms = (time.perf_counter() - t0) / reps * 1e3
# kernel only, raw frames resident
dev = out.device
raw = torch.from_numpy(np.concatenate(frames)).to(dev)
off = torch.arange(0, (B + 1) * N, N, dtype=torch.int32, device=dev)
tab = torch.from_numpy(ip.params_table(params)).to(dev)
o = torch.empty(B * N, 1 + F, device=dev)
kept = torch.empty(B + 1, dtype=torch.int32, device=dev)
ws = torch.empty(L.load().gdmae_augment_collate_workspace_bytes(B * N), dtype=torch.uint8, device=dev)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(13):
    if i == 3:
        s.record()
    L.call("gdmae_augment_collate", L.ptr(raw), B * N, F, L.ptr(off), B, L.ptr(tab), L.host_f32(pipe.xy_range), L.ptr(o), L.ptr(kept),
           L.ptr(ws), L.stream())
e.record()
torch.cuda.synchronize()
k_us = s.elapsed_time(e) * 1e3 / 10
n_kept = int(kept[B])
algo_bytes = B * N * F * 4 + n_kept * (1 + F) * 4
t0 = time.perf_counter()
ref, _ = io.pipeline(frames, params, rng, [np.random.permutation(k) for k in io.pipeline(frames, params, rng)[1]])
cpu_ms = (time.perf_counter() - t0) * 1e3 / 2        # pipeline() evaluated twice above
print(json.dumps({"workload": f"{B} frames x {N} points x {F} features", "kept_points": n_kept,
                  "gpu_kernel_us": round(k_us, 1), "gpu_kernel_GBs": round(algo_bytes / k_us / 1e3, 1),
                  "gpu_end_to_end_ms_incl_pinned_copy_h2d_shuffle": round(e2e_ms, 2),
                  "frames_per_s_end_to_end": round(B / e2e_ms * 1e3, 1), "cpu_numpy_ms_per_batch_1_core": round(cpu_ms, 1),
                  "frames_per_s_cpu_1_core": round(B / cpu_ms * 1e3, 1)}))
