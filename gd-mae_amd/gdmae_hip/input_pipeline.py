"""On-GPU input pipeline (SURVEY §8 "next" row f2): the reference's per-frame numpy chain of the SSL configuration

    random_world_flip -> random_world_rotation -> random_world_scaling      (data_augmentor.py:54-143)
    mask_points_and_boxes_outside_range -> shuffle_points                   (data_processor.py:77-100)
    collate_batch (batch index prepended, frames concatenated)              (dataset.py:181-186)

as ONE HIP pass over the raw frames (gdmae_augment_collate) + one row permutation.  The random decisions are drawn
on the host with the same ``np.random`` calls, in the same order, as the reference (so a seeded run sees the same
flips / angles / scales); the shuffle is a device-side random permutation within each frame (``torch.rand`` keys),
or an explicit permutation for parity tests.  No fallback: without libgdmae_hip.so this raises.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

from . import lib as L

# tools/cfgs/waymo_models/gd_mae_ssl.yaml:18-31
SSL_AUG_CONFIG = (
    {"NAME": "random_world_flip", "PROBABILITY": 0.5, "ALONG_AXIS_LIST": ["x", "y"]},
    {"NAME": "random_world_rotation", "PROBABILITY": 1.0, "WORLD_ROT_ANGLE": [-0.78539816, 0.78539816]},
    {"NAME": "random_world_scaling", "PROBABILITY": 1.0, "WORLD_SCALE_RANGE": [0.95, 1.05]},
)


def draw_world_params(aug_config: Sequence[dict] = SSL_AUG_CONFIG) -> dict:
    """Random world transformation of ONE frame; consumes ``np.random`` exactly like DataAugmentor.forward does
    (data_augmentor.py:62, :97-101, :123-127) so that seeded runs agree with the reference."""
    out = {"flip_x": False, "flip_y": False, "angle": 0.0, "scale": 1.0}
    for cfg in aug_config:
        p = cfg["PROBABILITY"]
        if cfg["NAME"] == "random_world_flip":
            for axis in cfg["ALONG_AXIS_LIST"]:
                if np.random.choice([False, True], replace=False, p=[1 - p, p]):
                    if axis not in ("x", "y"):
                        raise NotImplementedError(axis)
                    out["flip_" + axis] = True
        elif cfg["NAME"] == "random_world_rotation":
            enable = np.random.choice([False, True], replace=False, p=[1 - p, p])
            lo, hi = cfg["WORLD_ROT_ANGLE"] if enable else (0.0, 0.0)
            out["angle"] = float(np.random.uniform(lo, hi))
        elif cfg["NAME"] == "random_world_scaling":
            enable = np.random.choice([False, True], replace=False, p=[1 - p, p])
            lo, hi = cfg["WORLD_SCALE_RANGE"] if enable else (1.0, 1.0)
            out["scale"] = float(np.random.uniform(lo, hi))
        else:
            raise NotImplementedError(f"augmentation {cfg['NAME']} is outside the SSL pre-training configuration")
    return out


def params_table(params: Sequence[dict]) -> np.ndarray:
    """(B, 8) fp32 rows [flip_x, flip_y, cos, sin, scale, 0, 0, 0]: cos / sin evaluated in fp64 and rounded to fp32
    (rotate_points_along_z: torch.cos on the float64 angle, ``.float()``), scale rounded to fp32 (in-place multiply of
    a float32 array by a python float)."""
    t = np.zeros((len(params), 8), np.float32)
    for i, p in enumerate(params):
        a = np.float64(p["angle"])
        t[i, :5] = (float(p["flip_x"]), float(p["flip_y"]), np.float32(np.cos(a)), np.float32(np.sin(a)), np.float32(p["scale"]))
    return t


class GpuInputPipeline:
    def __init__(self, point_cloud_range, aug_config: Optional[Sequence[dict]] = SSL_AUG_CONFIG, shuffle: bool = True,
                 device: Optional[torch.device] = None):
        r = [float(v) for v in point_cloud_range]
        self.xy_range = (r[0], r[1], r[3], r[4])
        self.aug_config = aug_config
        self.shuffle = shuffle
        self.device = device or torch.device("cuda", torch.cuda.current_device())

    def _staging(self, n_elems: int) -> torch.Tensor:
        """Pinned host staging buffer, allocated once and grown geometrically (page-locking is slow)."""
        buf = getattr(self, "_pinned", None)
        if buf is None or buf.numel() < n_elems:
            buf = self._pinned = torch.empty(int(n_elems * 1.25) + 1024, dtype=torch.float32).pin_memory()
        ev = getattr(self, "_copied", None)
        if ev is not None:
            ev.synchronize()                        # previous batch's H2D copy has left the buffer
        return buf[:n_elems]

    def __call__(self, frames: List[np.ndarray], params: Optional[Sequence[dict]] = None,
                 perms: Optional[Sequence[np.ndarray]] = None) -> torch.Tensor:
        """frames: B raw clouds (n_i, F) float32 (x, y, z, features...).  Returns the collated batch (N, 1 + F) on the
        device.  ``params`` / ``perms``: explicit world transformations and per-frame permutations of the KEPT points
        (parity tests); by default they are drawn (np.random for the world parameters, torch.rand keys for the order)."""
        B = len(frames)
        F = int(frames[0].shape[1])
        if params is None:
            ident = {"flip_x": False, "flip_y": False, "angle": 0.0, "scale": 1.0}
            params = [draw_world_params(self.aug_config) if self.aug_config else dict(ident) for _ in range(B)]
        off = np.zeros(B + 1, np.int32)
        off[1:] = np.cumsum([f.shape[0] for f in frames])
        n_raw = int(off[-1])
        host = self._staging(max(n_raw, 1) * F).view(-1, F)[:max(n_raw, 1)]
        hv = host.numpy()
        for i, f in enumerate(frames):
            assert f.dtype == np.float32 and f.shape[1] == F
            hv[off[i]:off[i + 1]] = f
        dev = self.device
        raw = host.to(dev, non_blocking=True)
        self._copied = torch.cuda.Event()
        self._copied.record()                       # the staging buffer is reused by the next call
        off_d = torch.from_numpy(off).to(dev, non_blocking=True)
        tab_d = torch.from_numpy(params_table(params)).to(dev, non_blocking=True)
        out = torch.empty(max(n_raw, 1), 1 + F, dtype=torch.float32, device=dev)
        kept = torch.empty(B + 1, dtype=torch.int32, device=dev)
        ws = torch.empty(L.load().gdmae_augment_collate_workspace_bytes(n_raw), dtype=torch.uint8, device=dev)
        L.call("gdmae_augment_collate", L.ptr(raw), n_raw, F, L.ptr(off_d), B, L.ptr(tab_d), L.host_f32(self.xy_range), L.ptr(out),
               L.ptr(kept), L.ptr(ws), L.stream())
        kept_h = kept.tolist()                      # the one host sync of the pipeline (it runs a batch ahead)
        n = kept_h[B]
        for b in range(B - 1, -1, -1):              # empty frames: first row = that of the next frame
            if kept_h[b] < 0:
                kept_h[b] = kept_h[b + 1]
        pts = out[:n]
        if perms is not None:
            idx = torch.cat([torch.from_numpy(np.asarray(p, np.int64)) + kept_h[b] for b, p in enumerate(perms)]).to(dev)
        elif self.shuffle and n > 0:
            # random order within each frame: sort by (frame index + U[0,1))
            idx = torch.argsort(pts[:, 0] + torch.rand(n, device=dev))      # fp32 keys: 2^-20 resolution at B <= 8
        else:
            return pts
        return pts.index_select(0, idx)             # rows are 4 * (1 + F) = 20-24 bytes: below the 16-byte row kernels
