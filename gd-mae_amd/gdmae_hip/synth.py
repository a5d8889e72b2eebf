"""Seeded synthetic LiDAR-like clouds (the exact recipe of SURVEY.md §8d 'Synthetic cloud generator').

64-beam spinning-sensor geometry with a ground plane and per-sector walls, plus near-range clutter,
then the dataset-side xy range filter (reference ``common_utils.mask_points_by_range``,
pcdet/utils/common_utils.py:124-127, inclusive bounds) and a seeded shuffle
(``data_processor.shuffle_points``).  Frame f of a run uses seed = base_seed + f.
"""
from __future__ import annotations

import numpy as np


def synth_frame(seed: int, point_cloud_range, beams=64, azimuths=2650, extra=10400, features=5,
                incl_deg=(-17.6, 2.4), sensor_h=2.0, max_range=97.5) -> np.ndarray:
    rng = np.random.default_rng(seed)
    incl = np.deg2rad(np.linspace(incl_deg[0], incl_deg[1], beams))
    az = np.linspace(-np.pi, np.pi, azimuths, endpoint=False)
    I, A = np.meshgrid(incl, az, indexing='ij')  # beam-major
    I, A = I.ravel(), A.ravel()
    with np.errstate(divide='ignore'):
        ground = np.where(I < -1e-3, sensor_h / np.tan(-I), np.inf)
    wall = np.exp(rng.uniform(np.log(6.0), np.log(80.0), 180))
    sector = np.minimum((A + np.pi) / (2 * np.pi) * 180, 179).astype(np.int64)
    wall_r = wall[sector] / np.cos(I)
    r = np.minimum(ground, wall_r) * (1 + rng.normal(0, 0.002, I.shape))
    ok = r < max_range
    r, I, A = r[ok], I[ok], A[ok]
    x = r * np.cos(I) * np.cos(A)
    y = r * np.cos(I) * np.sin(A)
    z = sensor_h + r * np.sin(I)
    ex = rng.uniform(-20, 20, (extra, 2))
    ez = rng.uniform(0, 2, extra) * (rng.uniform(0, 1, extra) < 0.3)
    xyz = np.concatenate([np.stack([x, y, z], 1), np.concatenate([ex, ez[:, None]], 1)], 0)
    n = xyz.shape[0]
    cols = [xyz, np.tanh(rng.gamma(2.0, 0.15, (n, 1)))]
    if features >= 5:
        cols.append(rng.uniform(0, 0.2, (n, 1)))
    pts = np.concatenate(cols, 1)[:, :features].astype(np.float32)
    lo, hi = point_cloud_range[:3], point_cloud_range[3:6]
    m = (pts[:, 0] >= lo[0]) & (pts[:, 0] <= hi[0]) & (pts[:, 1] >= lo[1]) & (pts[:, 1] <= hi[1])
    pts = pts[m]
    return pts[rng.permutation(pts.shape[0])]


def collate_points(frames) -> np.ndarray:
    """Batch assembly of ``DatasetTemplate.collate_batch`` for the 'points' key (reference
    pcdet/datasets/dataset.py:181-186): prepend the frame index as column 0 (stored as float) and
    concatenate frames in order."""
    out = []
    for i, p in enumerate(frames):
        out.append(np.pad(p, ((0, 0), (1, 0)), mode='constant', constant_values=i))
    return np.concatenate(out, axis=0).astype(np.float32)


def synth_batch(base_seed: int, batch_size: int, point_cloud_range, **kw) -> np.ndarray:
    return collate_points([synth_frame(base_seed + f, point_cloud_range, **kw) for f in range(batch_size)])
