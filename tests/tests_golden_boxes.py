"""Synthetic ground-truth boxes for the fine-tune tests (same recipe as tests/golden/make_golden_head.py)."""
import numpy as np


def synth_boxes(rng, B, n_max, pcr, n_class):
    out = np.zeros((B, n_max, 8), dtype=np.float32)
    for b in range(B):
        n = int(rng.integers(n_max // 2, n_max - 1))
        out[b, :n, 0] = rng.uniform(pcr[0] + 1, pcr[3] - 1, n)
        out[b, :n, 1] = rng.uniform(pcr[1] + 1, pcr[4] - 1, n)
        out[b, :n, 2] = rng.uniform(-1.5, 0.5, n)
        cls = rng.integers(1, n_class + 1, n)
        size = {1: (4.2, 1.8, 1.6), 2: (0.8, 0.7, 1.7), 3: (1.8, 0.7, 1.6)}
        for i in range(n):
            out[b, i, 3:6] = np.array(size[min(int(cls[i]), 3)]) * rng.uniform(0.8, 1.3, 3)
        out[b, :n, 6] = rng.uniform(-np.pi, np.pi, n)
        out[b, :n, 7] = cls
    return out
