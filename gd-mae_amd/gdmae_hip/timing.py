"""Optional HIP-event timing of individual kernel launches (used by bench.py's roofline leg only).

Events are recorded on torch's current stream, which is the stream every gdmae_hip launch goes to.
"""
from __future__ import annotations

import contextlib

import torch

_ACTIVE = None


class KernelTimers:
    def __init__(self):
        self.records = {}     # name -> list of (start_event, end_event, algorithmic_bytes, algorithmic_flops, extra)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [r[0].elapsed_time(r[1]) for r in recs]
            by = [r[2] for r in recs]
            fl = [r[3] for r in recs]
            out[name] = {"launches": len(recs), "total_ms": sum(ms), "avg_us": 1e3 * sum(ms) / max(len(ms), 1),
                         "bytes_per_launch": sum(by) / max(len(by), 1), "total_bytes": sum(by),
                         "flops_per_launch": sum(fl) / max(len(fl), 1), "total_flops": sum(fl), "extra": recs[-1][4]}
        return out


def enabled() -> bool:
    return _ACTIVE is not None


@contextlib.contextmanager
def collect():
    global _ACTIVE
    prev, _ACTIVE = _ACTIVE, KernelTimers()
    try:
        yield _ACTIVE
    finally:
        _ACTIVE = prev


@contextlib.contextmanager
def kernel(name: str, algorithmic_bytes: float, flops: float = 0.0, extra=None):
    if _ACTIVE is None:
        yield
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    yield
    e.record()
    _ACTIVE.records.setdefault(name, []).append((s, e, float(algorithmic_bytes), float(flops), extra or {}))
