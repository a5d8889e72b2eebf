"""Optional HIP-event timing of individual kernel launches (used by bench.py's roofline leg only).

Events are recorded on torch's current stream, which is the stream every gdmae_hip launch goes to.
"""
from __future__ import annotations

import contextlib

import torch

_ACTIVE = None


class KernelTimers:
    def __init__(self):
        self.records = {}     # name -> list of (start_event, end_event, algorithmic_bytes)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [s.elapsed_time(e) for s, e, _ in recs]
            by = [b for _, _, b in recs]
            out[name] = {"launches": len(recs), "total_ms": sum(ms), "avg_us": 1e3 * sum(ms) / max(len(ms), 1),
                         "bytes_per_launch": sum(by) / max(len(by), 1), "total_bytes": sum(by)}
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
def kernel(name: str, algorithmic_bytes: float):
    if _ACTIVE is None:
        yield
        return
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    yield
    e.record()
    _ACTIVE.records.setdefault(name, []).append((s, e, float(algorithmic_bytes)))
