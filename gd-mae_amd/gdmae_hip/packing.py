"""MFMA-fragment-ordered weight images for the fused token GEMMs of the encoder layers (csrc/tok_gemm.hip).

A layer's ten GEMM operands (five forward, five transposed for the input-gradient products) are packed from the fp32
master weights by ``gdmae_tok_gemm_pack``.  Layers whose parameters live in a flat optimizer buffer are REGISTERED: their
images are refreshed by ONE launch per optimizer step (``repack_registered``, called next to the bf16 shadow refresh);
everything else is packed on the fly at every use.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import lib as L

_REG = {}          # id(Win) -> dict(ref, refs, vers, packed, jobs)
_TABLE = {}        # device index -> (signature, device job table, n_jobs)


def _vers(ts):
    return tuple(t._version for t in ts)


def _current(ent, ts):
    """A registered image is valid only for the weight VALUES it was packed from: the optimizer's kernels update the flat buffer
    behind torch's back (no version bump) and refresh every image right after (repack_registered), whereas any in-place change
    through torch between two optimizer steps (load_state_dict after a first forward, a mid-training resume, an EMA swap,
    p.data.copy_) bumps ``_version`` and keeps ``data_ptr`` - the image is then re-packed on the spot, INTO THE SAME BUFFER
    (native argument structs cache its address)."""
    v = _vers(ts)
    if ent["vers"] != v:
        jobs = torch.tensor(ent["jobs"], dtype=torch.int64).to(ent["packed"].device)
        L.call("gdmae_tok_gemm_pack", L.ptr(jobs), len(ent["jobs"]) // 6, L.stream())
        ent["packed"]._gd_jobs = jobs
        ent["vers"] = v
    return ent


def _jobs_for(Win, Wo, W1, W2, packed):
    d, ff = Wo.shape[0], W1.shape[0]
    jobs = (C.c_longlong * (6 * L.load().gdmae_layer_pack_job_count(d, ff)))()
    L.call("gdmae_layer_pack_jobs", L.ptr(Win), L.ptr(Wo), L.ptr(W1), L.ptr(W2), d, ff, L.ptr(packed), jobs)
    return list(jobs)


def _ok(Win, Wo, W1, W2):
    return all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() for t in (Win, Wo, W1, W2))


def pack_now(Win, Wo, W1, W2):
    """Fresh packed image of a layer (one launch); None when the weights are not fp32 device tensors."""
    if not _ok(Win, Wo, W1, W2):
        return None
    d, ff = Wo.shape[0], W1.shape[0]
    packed = torch.empty(L.load().gdmae_layer_packed_bytes(d, ff), dtype=torch.uint8, device=Win.device)
    jobs = torch.tensor(_jobs_for(Win.detach(), Wo.detach(), W1.detach(), W2.detach(), packed), dtype=torch.int64).to(Win.device)
    L.call("gdmae_tok_gemm_pack", L.ptr(jobs), jobs.numel() // 6, L.stream())
    packed._gd_jobs = jobs           # keep the table alive until the launch has run
    return packed


def registered(Win, Wo, W1, W2):
    """Packed image of a layer whose weights are refreshed by the optimizer: packed now, then once per step."""
    if not _ok(Win, Wo, W1, W2):
        return None
    key = id(Win)
    ent = _REG.get(key)
    ws = (Win, Wo, W1, W2)
    if ent is not None and ent["ref"]() is Win and ent["ptrs"] == tuple(t.data_ptr() for t in ws):
        return _current(ent, ws)["packed"]
    packed = pack_now(Win, Wo, W1, W2)
    _REG[key] = dict(ref=weakref.ref(Win), refs=[weakref.ref(t) for t in ws], vers=_vers(ws), packed=packed,
                     ptrs=tuple(t.data_ptr() for t in ws),
                     jobs=_jobs_for(Win.detach(), Wo.detach(), W1.detach(), W2.detach(), packed))
    _TABLE.clear()
    return packed


# ---- sparse-convolution weights (csrc/spconv.hip): per-tap images of W (cout, 3, 3, cin) and of its per-tap transposes
def _conv_jobs(W, fwd, bwd):
    cout, _, _, cin = W.shape
    jobs = (C.c_longlong * 54)()
    L.call("gdmae_spconv_pack_jobs", L.ptr(W), cin, cout, 0, L.ptr(fwd), jobs)
    out = list(jobs)
    L.call("gdmae_spconv_pack_jobs", L.ptr(W), cin, cout, 1, L.ptr(bwd), jobs)
    return out + list(jobs)


def conv_supported(W):
    return (W.is_cuda and W.dtype == torch.float32 and W.is_contiguous() and W.dim() == 4 and W.shape[1] == 3 and W.shape[2] == 3
            and W.shape[0] in (128, 256) and W.shape[3] in (128, 256))


def conv_pack_now(W):
    """(forward images, input-gradient images) of a sparse-conv weight, packed now (one launch)."""
    cout, _, _, cin = W.shape
    nb = L.load().gdmae_spconv_packed_bytes(cin, cout)
    fwd = torch.empty(nb, dtype=torch.uint8, device=W.device)
    bwd = torch.empty(nb, dtype=torch.uint8, device=W.device)
    jobs = torch.tensor(_conv_jobs(W.detach(), fwd, bwd), dtype=torch.int64).to(W.device)
    L.call("gdmae_tok_gemm_pack", L.ptr(jobs), 18, L.stream())
    fwd._gd_jobs = jobs
    return fwd, bwd


def conv_registered(W):
    """The same for a weight owned by a flat optimizer: packed now, then refreshed with the encoder layers' images by the ONE
    pack launch per optimizer step."""
    key = id(W)
    ent = _REG.get(key)
    if ent is not None and ent["ref"]() is W and ent["ptrs"] == (W.data_ptr(),):
        _current(ent, (W,))
        return ent["packed"], ent["packed2"]
    fwd, bwd = conv_pack_now(W)
    _REG[key] = dict(ref=weakref.ref(W), refs=[weakref.ref(W)], vers=_vers((W,)), packed=fwd, packed2=bwd, ptrs=(W.data_ptr(),),
                     jobs=_conv_jobs(W.detach(), fwd, bwd))
    _TABLE.clear()
    return fwd, bwd


# ---- decoder conv_out (Conv2d 3x3, weight (C2, Cin, 3, 3)): per (source stage, tap) the (w, C2) image A[ci][co] = W[co][col + ci][tap]
#      of the input-gradient convolution over the tile-compact output gradient (gdmae_hip/decoder.py backward)
def _dec_jobs(W, widths, packed):
    C2, Cin = W.shape[0], W.shape[1]
    jobs, col, off = [], 0, 0
    for w in widths:
        for k in range(9):
            jobs += [W.data_ptr() + 4 * (col * 9 + k), packed.data_ptr() + off, w, C2, Cin * 9, 1 | (9 << 2)]
            off += w * C2 * 2
        col += w
    return jobs


def decoder_conv_packed(W, widths, registered_ok):
    """Packed images (one buffer, stage-major then tap) of the decoder's conv_out weight; registered for the per-step refresh when
    the weight is owned by a flat optimizer, packed on the spot otherwise.  None when the shapes are not served."""
    if not (W.is_cuda and W.dtype == torch.float32 and W.is_contiguous() and W.dim() == 4 and W.shape[0] == 128
            and all(w == 128 for w in widths) and sum(widths) == W.shape[1]):
        return None
    key = ("dec", id(W))
    ent = _REG.get(key)
    if registered_ok and ent is not None and ent["ref"]() is W and ent["ptrs"] == (W.data_ptr(),):
        return _current(ent, (W,))["packed"]
    packed = torch.empty(sum(widths) * W.shape[0] * 9 * 2, dtype=torch.uint8, device=W.device)
    jobs = _dec_jobs(W, widths, packed)
    jd = torch.tensor(jobs, dtype=torch.int64).to(W.device)
    L.call("gdmae_tok_gemm_pack", L.ptr(jd), len(jobs) // 6, L.stream())
    packed._gd_jobs = jd
    if registered_ok:
        _REG[key] = dict(ref=weakref.ref(W), refs=[weakref.ref(W)], vers=_vers((W,)), packed=packed, ptrs=(W.data_ptr(),), jobs=jobs)
        _TABLE.clear()
    return packed


def repack_registered():
    """Refresh every registered image with ONE launch per device (called after the optimizer step)."""
    dead = [k for k, e in _REG.items() if e["ref"]() is None]
    for k in dead:
        del _REG[k]
    if dead:
        _TABLE.clear()
    if not _REG:
        return
    by_dev = {}
    for e in _REG.values():
        by_dev.setdefault(e["packed"].device, []).append(e)
    for dev, ents in by_dev.items():
        sig = tuple(id(e["packed"]) for e in ents)
        tab = _TABLE.get(dev.index)
        if tab is None or tab[0] != sig:
            flat = [v for e in ents for v in e["jobs"]]
            tab = (sig, torch.tensor(flat, dtype=torch.int64).to(dev), len(flat) // 6)
            _TABLE[dev.index] = tab
        with torch.cuda.device(dev):
            L.call("gdmae_tok_gemm_pack", L.ptr(tab[1]), tab[2], L.stream())
        for e in ents:
            ts = [r() for r in e["refs"]]
            if all(t is not None for t in ts):
                e["vers"] = _vers(ts)
