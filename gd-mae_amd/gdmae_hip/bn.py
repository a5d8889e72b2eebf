"""BatchNorm(train) bookkeeping on the device, one launch per direction.

Every BatchNorm of the path (DynVFE BatchNorm1d x2, the sparse-conv blocks' BatchNorm1d, the decoder's four
BatchNorm2d; reference network_utils.py:7-21, spconv_utils.py post_act_block, spt_backbone_mae.py:30-52) needs a
handful of per-channel vectors around its row kernels: mean / rstd / folded affine in the forward (+ the
running-statistics update nn.BatchNorm performs), dgamma / dbeta / the two chain-rule coefficients in the backward.
As torch vector ops that is ~25 launches of 128..384-element kernels per BatchNorm per direction (several hundred
per training step, each costing more host time than GPU time); here it is gdmae_bn_fold / gdmae_bn_bwd_coeffs.
"""
from __future__ import annotations

import torch

from . import lib as L
from . import ops


def fold(x2d: torch.Tensor, count: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float, bn=None, partials=None):
    """Column statistics of the contiguous (R, C) fp32/bf16 matrix over ``count`` samples.
    -> (stats f64[2C] = mean|rstd, ab f32[2C] = a|b with y = a*x + b, mv f32[2C] = mean|biased var).
    ``bn``: the nn.BatchNorm module whose running statistics are updated in the same launch (training mode).
    ``partials``: (nblk, 2, C) fp32 partial {sums, sums of squares} of x2d's columns that its producer left (the dense convolution's
    epilogue, gdmae_hip.dense): the fold runs from them, without the pass over x2d."""
    if isinstance(bn, torch.nn.SyncBatchNorm):
        # tools/train.py:120-121 (--sync_bn, off by default in every shipped script): statistics over all ranks need two small
        # collectives per BatchNorm and direction; the fused row kernels compute per-GPU statistics (the reference default) and must
        # not silently do so under a module that promises synchronised ones
        raise NotImplementedError("SyncBatchNorm is not supported by the fused BatchNorm kernels (per-GPU statistics, the reference's "
                                  "default); run without --sync_bn")
    R, C = x2d.shape
    assert x2d.is_contiguous() and x2d.dtype in (torch.float32, torch.bfloat16)
    assert gamma.dtype == torch.float32 and beta.dtype == torch.float32
    dev = x2d.device
    stats = torch.empty(2 * C, dtype=torch.float64, device=dev)
    ab = torch.empty(2 * C, dtype=torch.float32, device=dev)
    mv = torch.empty(2 * C, dtype=torch.float32, device=dev)
    ws = torch.empty(L.load().gdmae_colstats_workspace_bytes(C), dtype=torch.uint8, device=dev)
    rm = rv = nb = None
    mom = 0.0
    if bn is not None and bn.training and bn.running_mean is not None:
        rm, rv, nb, mom = bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum)
        assert rm.dtype == torch.float32 and rv.dtype == torch.float32 and nb.dtype == torch.int64
    if partials is not None:
        assert partials.dtype == torch.float32 and partials.is_contiguous() and partials.shape[1:] == (2, C), partials.shape
        L.call("gdmae_bn_fold_partials", L.ptr(partials), partials.shape[0], C, float(count), L.ptr(gamma), L.ptr(beta), float(eps), mom,
               L.ptr(rm) if rm is not None else None, L.ptr(rv) if rv is not None else None, L.ptr(nb) if nb is not None else None,
               L.ptr(stats), L.ptr(ab), L.ptr(mv), L.stream())
        return stats, ab, mv
    L.call("gdmae_bn_fold", L.ptr(x2d), R, C, int(x2d.dtype == torch.bfloat16), float(count), L.ptr(gamma), L.ptr(beta), float(eps),
           mom, L.ptr(rm) if rm is not None else None, L.ptr(rv) if rv is not None else None,
           L.ptr(nb) if nb is not None else None, L.ptr(stats), L.ptr(ab), L.ptr(mv), L.ptr(ws), L.stream())
    return stats, ab, mv


def bwd_coeffs(st: torch.Tensor, n_st: int, stats: torch.Tensor, ab: torch.Tensor, gamma: torch.Tensor, count: int,
               tot: torch.Tensor | None = None, direct=(None, None)):
    """st f64[n_st*C] column sums {dh, dh*x, (g)} -> (dgamma, dbeta, c01 f32[2C]); with ``direct`` = flat-gradient
    views of (gamma, beta) the parameter gradients are accumulated there and (None, None, c01) is returned."""
    C = gamma.numel()
    dev = gamma.device
    c01 = torch.empty(2 * C, dtype=torch.float32, device=dev)
    dg, db = direct
    acc = int(dg is not None and db is not None)
    if not acc:
        dgb = torch.empty(2 * C, dtype=torch.float32, device=dev)
        dg, db = dgb[:C], dgb[C:]
    if isinstance(st, tuple):         # (workspace of gdmae_rows_bwd_stats called with out = NULL, its partial-row count)
        part, nblk = st
        L.call("gdmae_bn_bwd_coeffs_rows", L.ptr(part), nblk, n_st, L.ptr(stats), L.ptr(ab), L.ptr(gamma), C, float(count),
               L.ptr(tot) if tot is not None else None, L.ptr(dg), L.ptr(db), acc, L.ptr(c01), L.stream())
    else:
        L.call("gdmae_bn_bwd_coeffs", L.ptr(st), n_st, L.ptr(stats), L.ptr(ab), L.ptr(gamma), C, float(count),
               L.ptr(tot) if tot is not None else None, L.ptr(dg), L.ptr(db), acc, L.ptr(c01), L.stream())
    return (None, None, c01) if acc else (dg, db, c01)


def direct_pair(gamma, beta):
    return ops.direct_grad(gamma), ops.direct_grad(beta)
