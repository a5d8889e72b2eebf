"""Fused BatchNorm1d(train) + ReLU (+ per-pillar max) for the DynVFE point MLP.

Replaces, per Linear of ``dvfe_mlps`` (reference pcdet/models/backbones_3d/vfe/dyn_vfe.py:107-109,
pcdet/models/model_utils/network_utils.py:7-21): BatchNorm1d statistics + normalisation + ReLU as separate
passes over the (N, C) point activations (N = 1.4 M points per 8-frame batch), and for the last layer the
``torch_scatter.scatter_max`` pass.  Here each Linear output is read once for the column statistics
(gdmae_colstats) and once by a row kernel that applies the folded affine a*x+b and the ReLU (and, for the last
layer, reduces straight to the per-pillar maximum, so the normalised (N, 128) tensor is never written).
The backward uses the BatchNorm chain rule on column sums (same algebra as gdmae_hip/decoder.py): one
statistics pass and one dx pass per layer; for the max layer the statistics only touch the arg-max points.
"""
from __future__ import annotations

import torch

from . import lib as L
from .decoder import colstats


def _bf(t):
    return int(t.dtype == torch.bfloat16)


def _bn_fold(x, gamma, beta, eps):
    n = x.shape[0]
    s1, s2 = colstats(x)
    mean = s1 / n
    var = (s2 / n - mean * mean).clamp_(min=0)
    r = torch.rsqrt(var + eps)
    a = gamma.detach().double() * r
    b = beta.detach().double() - a * mean
    return mean, var, r, a.float(), b.float()


def _bn_chain(s_dh, s_dhx, mean, r, a, gamma, n):
    """BatchNorm backward on column sums: returns (dgamma, dbeta, c0, c1) with dx = a*dh + c0 + c1*x."""
    da = s_dhx - s_dh * mean                  # total derivative w.r.t. a (b = beta - a * mean)
    dgamma = da * r
    dv = -0.5 * (da * gamma.double()) * r * r * r
    dmu = -s_dh * a.double() - 2.0 * mean * dv
    return dgamma, s_dh, (dmu / n).float(), (2.0 * dv / n).float()


class BNReLURows(torch.autograd.Function):
    """relu(BatchNorm1d_train(x)) for x (N, C) fp32/bf16; returns (out [x.dtype], mean, biased var)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x = x.contiguous()
        n, C = x.shape
        mean, var, r, a, b = _bn_fold(x, gamma, beta, eps)
        out = torch.empty_like(x)
        L.call("gdmae_rows_affine_relu_scatter", L.ptr(x), _bf(x), None, n, C, L.ptr(a), L.ptr(b), L.ptr(out), _bf(out), C, 0,
               L.stream())
        ctx.save_for_backward(x, a, b, mean, r, gamma.detach())
        mf, vf = mean.float(), var.float()
        ctx.mark_non_differentiable(mf, vf)
        return out, mf, vf

    @staticmethod
    def backward(ctx, g, _m, _v):
        x, a, b, mean, r, gamma = ctx.saved_tensors
        n, C = x.shape
        g = g.contiguous()
        st = torch.empty(3 * C, dtype=torch.float64, device=x.device)
        ws = torch.empty(L.load().gdmae_rows_bwd_stats_workspace_bytes(C), dtype=torch.uint8, device=x.device)
        L.call("gdmae_rows_bwd_stats", L.ptr(x), _bf(x), None, n, C, L.ptr(a), L.ptr(b), L.ptr(g), _bf(g), C, 0, L.ptr(st),
               L.ptr(ws), L.stream())
        dgamma, dbeta, c0, c1 = _bn_chain(st[:C], st[C:2 * C], mean, r, a, gamma, n)
        dx = torch.empty_like(x)
        L.call("gdmae_rows_bwd", L.ptr(x), _bf(x), None, n, C, L.ptr(a), L.ptr(b), L.ptr(c0), L.ptr(c1), L.ptr(g), _bf(g), C, 0,
               L.ptr(dx), _bf(dx), L.stream())
        return dx, dgamma.to(gamma.dtype), dbeta.to(gamma.dtype), None


class BNReLUSegmentMax(torch.autograd.Function):
    """max over each pillar's points of relu(BatchNorm1d_train(x)); returns (out (M, C) fp32, mean, biased var)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, pt_off, pillar_pts, inverse32):
        x = x.contiguous()
        n, C = x.shape
        M = pt_off.numel() - 1
        mean, var, r, a, b = _bn_fold(x, gamma, beta, eps)
        out = torch.empty(M, C, dtype=torch.float32, device=x.device)
        arg = torch.empty(M, C, dtype=torch.int32, device=x.device)
        L.call("gdmae_segment_max_affine", L.ptr(x), _bf(x), L.ptr(pt_off), L.ptr(pillar_pts), M, C, L.ptr(a), L.ptr(b),
               L.ptr(out), L.ptr(arg), L.stream())
        ctx.save_for_backward(x, out, arg, inverse32, a, mean, r, gamma.detach())
        mf, vf = mean.float(), var.float()
        ctx.mark_non_differentiable(mf, vf)
        return out, mf, vf

    @staticmethod
    def backward(ctx, g, _m, _v):
        x, out, arg, inv, a, mean, r, gamma = ctx.saved_tensors
        n, C = x.shape
        M = out.shape[0]
        g = g.float().contiguous()
        st = torch.empty(2 * C, dtype=torch.float64, device=x.device)
        ws = torch.empty(L.load().gdmae_rows_bwd_stats_workspace_bytes(C), dtype=torch.uint8, device=x.device)
        L.call("gdmae_segmax_bwd_stats", L.ptr(x), _bf(x), L.ptr(out), L.ptr(arg), L.ptr(g), M, C, L.ptr(st), L.ptr(ws), L.stream())
        dgamma, dbeta, c0, c1 = _bn_chain(st[:C], st[C:], mean, r, a, gamma, n)
        dx = torch.empty_like(x)
        L.call("gdmae_segmax_bn_bwd", L.ptr(x), _bf(x), L.ptr(out), L.ptr(arg), L.ptr(g), L.ptr(inv), n, C, L.ptr(a), L.ptr(c0),
               L.ptr(c1), L.ptr(dx), _bf(dx), L.stream())
        return dx, dgamma.to(gamma.dtype), dbeta.to(gamma.dtype), None, None, None, None
