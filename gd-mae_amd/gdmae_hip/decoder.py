"""Sparse-aware generative decoder: the same arithmetic as the reference's dense decoder
(pcdet/models/backbones_3d/spt_backbone_mae.py:120-143: densify -> ConvTranspose2d(k = s) + BN2d + ReLU per
source stage -> cat -> Conv2d 3x3 + BN2d + ReLU -> gather at all pillar sites), restructured around what
is actually dense.

Exact identities used (no approximation):
* ConvTranspose2d with kernel = stride maps each input site to its own s x s output block, so on a map
  that is zero outside the active set it is one token GEMM  P = X @ W.view(Cin, s*s*Cout)  and zeros
  elsewhere.
* BatchNorm2d statistics over ALL B*Y*X sites of such a map follow from the token values alone
  (sum and sum of squares over tokens, divided by the dense count); after BN + ReLU every empty site
  holds the same per-channel constant relu(beta - gamma*mean/sigma).
* The concatenated 384-channel map is therefore "constant background + token rows": it is written once
  (fill + three row scatters into column slices) instead of 3 dense deconvs, 3 dense BN passes, 3 ReLUs
  and a cat copy.
* The 3x3 convolution is the one genuinely dense contraction (MIOpen implicit GEMM, bf16 in throughput
  mode).  Its BatchNorm needs dense statistics (one read pass, gdmae_colstats) but its output is only
  consumed at the M pillar sites, so BN + ReLU are applied to the gathered rows only; in the backward
  the dense gradient of the conv output is an affine function of the conv output plus M sparse rows.

Autograd: the token-side algebra is ordinary differentiable torch code on token-sized tensors; only
the two dense boundaries are custom Functions (BuildDenseCat, DenseBNReLUGather).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import lib as L
from . import ops

BN_EPS, BN_MOM = 1e-3, 0.01


def colstats(x2d: torch.Tensor):
    """(sum, sumsq) per column of a contiguous (R, C) fp32/bf16 device matrix, as float64 (C,) tensors."""
    R, C = x2d.shape
    assert x2d.is_contiguous() and x2d.dtype in (torch.float32, torch.bfloat16)
    out = torch.empty(2 * C, dtype=torch.float64, device=x2d.device)
    ws = torch.empty(L.load().gdmae_colstats_workspace_bytes(C), dtype=torch.uint8, device=x2d.device)
    L.call("gdmae_colstats", L.ptr(x2d), R, C, int(x2d.dtype == torch.bfloat16), L.ptr(out), L.ptr(ws), L.stream())
    return out[:C], out[C:]


def _gather_slice(table2d, idx, col0, ncol):
    es = table2d.element_size()
    out = torch.empty(idx.numel(), ncol, dtype=table2d.dtype, device=table2d.device)
    L.call("gdmae_gather_rows_strided", L.ptr(table2d), L.ptr(idx), idx.numel(), ncol * es, table2d.shape[1] * es,
           col0 * es, L.ptr(out), L.stream())
    return out


def _scatter_slice(src2d, idx, table2d, col0):
    es = table2d.element_size()
    src2d = src2d.contiguous()
    assert src2d.dtype == table2d.dtype
    L.call("gdmae_scatter_rows_strided", L.ptr(src2d), L.ptr(idx), idx.numel(), src2d.shape[1] * es,
           table2d.shape[1] * es, col0 * es, L.ptr(table2d), L.stream())


def _bf(t):
    return int(t.dtype == torch.bfloat16)


class StagesBNToDense(torch.autograd.Function):
    """All decoder source stages at once: BatchNorm2d(train) + ReLU of the (implicit) dense deconvolution maps,
    written as ONE channels-last concatenated map Z (R, sum C_i).

    forward(R, zdtype, eps, sites_0, P_0, gamma_0, beta_0, sites_1, ...): P_i (n_i, C_i) are the deconvolution
    outputs of the active sites (unique full-resolution cells sites_i); every other site of map i is zero before
    BN, so the batch statistics over all R sites are column sums of P_i and the post-BN/ReLU value of every other
    site is the constant relu(beta - gamma * mean * rstd).  Returns (Z, mean_0, var_0, mean_1, var_1, ...).
    The backward is the hand-derived BatchNorm chain rule on column sums (three row kernels per stage)."""

    @staticmethod
    def forward(ctx, R, zdtype, eps, *args):
        k = len(args) // 4
        sites, Ps, gammas, betas = args[0::4], args[1::4], args[2::4], args[3::4]
        widths = [int(P.shape[1]) for P in Ps]
        a_l, b_l, mean_l, r_l, stats_out = [], [], [], [], []
        for P, g, be in zip(Ps, gammas, betas):
            s1, s2 = colstats(P)
            mean = s1 / R
            var = (s2 / R - mean * mean).clamp_(min=0)
            r = torch.rsqrt(var + eps)
            a = g.detach().double() * r
            b = be.detach().double() - a * mean
            a_l.append(a.float()), b_l.append(b.float()), mean_l.append(mean), r_l.append(r)
            stats_out += [mean.float(), var.float()]
        Z = torch.relu(torch.cat(b_l)).to(zdtype).expand(R, sum(widths)).contiguous()
        col = 0
        for i in range(k):
            L.call("gdmae_rows_affine_relu_scatter", L.ptr(Ps[i]), _bf(Ps[i]), L.ptr(sites[i]), Ps[i].shape[0], widths[i],
                   L.ptr(a_l[i]), L.ptr(b_l[i]), L.ptr(Z), _bf(Z), Z.shape[1], col, L.stream())
            col += widths[i]
        ctx.save_for_backward(*sites, *Ps, *a_l, *b_l, *mean_l, *r_l, *[g.detach() for g in gammas])
        ctx.k, ctx.widths, ctx.R = k, widths, R
        ctx.mark_non_differentiable(*stats_out)
        return (Z, *stats_out)

    @staticmethod
    def backward(ctx, dZ, *_):
        k, R = ctx.k, ctx.R
        sv = ctx.saved_tensors
        sites, Ps, a_l, b_l, mean_l, r_l, gammas = (sv[i * k:(i + 1) * k] for i in range(7))
        dZ = dZ.contiguous()
        tot, _ = colstats(dZ)                                   # column sums over ALL sites (fp64)
        grads = [None, None, None]
        col = 0
        for i, w in enumerate(ctx.widths):
            P = Ps[i]
            n = P.shape[0]
            st = torch.empty(3 * w, dtype=torch.float64, device=P.device)
            ws = torch.empty(L.load().gdmae_rows_bwd_stats_workspace_bytes(w), dtype=torch.uint8, device=P.device)
            L.call("gdmae_rows_bwd_stats", L.ptr(P), _bf(P), L.ptr(sites[i]), n, w, L.ptr(a_l[i]), L.ptr(b_l[i]), L.ptr(dZ),
                   _bf(dZ), dZ.shape[1], col, L.ptr(st), L.ptr(ws), L.stream())
            s_dh, s_dhp, s_g = st[:w], st[w:2 * w], st[2 * w:]
            a, b, mean, r, g = a_l[i].double(), b_l[i].double(), mean_l[i], r_l[i], gammas[i].double()
            gbg = tot[col:col + w] - s_g                           # every non-token site shares the background value
            db = s_dh + gbg * (b > 0)
            da = s_dhp - db * mean                                 # total derivative w.r.t. a (b = beta - a * mean)
            dgamma = da * r
            dv = -0.5 * (da * g) * r * r * r                       # a = gamma * rsqrt(var + eps)
            dmu = -db * a - 2.0 * mean * dv                        # var = s2 / R - mean^2
            c0, c1 = (dmu / R).float(), (2.0 * dv / R).float()
            dP = torch.empty_like(P)
            L.call("gdmae_rows_bwd", L.ptr(P), _bf(P), L.ptr(sites[i]), n, w, L.ptr(a_l[i]), L.ptr(b_l[i]), L.ptr(c0), L.ptr(c1),
                   L.ptr(dZ), _bf(dZ), dZ.shape[1], col, L.ptr(dP), _bf(dP), L.stream())
            grads += [None, dP, dgamma.to(gammas[i].dtype), db.to(gammas[i].dtype)]
            col += w
        return tuple(grads)


class DenseBNReLUGather(torch.autograd.Function):
    """relu(BatchNorm2d_train(Y))[sites]: batch statistics over all R rows of the channels-last map Y (R, C),
    affine + ReLU applied only to the gathered rows.  Returns (out (M, C) fp32, mean, biased var)."""

    @staticmethod
    def forward(ctx, Y, gamma, beta, sites, eps):
        R, C = Y.shape
        s1, s2 = colstats(Y)
        mean64 = s1 / R
        var64 = (s2 / R - mean64 * mean64).clamp_(min=0)
        mean, var = mean64.float(), var64.float()
        inv = torch.rsqrt(var + eps)
        rows = ops.gather_rows_raw(Y, sites).float()
        yhat = (rows - mean) * inv
        out = torch.relu(yhat * gamma + beta)
        ctx.save_for_backward(Y, sites, yhat, out > 0, inv, mean, gamma)
        ctx.mark_non_differentiable(mean, var)
        return out, mean, var

    @staticmethod
    def backward(ctx, dout, _dm, _dv):
        Y, sites, yhat, mask, inv, mean, gamma = ctx.saved_tensors
        R = Y.shape[0]
        g = dout * mask
        dgamma = (g * yhat).sum(0)
        dbeta = g.sum(0)
        dyh = g * gamma
        m1 = dyh.sum(0, dtype=torch.float64) / R
        m2 = (dyh * yhat).sum(0, dtype=torch.float64) / R
        inv64, mean64 = inv.double(), mean.double()
        # dy = inv * (dyhat - m1 - yhat * m2),  yhat = (y - mean) * inv   ->   dense part = k0 + k1 * y
        k1 = (-(inv64 * inv64) * m2)
        k0 = (-inv64 * m1 - k1 * mean64)
        dY = torch.addcmul(k0.to(Y.dtype), Y, k1.to(Y.dtype))
        rows = ops.gather_rows_raw(dY, sites).float() + dyh * inv
        ops.scatter_rows_raw(rows.to(Y.dtype), sites, dY)
        return dY, dgamma, dbeta, None, None


class DenseConv3x3(torch.autograd.Function):
    """conv2d(k3 s1 p1, no bias) of the channels-last map with the weight's bf16 shadow under autocast.  An explicit
    Function because the shadow (gdmae_hip.optim) is a detached copy: the weight gradient has to be routed back to
    the fp32 parameter by hand (MIOpen backward-data + backward-weights in one convolution_backward call)."""

    @staticmethod
    def forward(ctx, zin, weight):
        cdt = torch.bfloat16 if torch.is_autocast_enabled() else weight.dtype
        wc = ops.shadow(weight, cdt)
        zc = zin.to(cdt)
        ctx.save_for_backward(zc, wc)
        ctx.in_dtype = zin.dtype
        with torch.autocast("cuda", enabled=False):
            return F.conv2d(zc, wc, None, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        zc, wc = ctx.saved_tensors
        dz, dw, _ = torch.ops.aten.convolution_backward(dy.to(zc.dtype), zc, wc, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                        [ctx.needs_input_grad[0], True, False])
        return (None if dz is None else dz.to(ctx.in_dtype)), dw.float()


def _update_running(bn, mean, var_biased, n):
    """nn.BatchNorm running statistics (momentum 0.01, unbiased variance) for checkpoint parity."""
    if bn.running_mean is None:
        return
    with torch.no_grad():
        bn.running_mean.mul_(1 - bn.momentum).add_(mean.to(bn.running_mean.dtype), alpha=bn.momentum)
        bn.running_var.mul_(1 - bn.momentum).add_(var_biased.to(bn.running_var.dtype) * (n / max(n - 1, 1)), alpha=bn.momentum)
        bn.num_batches_tracked += 1


def upsampled_sites(stage_plan, s: int, Y: int, X: int) -> torch.Tensor:
    """(n_tok * s*s,) int32 full-resolution cell of every (token, dy, dx) of a stride-s stage."""
    c = stage_plan.tok_cell
    if s == 1:
        return c
    x = c % stage_plan.X
    r = torch.div(c, stage_plan.X, rounding_mode='floor')
    y = r % stage_plan.Y
    b = torch.div(r, stage_plan.Y, rounding_mode='floor')
    d = torch.arange(s, device=c.device, dtype=c.dtype)
    yy = (y * s).view(-1, 1, 1) + d.view(1, s, 1)
    xx = (x * s).view(-1, 1, 1) + d.view(1, 1, s)
    return ((b.view(-1, 1, 1) * Y + yy) * X + xx).reshape(-1).contiguous()


def sparse_decoder(model_cfg, deblocks, conv_out, hidden, pillar_cell, B, Y, X, want_dense=False):
    """hidden: list of SparseConvTensor per stage.  Returns (features at the pillar sites (M, C) fp32,
    dense spatial_features (B, C, Y, X) or None)."""
    R = B * Y * X
    args, bns = [], []
    cdt = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
    for i, src in enumerate(model_cfg.FEATURES_SOURCE):
        h = hidden[int(src[-1]) - 1]
        sp = h.stage_plan
        deconv, bn = deblocks[i][0], deblocks[i][1]
        s = int(deconv.stride[0])
        assert deconv.kernel_size == (s, s) and sp.Y * s == Y and sp.X * s == X and deconv.bias is None
        cin, cout = deconv.weight.shape[0], deconv.weight.shape[1]
        wmat = deconv.weight.permute(0, 2, 3, 1).reshape(cin, s * s * cout)      # columns ordered (dy, dx, c)
        P = ops.linear(h.features, wmat.t()).view(-1, cout)                      # (n_tok * s*s, cout)
        args += [upsampled_sites(sp, s, Y, X), P, bn.weight, bn.bias]
        bns.append(bn)
    outs = StagesBNToDense.apply(R, cdt, bns[0].eps, *args)                      # Z (R, 384) channels-last
    Z = outs[0]
    for i, bn in enumerate(bns):
        if bn.training:
            _update_running(bn, outs[1 + 2 * i], outs[2 + 2 * i], R)
    conv, bn2 = conv_out[0], conv_out[1]
    zin = Z.view(B, Y, X, -1).permute(0, 3, 1, 2)                                # NCHW view of NHWC memory
    y2 = DenseConv3x3.apply(zin, conv.weight)
    y2 = y2.permute(0, 2, 3, 1)
    if not y2.is_contiguous():
        y2 = y2.contiguous()
    y2 = y2.view(R, -1)
    out, mean2, var2 = DenseBNReLUGather.apply(y2, bn2.weight, bn2.bias, pillar_cell, bn2.eps)
    if bn2.training:
        _update_running(bn2, mean2, var2, R)
    dense = None
    if want_dense:
        with torch.no_grad():
            a2 = bn2.weight * torch.rsqrt(var2 + bn2.eps)
            dense = torch.relu(y2.float() * a2 + (bn2.bias - a2 * mean2)).view(B, Y, X, -1).permute(0, 3, 1, 2)
    return out, dense
