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
* The forward 3x3 convolution is the one genuinely dense contraction (MIOpen implicit GEMM, bf16 in throughput
  mode).  Its BatchNorm needs dense statistics (one read pass, gdmae_colstats) but its output is only
  consumed at the M pillar sites, so BN + ReLU are applied to the gathered rows only.
* In the backward the gradient of the conv output is an affine function of the conv output plus M sparse
  rows, and the conv input gradient is only needed where the input is not background: per source stage the
  9 shifted output-gradient rows of its active sites are gathered (gdmae_conv3x3_grad_taps) and two GEMMs give
  the input-gradient rows and the weight gradient; the background's share (input gradient summed over all
  other sites, weight gradient against the constant) is a closed form of 9 border-region sums.

Autograd: one hand-derived Function (DecoderHead) from the token-side deconvolution rows to the pillar rows.
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


# S_k = sum of the output gradient over the sites s whose tap s+k lies inside the map, from the 9 region sums
# [all, row 0, row H-1, col 0, col W-1, corner(0,0), corner(0,W-1), corner(H-1,0), corner(H-1,W-1)]
_TAP_REGION = torch.zeros(9, 9, dtype=torch.float64)
for _ky in (-1, 0, 1):
    for _kx in (-1, 0, 1):
        _k = (_ky + 1) * 3 + (_kx + 1)
        _TAP_REGION[_k, 0] = 1
        if _ky:
            _TAP_REGION[_k, 1 if _ky < 0 else 2] = -1
        if _kx:
            _TAP_REGION[_k, 3 if _kx < 0 else 4] = -1
        if _ky and _kx:
            _TAP_REGION[_k, 5 + (0 if _ky < 0 else 2) + (0 if _kx < 0 else 1)] = 1


class DecoderHead(torch.autograd.Function):
    """The whole generative-decoder head on the token rows (reference spt_backbone_mae.py:30-52,125-135):
    ConvTranspose2d(k=s) outputs P_i of the active sites -> BatchNorm2d(train)+ReLU of the (implicit) dense maps ->
    channel concat Z (R, sum C_i) channels-last -> dense 3x3 conv_out (MIOpen) -> BatchNorm2d(train)+ReLU evaluated
    only at the pillar sites.

    forward(geom, conv_w, gamma2, beta2, pillar_cell, cell2pillar, sites_0, P_0, gamma_0, beta_0, sites_1, ...)
      geom = (B, H, W, eps1, eps2, compute dtype).  Every non-active site of map i is zero before its BatchNorm, so
      the batch statistics over all R sites are column sums of P_i and every other site holds the constant
      background relu(beta - gamma*mean*rstd).  Returns (out (M, C2) fp32, Y (R, C2) [non-differentiable, for the
      dense spatial_features], mean2, var2, mean_0, var_0, ...).
    backward: hand-derived.  The conv output gradient is dY = k0 + k1*Y + rows-at-pillar-sites (BatchNorm2d chain
      rule), which is dense - but the input gradient is only needed at the active sites of each stage (everything
      else is the shared background whose gradient is a border-corrected closed form), and the weight gradient
      splits into background x region-sums plus active-site rows.  So instead of MIOpen's dense backward-data and
      backward-weights over all R sites, gdmae_conv3x3_grad_taps gathers the 9 shifted dY rows of the active sites
      and two GEMMs per stage finish the job; dY / dZ are never materialised."""

    @staticmethod
    def forward(ctx, geom, conv_w, gamma2, beta2, pillar_cell, cell2pillar, *args):
        B, H, W, eps1, eps2, cdt = geom
        R = B * H * W
        k = len(args) // 4
        sites, Ps, gammas, betas = args[0::4], args[1::4], args[2::4], args[3::4]
        widths = [int(P.shape[1]) for P in Ps]
        a_l, b_l, mean_l, r_l, stats_out = [], [], [], [], []
        for P, g, be in zip(Ps, gammas, betas):
            s1, s2 = colstats(P)
            mean = s1 / R
            var = (s2 / R - mean * mean).clamp_(min=0)
            r = torch.rsqrt(var + eps1)
            a = g.detach().double() * r
            b = be.detach().double() - a * mean
            a_l.append(a.float()), b_l.append(b.float()), mean_l.append(mean), r_l.append(r)
            stats_out += [mean.float(), var.float()]
        bgz = torch.relu(torch.cat(b_l)).to(cdt)                      # background value of every non-active site
        Z = bgz.expand(R, sum(widths)).contiguous()
        col = 0
        for i in range(k):
            L.call("gdmae_rows_affine_relu_scatter", L.ptr(Ps[i]), _bf(Ps[i]), L.ptr(sites[i]), Ps[i].shape[0], widths[i],
                   L.ptr(a_l[i]), L.ptr(b_l[i]), L.ptr(Z), _bf(Z), Z.shape[1], col, L.stream())
            col += widths[i]
        wc = ops.shadow(conv_w, cdt)
        with torch.autocast("cuda", enabled=False):
            y2 = F.conv2d(Z.view(B, H, W, -1).permute(0, 3, 1, 2), wc, None, 1, 1)
        y2 = y2.permute(0, 2, 3, 1)
        if not y2.is_contiguous():
            y2 = y2.contiguous()
        y2 = y2.view(R, -1)
        s1, s2 = colstats(y2)
        mean64 = s1 / R
        var64 = (s2 / R - mean64 * mean64).clamp_(min=0)
        mean2, var2 = mean64.float(), var64.float()
        inv = torch.rsqrt(var2 + eps2)
        yhat = (ops.gather_rows_raw(y2, pillar_cell).float() - mean2) * inv
        out = torch.relu(yhat * gamma2 + beta2)
        ctx.save_for_backward(*sites, *Ps, *a_l, *b_l, *mean_l, *r_l, *[g.detach() for g in gammas], Z, y2, bgz, s1,
                              pillar_cell, cell2pillar, yhat, out > 0, inv, mean2, gamma2.detach(), conv_w.detach())
        ctx.k, ctx.widths, ctx.geom = k, widths, geom
        ctx.mark_non_differentiable(y2, mean2, var2, *stats_out)
        return (out, y2, mean2, var2, *stats_out)

    @staticmethod
    def backward(ctx, dout, *_):
        k = ctx.k
        B, H, W, eps1, eps2, cdt = ctx.geom
        R = B * H * W
        sv = ctx.saved_tensors
        sites, Ps, a_l, b_l, mean_l, r_l, gammas = (sv[i * k:(i + 1) * k] for i in range(7))
        Z, y2, bgz, s1, pillar_cell, cell2pillar, yhat, mask, inv, mean2, gamma2, conv_w = sv[7 * k:]
        C2, Cin = y2.shape[1], Z.shape[1]
        dev = y2.device
        # ---- BatchNorm2d #2 (+ReLU) at the pillar rows: dY = k0 + k1*Y + rows[pillar sites]
        g = dout * mask
        dgamma2 = (g * yhat).sum(0)
        dbeta2 = g.sum(0)
        dyh = g * gamma2
        m1 = dyh.sum(0, dtype=torch.float64) / R
        m2 = (dyh * yhat).sum(0, dtype=torch.float64) / R
        inv64, mean64 = inv.double(), mean2.double()
        k1 = -(inv64 * inv64) * m2
        k0 = -inv64 * m1 - k1 * mean64
        rows = (dyh * inv).contiguous()
        k0f, k1f = k0.float(), k1.float()
        # ---- region sums of dY (all / border rows / border columns / corners) -> S_k per tap
        Yv = y2.view(B, H, W, C2)
        f64 = torch.float64
        regY = torch.stack([s1, Yv[:, 0].sum((0, 1), dtype=f64), Yv[:, H - 1].sum((0, 1), dtype=f64),
                            Yv[:, :, 0].sum((0, 1), dtype=f64), Yv[:, :, W - 1].sum((0, 1), dtype=f64),
                            Yv[:, 0, 0].sum(0, dtype=f64), Yv[:, 0, W - 1].sum(0, dtype=f64),
                            Yv[:, H - 1, 0].sum(0, dtype=f64), Yv[:, H - 1, W - 1].sum(0, dtype=f64)])
        px = pillar_cell % W
        py = torch.div(pillar_cell, W, rounding_mode='floor') % H
        y0, yl, x0, xl = py == 0, py == H - 1, px == 0, px == W - 1
        pm = torch.stack([torch.ones_like(y0), y0, yl, x0, xl, y0 & x0, y0 & xl, yl & x0, yl & xl]).to(rows.dtype)
        regR = (pm @ rows).double()
        cnt = torch.tensor([R, B * W, B * W, B * H, B * H, B, B, B, B], dtype=f64, device=dev)
        regD = cnt[:, None] * k0[None, :] + regY * k1[None, :] + regR
        S = _TAP_REGION.to(dev) @ regD                                     # (9, C2)
        Wk = conv_w.permute(2, 3, 0, 1).reshape(9, C2, Cin)              # W_k[o, i], k = (ky+1)*3 + (kx+1)
        tot = torch.einsum('ko,koi->i', S, Wk.double())                   # column sums of dZ over ALL sites
        dWk = (S[:, :, None] * bgz.double()[None, None, :]).float()       # background part of the weight gradient
        Wd = Wk.to(cdt)
        grads = [None] * 6
        dW_rows = []
        col = 0
        for i, w in enumerate(ctx.widths):
            P = Ps[i]
            n = P.shape[0]
            G = torch.empty(n, 9 * C2, dtype=cdt, device=dev)
            L.call("gdmae_conv3x3_grad_taps", L.ptr(y2), _bf(y2), L.ptr(k0f), L.ptr(k1f), L.ptr(rows), L.ptr(cell2pillar),
                   L.ptr(sites[i]), n, H, W, C2, L.ptr(G), L.stream())
            dX = G @ Wd[:, :, col:col + w].reshape(9 * C2, w)             # dZ rows of this stage's active sites
            Zd = _gather_slice(Z, sites[i], col, w) - bgz[col:col + w]
            dW_rows.append(ops.splitk_tn(G, Zd))                          # (9*C2, w) fp32
            del G
            st = torch.empty(3 * w, dtype=torch.float64, device=dev)
            ws = torch.empty(L.load().gdmae_rows_bwd_stats_workspace_bytes(w), dtype=torch.uint8, device=dev)
            L.call("gdmae_rows_bwd_stats", L.ptr(P), _bf(P), None, n, w, L.ptr(a_l[i]), L.ptr(b_l[i]), L.ptr(dX), _bf(dX), w, 0,
                   L.ptr(st), L.ptr(ws), L.stream())
            s_dh, s_dhp, s_g = st[:w], st[w:2 * w], st[2 * w:]
            a, b, mean, r, gm = a_l[i].double(), b_l[i].double(), mean_l[i], r_l[i], gammas[i].double()
            gbg = tot[col:col + w] - s_g                           # every non-active site shares the background value
            db = s_dh + gbg * (b > 0)
            da = s_dhp - db * mean                                 # total derivative w.r.t. a (b = beta - a * mean)
            dgamma = da * r
            dv = -0.5 * (da * gm) * r * r * r                      # a = gamma * rsqrt(var + eps)
            dmu = -db * a - 2.0 * mean * dv                        # var = s2 / R - mean^2
            c0, c1 = (dmu / R).float(), (2.0 * dv / R).float()
            dP = torch.empty_like(P)
            L.call("gdmae_rows_bwd", L.ptr(P), _bf(P), None, n, w, L.ptr(a_l[i]), L.ptr(b_l[i]), L.ptr(c0), L.ptr(c1), L.ptr(dX),
                   _bf(dX), w, 0, L.ptr(dP), _bf(dP), L.stream())
            grads += [None, dP, dgamma.to(gammas[i].dtype), db.to(gammas[i].dtype)]
            col += w
        dWk = dWk + torch.cat(dW_rows, dim=1).view(9, C2, Cin)
        grads[1] = dWk.permute(1, 2, 0).reshape(C2, Cin, 3, 3).to(conv_w.dtype)
        grads[2], grads[3] = dgamma2, dbeta2
        return tuple(grads)


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


def sparse_decoder(model_cfg, deblocks, conv_out, hidden, pillar_cell, cell2pillar, B, Y, X, want_dense=False):
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
    conv, bn2 = conv_out[0], conv_out[1]
    outs = DecoderHead.apply((B, Y, X, bns[0].eps, bn2.eps, cdt), conv.weight, bn2.weight, bn2.bias, pillar_cell, cell2pillar,
                             *args)
    out, y2, mean2, var2 = outs[:4]
    for i, bn in enumerate(bns):
        if bn.training:
            _update_running(bn, outs[4 + 2 * i], outs[5 + 2 * i], R)
    if bn2.training:
        _update_running(bn2, mean2, var2, R)
    dense = None
    if want_dense:
        with torch.no_grad():
            a2 = bn2.weight * torch.rsqrt(var2 + bn2.eps)
            dense = torch.relu(y2.float() * a2 + (bn2.bias - a2 * mean2)).view(B, Y, X, -1).permute(0, 3, 1, 2)
    return out, dense
