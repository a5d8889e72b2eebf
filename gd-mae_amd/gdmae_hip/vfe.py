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

from . import bn as gbn
from . import lib as L
from . import ops


def _bf(t):
    return int(t.dtype == torch.bfloat16)


class BNReLURows(torch.autograd.Function):
    """relu(BatchNorm1d_train(x)) [+ residual] for x (N, C) fp32/bf16; returns (out [x.dtype], mean, biased var).
    ``bn``: the nn.BatchNorm1d / 2d module (running statistics updated in the statistics launch) or None; ``residual``: (N, C) rows
    of x's dtype added AFTER the ReLU in the same pass (identity shortcut of a dense block; its gradient is the output gradient)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, bn=None, residual=None, partials=None):
        x = x.contiguous()
        n, C = x.shape
        stats, ab, mv = gbn.fold(x, n, gamma, beta, eps, bn, partials)       # partials: column-sum rows x's producer left (bn.fold)
        out = torch.empty_like(x)
        if residual is not None:
            assert residual.shape == x.shape and residual.dtype == x.dtype and residual.is_contiguous() and C % 8 == 0
            L.call("gdmae_rows_affine_relu_add", L.ptr(x), _bf(x), n, C, L.ptr(ab), L.ptr(ab[C:]), L.ptr(residual), L.ptr(out), _bf(out),
                   L.stream())
        else:
            L.call("gdmae_rows_affine_relu_scatter", L.ptr(x), _bf(x), None, n, C, L.ptr(ab), L.ptr(ab[C:]), L.ptr(out), _bf(out), C, 0,
                   L.stream())
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, ab, stats, gamma.detach())
        ctx.direct = gbn.direct_pair(gamma, beta)
        mf, vf = mv[:C], mv[C:]
        ctx.mark_non_differentiable(mf, vf)
        ctx.set_materialize_grads(False)      # no zero tensors for the unused gradients of those outputs
        return out, mf, vf

    @staticmethod
    def backward(ctx, g, _m, _v):
        if g is None:
            return None, None, None, None, None, None, None
        x, ab, stats, gamma = ctx.saved_tensors
        n, C = x.shape
        g = g.contiguous()
        st = torch.empty(3 * C, dtype=torch.float64, device=x.device)
        ws = torch.empty(L.load().gdmae_rows_bwd_stats_workspace_bytes(C), dtype=torch.uint8, device=x.device)
        L.call("gdmae_rows_bwd_stats", L.ptr(x), _bf(x), None, n, C, L.ptr(ab), L.ptr(ab[C:]), L.ptr(g), _bf(g), C, 0, L.ptr(st),
               L.ptr(ws), L.stream())
        dgamma, dbeta, c01 = gbn.bwd_coeffs(st, 3, stats, ab, gamma, n, None, ctx.direct)
        dx = torch.empty_like(x)
        L.call("gdmae_rows_bwd", L.ptr(x), _bf(x), None, n, C, L.ptr(ab), L.ptr(ab[C:]), L.ptr(c01), L.ptr(c01[C:]), L.ptr(g),
               _bf(g), C, 0, L.ptr(dx), _bf(dx), L.stream())
        return dx, dgamma, dbeta, None, None, (g if ctx.has_res else None), None


class BNReLURowsCat(torch.autograd.Function):
    """cat([relu(BatchNorm_train(x_i)) for i], dim=1) for row matrices x_i (N, C_i) of one dtype, written straight into the column
    slices of the (N, sum C_i) result (the row kernels take a row pitch and a first column); the backward reads its slice of the
    output gradient in place.  apply(bns, eps, x_0, gamma_0, beta_0, x_1, ...) -> (N, sum C_i)."""

    @staticmethod
    def forward(ctx, bns, eps, *args):
        xs = [a.contiguous() for a in args[0::3]]
        gammas, betas = args[1::3], args[2::3]
        n, Ct = xs[0].shape[0], sum(x.shape[1] for x in xs)
        assert all(x.shape[0] == n and x.dtype == xs[0].dtype and x.shape[1] % 8 == 0 for x in xs)
        out = torch.empty(n, Ct, dtype=xs[0].dtype, device=xs[0].device)
        saved, col = [], 0
        for x, g_, b_, bn in zip(xs, gammas, betas, bns):
            C = x.shape[1]
            stats, ab, _ = gbn.fold(x, n, g_, b_, eps, bn)
            L.call("gdmae_rows_affine_relu_scatter", L.ptr(x), _bf(x), None, n, C, L.ptr(ab), L.ptr(ab[C:]), L.ptr(out), _bf(out), Ct, col,
                   L.stream())
            saved += [x, ab, stats, g_.detach()]
            col += C
        ctx.save_for_backward(*saved)
        ctx.direct = [gbn.direct_pair(g_, b_) for g_, b_ in zip(gammas, betas)]
        return out

    @staticmethod
    def backward(ctx, g):
        sv = ctx.saved_tensors
        g = g.contiguous()
        n, Ct = g.shape
        grads, col = [None, None], 0
        for i in range(len(sv) // 4):
            x, ab, stats, gamma = sv[4 * i:4 * i + 4]
            C = x.shape[1]
            st = torch.empty(3 * C, dtype=torch.float64, device=x.device)
            ws = torch.empty(L.load().gdmae_rows_bwd_stats_workspace_bytes(C), dtype=torch.uint8, device=x.device)
            L.call("gdmae_rows_bwd_stats", L.ptr(x), _bf(x), None, n, C, L.ptr(ab), L.ptr(ab[C:]), L.ptr(g), _bf(g), Ct, col, L.ptr(st),
                   L.ptr(ws), L.stream())
            dgamma, dbeta, c01 = gbn.bwd_coeffs(st, 3, stats, ab, gamma, n, None, ctx.direct[i])
            dx = torch.empty_like(x)
            L.call("gdmae_rows_bwd", L.ptr(x), _bf(x), None, n, C, L.ptr(ab), L.ptr(ab[C:]), L.ptr(c01), L.ptr(c01[C:]), L.ptr(g), _bf(g),
                   Ct, col, L.ptr(dx), _bf(dx), L.stream())
            grads += [dx, dgamma, dbeta]
            col += C
        return tuple(grads)


class _BnChannels:
    """The slice [c0, c1) of a BatchNorm module's running statistics, as the fused layers read a module (PointLayer2Max)."""

    def __init__(self, bn, c0, c1, count_batch):
        self.training, self.momentum = bn.training, bn.momentum
        self.running_mean = None if bn.running_mean is None else bn.running_mean[c0:c1]
        self.running_var = None if bn.running_var is None else bn.running_var[c0:c1]
        # num_batches_tracked counts batches, not channel blocks: only the first block increments the module's counter
        self.num_batches_tracked = bn.num_batches_tracked if count_batch else torch.zeros_like(bn.num_batches_tracked)


def point_layer2_max(y1, row_pillar, weight, bn, pt_off):
    """PointLayer2Max for 64 -> 128 and - config E's DynVFE (64 -> 256, gd_mae ONCE config) - for wider layers as independent
    128-channel blocks: Linear without bias, BatchNorm and the pillar maximum are all per OUTPUT channel, so block b of the result is
    the fused layer on rows [128 b, 128 b + 128) of the weight / BatchNorm vectors (the recompute-fused kernels of csrc/vfe_layer2.hip
    are written for 128 outputs); the input gradients of the blocks meet in one pass (ops.FanOut).  -> (M, C) fp32."""
    C = weight.shape[0]
    if C == 128:
        return PointLayer2Max.apply(y1, row_pillar, weight, bn.weight, bn.bias, bn.eps, pt_off, bn)[0]
    assert C % 128 == 0
    nb = C // 128
    ys = ops.FanOut.apply(y1, nb) if (y1.requires_grad and torch.is_grad_enabled()) else (y1,) * nb
    outs = []
    for b in range(nb):
        c0, c1 = 128 * b, 128 * (b + 1)
        outs.append(PointLayer2Max.apply(ys[b], row_pillar, weight[c0:c1], bn.weight[c0:c1], bn.bias[c0:c1], bn.eps, pt_off,
                                         _BnChannels(bn, c0, c1, b == 0))[0])
    return torch.cat(outs, dim=1)


class BNReLUSegmentMax(torch.autograd.Function):
    """max over each pillar's points of relu(BatchNorm1d_train(x)); returns (out (M, C) fp32, mean, biased var)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, pt_off, pillar_pts, inverse32, bn=None):
        x = x.contiguous()
        n, C = x.shape
        M = pt_off.numel() - 1
        stats, ab, mv = gbn.fold(x, n, gamma, beta, eps, bn)
        out = torch.empty(M, C, dtype=torch.float32, device=x.device)
        arg = torch.empty(M, C, dtype=torch.int32, device=x.device)
        L.call("gdmae_segment_max_affine", L.ptr(x), _bf(x), L.ptr(pt_off), L.ptr(pillar_pts), M, C, L.ptr(ab), L.ptr(ab[C:]),
               L.ptr(out), L.ptr(arg), L.stream())
        ctx.save_for_backward(x, out, arg, inverse32, ab, stats, gamma.detach())
        ctx.direct = gbn.direct_pair(gamma, beta)
        mf, vf = mv[:C], mv[C:]
        ctx.mark_non_differentiable(mf, vf)
        ctx.set_materialize_grads(False)      # no zero tensors for the unused gradients of those outputs
        return out, mf, vf

    @staticmethod
    def backward(ctx, g, _m, _v):
        if g is None:
            return (None,) * 8
        x, out, arg, inv, ab, stats, gamma = ctx.saved_tensors
        n, C = x.shape
        M = out.shape[0]
        g = g.float().contiguous()
        st = torch.empty(2 * C, dtype=torch.float64, device=x.device)
        ws = torch.empty(L.load().gdmae_rows_bwd_stats_workspace_bytes(C), dtype=torch.uint8, device=x.device)
        L.call("gdmae_segmax_bwd_stats", L.ptr(x), _bf(x), L.ptr(out), L.ptr(arg), L.ptr(g), M, C, L.ptr(ab), L.ptr(ab[C:]),
               L.ptr(st), L.ptr(ws), L.stream())
        dgamma, dbeta, c01 = gbn.bwd_coeffs(st, 2, stats, ab, gamma, n, None, ctx.direct)
        dx = torch.empty_like(x)
        L.call("gdmae_segmax_bn_bwd", L.ptr(x), _bf(x), L.ptr(out), L.ptr(arg), L.ptr(g), L.ptr(inv), n, C, L.ptr(ab), L.ptr(c01),
               L.ptr(c01[C:]), L.ptr(dx), _bf(dx), L.stream())
        return dx, dgamma, dbeta, None, None, None, None, None


class PointLayer1(torch.autograd.Function):
    """relu(BatchNorm1d_train(decorate(points) W^T)) for the first DynVFE layer (64 channels, Linear without bias) as
    gdmae_vfe_point_layer_fwd / _bwd: the decorated features and the (N, 64) pre-activation are recomputed from the
    points in registers wherever they are needed (csrc/vfe_fused.hip), the only (N, 64) tensors are the output and its
    gradient.  ``pillar_major``: row q of the output is point ``vox.pillar_pts[q]`` (the rows of a pillar contiguous,
    as PointLayer2Max wants them; inputs ``vox.points_pm`` / ``vox.row_pillar``) instead of point q.
    Returns (out [bf16 under autocast, else fp32], mean, biased var)."""

    @staticmethod
    def forward(ctx, vox, weight, gamma, beta, eps, bn=None, pillar_major=False, out_f16=False):
        dev = weight.device
        C, D = weight.shape
        assert weight.dtype == torch.float32 and weight.is_contiguous() and D == vox.n_cols + 5
        odt = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
        if out_f16:                      # (only inside PointLayers12Max: a float16 output of its own would demand float16 gradients)
            odt = torch.float16
        n = int(vox.N)
        out = torch.empty(n, C, dtype=odt, device=dev)
        stats = torch.empty(2 * C, dtype=torch.float64, device=dev)
        ab = torch.empty(2 * C, dtype=torch.float32, device=dev)
        mv = torch.empty(2 * C, dtype=torch.float32, device=dev)
        ws = torch.empty(L.load().gdmae_vfe_point_layer_workspace_bytes(vox.n_cols), dtype=torch.uint8, device=dev)
        ctx.pm = bool(pillar_major)
        rm = rv = nb = None
        mom = 0.0
        if bn is not None and bn.training and bn.running_mean is not None:
            rm, rv, nb, mom = bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum)
        L.call("gdmae_vfe_point_layer_fwd", *PointLayer1._geometry(vox, ctx.pm), L.ptr(weight), C, L.ptr(gamma), L.ptr(beta),
               float(eps), mom, L.ptr(rm) if rm is not None else None, L.ptr(rv) if rv is not None else None,
               L.ptr(nb) if nb is not None else None, L.ptr(stats), L.ptr(ab), L.ptr(mv), L.ptr(out), 2 if out_f16 else _bf(out), L.ptr(ws),
               L.stream())
        ctx.vox, ctx.ws = vox, ws
        ctx.save_for_backward(weight, gamma.detach(), stats, ab)
        ctx.direct = (ops.direct_grad(weight), *gbn.direct_pair(gamma, beta))
        mf, vf = mv[:C], mv[C:]
        ctx.mark_non_differentiable(mf, vf)
        ctx.set_materialize_grads(False)      # no zero tensors for the unused gradients of those outputs
        return out, mf, vf

    @staticmethod
    def _geometry(vox, pm):
        """leading arguments of gdmae_vfe_point_layer_*: per-point tables, or the pillar-major rows of the plan"""
        if pm:
            assert vox.points_pm is not None and vox.row_pillar is not None
            return (L.ptr(vox.points_pm), L.ptr(vox.voxel_coords), L.ptr(vox.row_pillar), L.ptr(vox.pillar_mean), 1,
                    int(vox.N), vox.n_cols, L.host_f32(vox.lo), L.host_f32(vox.vs))
        return (L.ptr(vox.points), L.ptr(vox.point_coords), L.ptr(vox.inverse32), L.ptr(vox.pillar_mean), 0, int(vox.N),
                vox.n_cols, L.host_f32(vox.lo), L.host_f32(vox.vs))

    @staticmethod
    def backward(ctx, g, _m, _v):
        if g is None:
            return (None,) * 8
        weight, gamma, stats, ab = ctx.saved_tensors
        vox, C = ctx.vox, weight.shape[0]
        g = g.contiguous()
        assert g.dtype in (torch.float32, torch.bfloat16)
        dw, dg, db = ctx.direct
        acc = int(dw is not None and dg is not None and db is not None)
        if not acc:
            dw = torch.empty_like(weight)
            dgb = torch.empty(2 * C, dtype=torch.float32, device=weight.device)
            dg, db = dgb[:C], dgb[C:]
        L.call("gdmae_vfe_point_layer_bwd", *PointLayer1._geometry(vox, ctx.pm), L.ptr(weight), C, L.ptr(gamma), L.ptr(stats),
               L.ptr(ab), L.ptr(g), _bf(g), L.ptr(dg), L.ptr(db), L.ptr(dw), acc, L.ptr(ctx.ws), L.stream())
        if acc:
            return None, None, None, None, None, None, None, None
        return None, dw, dg, db, None, None, None, None


class PointLayer2Max(torch.autograd.Function):
    """Per-pillar max of relu(BatchNorm1d_train(y1 W^T)) for the last DynVFE layer (64 -> 128, Linear without bias) in
    the bf16 throughput mode, as gdmae_vfe_max_layer_fwd / _bwd (csrc/vfe_layer2.hip): the (N, 128) pre-activation is
    recomputed from y1 in MFMA accumulators, never stored.  y1 (N, 64) bf16 with its rows in pillar-major order and
    row_pillar (N,) int32 = ``vox.row_pillar`` (PointLayer1(pillar_major=True)).  Returns (out (M, 128) fp32, mean, var)."""

    @staticmethod
    def forward(ctx, y1, row_pillar, weight, gamma, beta, eps, pt_off, bn=None):
        dev = y1.device
        C, K = weight.shape
        assert (C, K) == (128, 64) and y1.dtype in (torch.bfloat16, torch.float16) and y1.is_contiguous() and y1.shape[1] == K
        assert row_pillar.dtype == torch.int32 and row_pillar.numel() == y1.shape[0]
        n, M = y1.shape[0], pt_off.numel() - 1
        f16 = y1.dtype == torch.float16          # fp16 rows (inside PointLayers12Max): the kernels round the fp32 master weights themselves
        wb = weight.detach().contiguous() if f16 else ops.shadow(weight, torch.bfloat16).contiguous()
        assert not f16 or wb.dtype == torch.float32
        ctx.f16 = f16
        out = torch.empty(M, C, dtype=torch.float32, device=dev)
        arg = torch.empty(M, C, dtype=torch.int32, device=dev)
        stats = torch.empty(2 * C, dtype=torch.float64, device=dev)
        ab = torch.empty(2 * C, dtype=torch.float32, device=dev)
        mv = torch.empty(2 * C, dtype=torch.float32, device=dev)
        ws = torch.empty(L.load().gdmae_vfe_max_layer_workspace_bytes(), dtype=torch.uint8, device=dev)
        rm = rv = nb = None
        mom = 0.0
        if bn is not None and bn.training and bn.running_mean is not None:
            rm, rv, nb, mom = bn.running_mean, bn.running_var, bn.num_batches_tracked, float(bn.momentum)
        L.call("gdmae_vfe_max_layer_fwd_f16" if f16 else "gdmae_vfe_max_layer_fwd", L.ptr(y1), n, L.ptr(wb), L.ptr(pt_off), L.ptr(row_pillar), M, L.ptr(gamma),
               L.ptr(beta), float(eps), mom, L.ptr(rm) if rm is not None else None, L.ptr(rv) if rv is not None else None,
               L.ptr(nb) if nb is not None else None, L.ptr(stats), L.ptr(ab), L.ptr(mv), L.ptr(out), L.ptr(arg), L.ptr(ws),
               L.stream())
        ctx.ws = ws
        ctx.save_for_backward(y1, row_pillar, wb, gamma.detach(), stats, ab, out, arg)
        ctx.direct = (ops.direct_grad(weight), *gbn.direct_pair(gamma, beta))
        mf, vf = mv[:C], mv[C:]
        ctx.mark_non_differentiable(mf, vf)
        ctx.set_materialize_grads(False)      # no zero tensors for the unused gradients of those outputs
        return out, mf, vf

    @staticmethod
    def backward(ctx, g, _m, _v):
        if g is None:
            return (None,) * 8
        y1, row_pillar, wb, gamma, stats, ab, out, arg = ctx.saved_tensors
        n, (M, C) = y1.shape[0], out.shape
        g = g.float().contiguous()
        dw, dg, db = ctx.direct
        acc = int(dw is not None and dg is not None and db is not None)
        if not acc:
            dw = torch.empty(C, y1.shape[1], dtype=torch.float32, device=y1.device)
            dgb = torch.empty(2 * C, dtype=torch.float32, device=y1.device)
            dg, db = dgb[:C], dgb[C:]
        gm = torch.empty_like(out)
        dy1 = torch.empty(y1.shape, dtype=torch.bfloat16, device=y1.device)        # gradients stay bf16 (their range is why bf16 exists)
        L.call("gdmae_vfe_max_layer_bwd_f16" if ctx.f16 else "gdmae_vfe_max_layer_bwd", L.ptr(y1), n, L.ptr(wb), L.ptr(row_pillar), M, L.ptr(gamma), L.ptr(stats), L.ptr(ab),
               L.ptr(out), L.ptr(arg), L.ptr(g), L.ptr(gm), L.ptr(dy1), L.ptr(dg), L.ptr(db), L.ptr(dw), acc,
               L.ptr(ctx.ws), L.stream())
        if acc:
            return dy1, None, None, None, None, None, None, None
        return dy1, None, dw, dg, db, None, None, None



class _StandInCtx:
    """Autograd context of a Function whose forward / backward bodies run inside another Function (PointLayers12Max)."""

    def __init__(self):
        self.saved_tensors = ()
        self.needs_input_grad = ()

    def save_for_backward(self, *ts):
        self.saved_tensors = ts

    def mark_non_differentiable(self, *ts):
        pass

    def set_materialize_grads(self, flag):
        pass


class PointLayers12Max(torch.autograd.Function):
    """Both DynVFE layers of the 16-bit mode as ONE autograd node (dyn_vfe.py:74-112): PointLayer1 (pillar-major rows) and PointLayer2Max
    composed, so that the (N, 64) rows between them can be **fp16** - an fp16 tensor of its own between two nodes would demand an fp16
    gradient, while the gradient rows must stay bf16.  Round 6: the pillar maximum passes ONE point's value on, so the rounding of those
    rows and of the second layer's weights does not average over a pillar's points; with fp16 (11 significand bits, same bytes, same
    matrix-core rate) DynVFE's share of the small-case loss scatter goes away.  Layers wider than 128 outputs
    (config E: 64 -> 256) run as independent 128-channel blocks whose input gradients meet in one gdmae_sum_bf16 pass.
    forward(vox, W1, gamma1, beta1, eps1, bn1, W2, gamma2, beta2, eps2, bn2) -> (M, C2) fp32."""

    @staticmethod
    def forward(ctx, vox, w1, g1, b1, eps1, bn1, w2, g2, b2, eps2, bn2):
        c1 = _StandInCtx()
        y1, _, _ = PointLayer1.forward(c1, vox, w1, g1, b1, eps1, bn1, True, True)
        C2 = w2.shape[0]
        assert C2 % 128 == 0
        c2s, outs = [], []
        for blk in range(C2 // 128):
            lo, hi = 128 * blk, 128 * (blk + 1)
            c2 = _StandInCtx()
            if C2 == 128:          # the parameters themselves (not slices of them): their gradients go straight into the flat buffer
                o, _, _ = PointLayer2Max.forward(c2, y1, vox.row_pillar, w2, g2, b2, eps2, vox.pt_off, bn2)
            else:
                o, _, _ = PointLayer2Max.forward(c2, y1, vox.row_pillar, w2[lo:hi], g2[lo:hi], b2[lo:hi], eps2, vox.pt_off,
                                                 _BnChannels(bn2, lo, hi, blk == 0))
                # slices of flat-buffer parameters: the block's gradients accumulate into the matching rows of the flat views
                dwf, dgf, dbf = ops.direct_grad(w2), *gbn.direct_pair(g2, b2)
                if dwf is not None and dgf is not None and dbf is not None:
                    c2.direct = (dwf[lo:hi], dgf[lo:hi], dbf[lo:hi])
            c2s.append(c2), outs.append(o)
        # the tensors the bodies saved go through the real save_for_backward; the stand-ins keep the non-tensor metadata
        saved, counts = list(c1.saved_tensors), [len(c1.saved_tensors)]
        c1.saved_tensors = ()
        for c2 in c2s:
            saved += list(c2.saved_tensors)
            counts.append(len(c2.saved_tensors))
            c2.saved_tensors = ()
        ctx.save_for_backward(*saved)
        ctx.c1, ctx.c2s, ctx.counts = c1, c2s, counts
        ctx.set_materialize_grads(False)
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 11
        sv, pos = ctx.saved_tensors, 0
        c1, c2s = ctx.c1, ctx.c2s
        c1.saved_tensors = sv[pos:pos + ctx.counts[0]]
        pos += ctx.counts[0]
        dy1s, dw2, dg2, db2 = [], [], [], []
        acc2 = True
        for i, c2 in enumerate(c2s):
            c2.saved_tensors = sv[pos:pos + ctx.counts[i + 1]]
            pos += ctx.counts[i + 1]
            r = PointLayer2Max.backward(c2, g[:, 128 * i:128 * (i + 1)], None, None)
            c2.saved_tensors = ()
            dy1s.append(r[0])
            acc2 = acc2 and r[2] is None
            dw2.append(r[2]), dg2.append(r[3]), db2.append(r[4])
        if len(dy1s) == 1:
            dy1 = dy1s[0]
        else:
            dy1 = torch.empty_like(dy1s[0])
            L.call("gdmae_sum_bf16", L.host_ptrs(dy1s), len(dy1s), dy1.numel(), L.ptr(dy1), L.stream())
        r1 = PointLayer1.backward(c1, dy1, None, None)
        c1.saved_tensors = ()
        cat = (lambda ts: None if acc2 else torch.cat(ts, dim=0))
        return (None, r1[1], r1[2], r1[3], None, None, cat(dw2), cat(dg2), cat(db2), None, None)
