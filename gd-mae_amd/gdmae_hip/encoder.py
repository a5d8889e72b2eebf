"""Hand-written forward/backward of one post-norm encoder layer.

Same arithmetic as ``EncoderLayer.forward`` of the reference (pcdet/models/model_utils/sst_basic_block.py:77-84 with
``WindowAttention`` :22-54 and the cosine attention of cosine_msa.py): q = k = x + pos, v = x, packed in-projection
used as separate q/k and v slices, windowed cosine attention, out-projection, LN(x + attn), FFN with GELU(erf),
LN(x + ffn).  What differs is only the schedule: at 20-40 k tokens per stage every kernel of the layer runs for
5-30 us, so the step is bound by launch granularity.  Going through autograd op by op costs ~140 launches per layer
(dtype casts, bias-gradient reductions, gradient accumulation adds, split-K tails); this Function issues ~55:

* one ``gdmae_prep_tokens`` writes x and x + pos in the GEMM dtype (instead of an add and two casts);
* the fused add+LayerNorm kernels emit the bf16 copy the next GEMM needs, accept the second gradient branch on load
  (no accumulation kernels) and return the column sums that are the bias gradient of the preceding GEMM;
* activations that are operands of a weight-gradient GEMM live in buffers padded to a multiple of 2048 rows (pad
  rows zero), so dW = g^T x is ONE batched GEMM over K-chunks + one reduction, without a tail GEMM;
* the residual-stream gradient is assembled by one ``gdmae_add3``.
GEMMs are hipBLASLt through torch (bf16 under autocast, fp32 otherwise); attention, LayerNorm, prep/add kernels are
libgdmae_hip.so.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F

from . import lib as L
from . import ops, packing, timing

PAD = 2048


def _bf(t):
    return int(t.dtype == torch.bfloat16)


def _padded(n, cols, dtype, dev):
    n_pad = (n + PAD - 1) // PAD * PAD
    t = torch.empty(n_pad, cols, dtype=dtype, device=dev)
    if n_pad > n:
        t[n:].zero_()
    return t


def _dw(g_full: torch.Tensor, x_any: torch.Tensor, n: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """g^T x (fp32) with K = tokens split over a batched GEMM; both operands padded (zero rows) -> no tail GEMM."""
    if x_any.shape[0] != g_full.shape[0]:
        r = ops.splitk_tn(g_full[:n], x_any[:n])
        if out is not None:
            out.copy_(r)
            return out
        return r
    K, m = g_full.shape
    nn = x_any.shape[1]
    tiles = max(1, (m + 127) // 128) * max(1, (nn + 127) // 128)
    S = 1
    while S * 2 <= min(K // 256, max(1, 1024 // tiles), 256) and K % (S * 2) == 0:
        S *= 2
    a3 = g_full.view(S, K // S, m).transpose(1, 2)
    b3 = x_any.view(S, K // S, nn)
    if g_full.dtype != torch.float32 and ops._BMM_OUT_DTYPE_OK is not False:
        try:
            part = torch.bmm(a3, b3, out_dtype=torch.float32)
            ops._BMM_OUT_DTYPE_OK = True
        except (TypeError, RuntimeError):
            ops._BMM_OUT_DTYPE_OK = False
            part = torch.bmm(a3, b3).float()
    else:
        part = torch.bmm(a3, b3)
        if part.dtype != torch.float32:
            part = part.float()
    if out is not None:
        return torch.sum(part, 0, out=out)
    return part.sum(0)


def _ln_fwd(a, b, gamma, beta, eps, want_copy_dtype):
    n, d = a.shape
    y = torch.empty_like(a)
    stats = torch.empty(n, 2, dtype=torch.float32, device=a.device)
    yb = _padded(n, d, torch.bfloat16, a.device) if want_copy_dtype == torch.bfloat16 else None
    L.call("gdmae_add_layernorm_fwd", L.ptr(a), L.ptr(b), _bf(b), L.ptr(gamma), L.ptr(beta), n, d, float(eps), L.ptr(y),
           L.ptr(stats), None if yb is None else L.ptr(yb), L.stream())
    return y, stats, yb


def _ln_bwd(a, b, gamma, stats, dy, dy2, cdt):
    n, d = a.shape
    dx = torch.empty_like(a)
    dxb = _padded(n, d, torch.bfloat16, a.device) if cdt == torch.bfloat16 else None
    sums = torch.empty(3 * d, dtype=torch.float32, device=a.device)
    ws = torch.empty(L.load().gdmae_add_layernorm_workspace_bytes(d), dtype=torch.uint8, device=a.device)
    L.call("gdmae_add_layernorm_bwd", L.ptr(a), L.ptr(b), _bf(b), L.ptr(gamma), L.ptr(stats), L.ptr(dy),
           None if dy2 is None else L.ptr(dy2), 0 if dy2 is None else _bf(dy2), n, d, L.ptr(dx),
           None if dxb is None else L.ptr(dxb), L.ptr(sums), L.ptr(ws), L.stream())
    return dx, dxb, sums


def attn_forward_into(qk, v, out, tau_flat, wplan, nhead, tau_min):
    bf, es, d = _bf(qk), qk.element_size(), v.shape[1]
    base = 0
    for lvl, nw in enumerate(wplan.n_win):
        if nw > 0:
            with timing.kernel("k_win_attn_fwd", wplan.n_tok[lvl] * (4 * d * es + 4) + 8 * nw):
                L.call("gdmae_window_attention_fwd", L.ptr(qk), L.ptr(v), L.ptr(out), bf, L.ptr(wplan.csr_tok),
                       L.ptr(wplan.win_start[base:]), L.ptr(wplan.win_len[base:]), nw, wplan.max_tokens[lvl], d, nhead,
                       L.ptr(tau_flat), float(tau_min), L.stream())
        base += nw


def attn_backward_into(qk, v, g, dqk, dv, tau_flat, wplan, nhead, tau_min):
    bf, es, d = _bf(qk), qk.element_size(), v.shape[1]
    n_items = [nw * nhead for nw in wplan.n_win]
    part = torch.zeros(max(sum(n_items), 1), dtype=torch.float32, device=v.device)
    base, pbase = 0, 0
    for lvl, nw in enumerate(wplan.n_win):
        if nw > 0:
            with timing.kernel("k_win_attn_bwd", wplan.n_tok[lvl] * (7 * d * es + 4) + 8 * nw):
                L.call("gdmae_window_attention_bwd", L.ptr(qk), L.ptr(v), L.ptr(g), L.ptr(dqk), L.ptr(dv), bf,
                       L.ptr(part[pbase:]), L.ptr(wplan.csr_tok), L.ptr(wplan.win_start[base:]),
                       L.ptr(wplan.win_len[base:]), nw, wplan.max_tokens[lvl], d, nhead, L.ptr(tau_flat), float(tau_min),
                       L.stream())
        base += nw
        pbase += n_items[lvl]
    dtau = torch.empty(1, dtype=torch.float32, device=v.device)
    L.call("gdmae_sum_partials_gated", L.ptr(part), pbase, 1.0, L.ptr(dtau), L.ptr(tau_flat), float(tau_min), L.stream())
    return dtau


_direct = ops.direct_grad


class EncoderLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Win, bin_, tau, Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2, wplan, pos_table, nhead, tau_min, eps, act):
        ctx.direct = [_direct(p) for p in (Win, bin_, tau, Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2)]
        assert act == "gelu", "hot path uses ACTIVATION gelu (gd_mae_ssl.yaml:66)"
        x = x.float().contiguous()
        n, d = x.shape
        dev = x.device
        cdt = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
        sh = lambda t: ops.shadow(t, cdt)   # noqa: E731
        # q/k and v inputs in the GEMM dtype
        if cdt == torch.bfloat16:
            xb, xpb = _padded(n, d, cdt, dev), _padded(n, d, cdt, dev)
            L.call("gdmae_prep_tokens", L.ptr(x), L.ptr(pos_table), L.ptr(wplan.tok_pos), n, d, L.ptr(xb), L.ptr(xpb), 1, L.stream())
        else:
            xb, xpb = x, _padded(n, d, cdt, dev)
            L.call("gdmae_prep_tokens", L.ptr(x), L.ptr(pos_table), L.ptr(wplan.tok_pos), n, d, None, L.ptr(xpb), 0, L.stream())
        qk = F.linear(xpb[:n], sh(Win[:2 * d]), sh(bin_[:2 * d]))
        v = F.linear(xb[:n], sh(Win[2 * d:]), sh(bin_[2 * d:]))
        tau_flat = tau.detach().reshape(1).float().contiguous()
        o = _padded(n, d, cdt, dev)
        attn_forward_into(qk, v, o, tau_flat, wplan, nhead, tau_min)
        a = F.linear(o[:n], sh(Wo), sh(bo))
        x1, st1, x1b = _ln_fwd(x, a, g1.detach(), be1.detach(), eps, cdt)
        x1g = x1b if x1b is not None else x1
        h = F.linear(x1g[:n], sh(W1), sh(b1))
        gact = _padded(n, h.shape[1], cdt, dev)
        torch.ops.aten.gelu.out(h, out=gact[:n])
        f = F.linear(gact[:n], sh(W2), sh(b2))
        y, st2, _ = _ln_fwd(x1, f, g2.detach(), be2.detach(), eps, None)
        ctx.save_for_backward(x, xb, xpb, qk, v, o, a, x1, st1, x1g, h, gact, f, st2, tau_flat,
                              sh(Win), sh(Wo), sh(W1), sh(W2), g1.detach(), g2.detach())
        ctx.meta = (wplan, nhead, tau_min, cdt, n, d, tau.shape, tau.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x, xb, xpb, qk, v, o, a, x1, st1, x1g, h, gact, f, st2, tau_flat, Win_s, Wo_s, W1_s, W2_s, g1, g2) = ctx.saved_tensors
        wplan, nhead, tau_min, cdt, n, d, tau_shape, tau_dtype = ctx.meta
        dev = x.device
        dy = dy.float().contiguous()
        ff = h.shape[1]
        # ---- LN2 and FFN
        dx1_res, dfb, s2 = _ln_bwd(x1, f, g2, st2, dy, None, cdt)
        df = dfb if dfb is not None else dx1_res
        dW2 = _dw(df if dfb is not None else _as_padded(df, n), gact, n)
        dg = df[:n] @ W2_s
        dh = _padded(n, ff, cdt, dev)
        torch.ops.aten.gelu_backward.grad_input(dg, h, grad_input=dh[:n])
        db1 = ops.colsum_f32(dh[:n])
        dW1 = _dw(dh, x1g, n)
        dx1_b = dh[:n] @ W1_s
        # ---- LN1 (gradient = residual branch + FFN branch) and out-projection
        dx_res, dab, s1 = _ln_bwd(x, a, g1, st1, dx1_res, dx1_b, cdt)
        da = dab if dab is not None else dx_res
        dWo = _dw(da if dab is not None else _as_padded(da, n), o, n)
        do = da[:n] @ Wo_s
        # ---- attention
        dqk, dv = _padded(n, 2 * d, cdt, dev), _padded(n, d, cdt, dev)
        dtau = attn_backward_into(qk, v, do.contiguous(), dqk, dv, tau_flat, wplan, nhead, tau_min)
        dWin = torch.empty(3 * d, d, dtype=torch.float32, device=dev)
        _dw(dqk, xpb, n, out=dWin[:2 * d])
        _dw(dv, xb, n, out=dWin[2 * d:])
        dbin = torch.cat([ops.colsum_f32(dqk[:n]), ops.colsum_f32(dv[:n])])
        dx_qk = dqk[:n] @ Win_s[:2 * d]
        dx_v = dv[:n] @ Win_s[2 * d:]
        dx = torch.empty_like(x)
        L.call("gdmae_add3", L.ptr(dx_res), L.ptr(dx_qk), _bf(dx_qk), L.ptr(dx_v), _bf(dx_v), n * d, L.ptr(dx), L.stream())
        grads = [dWin, dbin, dtau.view(tau_shape).to(tau_dtype), dWo, s1[2 * d:], dW1, db1, dW2, s2[2 * d:],
                 s1[:d], s1[d:2 * d], s2[:d], s2[d:2 * d]]
        if all(t is not None for t in ctx.direct):
            # parameters live in a flat optimizer buffer: one fused copy per layer instead of 13 AccumulateGrad adds
            torch._foreach_add_(list(ctx.direct), [g_.view_as(t) for g_, t in zip(grads, ctx.direct)])
            grads = [None] * 13
        return (dx, *grads, None, None, None, None, None, None)


def _as_padded(t, n):
    """fp32 mode: copy an unpadded (n, C) gradient into a zero-padded buffer so _dw can use the no-tail path."""
    p = _padded(n, t.shape[1], t.dtype, t.device)
    p[:n].copy_(t[:n])
    return p


# ------------------------------------------------------------------------------------------------
# native executor: the whole layer forward / backward as ONE C-ABI call each (csrc/encoder_layer.hip)
# ------------------------------------------------------------------------------------------------
IMPL = "native"      # "native": gdmae_encoder_layer_fwd/bwd;  "python": EncoderLayerFn above (same arithmetic, op by op)

_GRAD_FIELDS = ("dWin", "dbin", "dtau", "dWo", "dbo", "dW1", "db1", "dW2", "db2", "dg1", "dbe1", "dg2", "dbe2")


import ctypes as _C

_PARAM_FIELDS = ("Win", "bin", "Wo", "bo", "W1", "b1", "W2", "b2", "g1", "be1", "g2", "be2", "tau")
_BYTES_CACHE = {}


def _layer_bytes(n, d, ff, nhead, bf):
    key = ((n + PAD - 1) // PAD, d, ff, nhead, bf)
    r = _BYTES_CACHE.get(key)
    if r is None:
        out = (_C.c_size_t * 3)()
        L.call("gdmae_encoder_layer_bytes", n, d, ff, nhead, bf, _C.byref(out, 0), _C.byref(out, _C.sizeof(_C.c_size_t)),
               _C.byref(out, 2 * _C.sizeof(_C.c_size_t)))
        r = _BYTES_CACHE[key] = (int(out[0]), int(out[1]), int(out[2]))
    return r


def _param_args(plist, cdt, direct):
    """LayerArgs pre-filled with parameter (and, with a flat optimizer, gradient) pointers.  With a flat optimizer the
    fp32 masters, their bf16 shadows and the gradient views are persistent buffers, so the struct is built once per
    layer; otherwise (plain autograd use) it is rebuilt on every call."""
    Win, bin_, tau, Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2 = plist
    def _persistent(t):      # fp32 master used as is, or a bf16 shadow that is current
        if cdt == t.dtype:
            return True
        base = t._base if t._base is not None else t
        sh = getattr(base, "_gd_shadow", None)
        return sh is not None and sh[1] == base._version
    stable = all(t is not None for t in direct) and all(_persistent(t) for t in (Win, bin_, Wo, bo, W1, b1, W2, b2))
    cache = getattr(Win, "_gd_layer_args", None)       # lives (and dies) with the layer's in-projection parameter
    versions = tuple(t._version for t in (Win, bin_, Wo, bo, W1, b1, W2, b2))   # every shadowed parameter, not only Win
    if stable and cache is not None:
        hit = cache.get(cdt)
        if hit is not None and hit[2] == versions and hit[3] == direct[0].data_ptr():
            return hit[0], hit[1]
    sh = lambda t: ops.shadow(t, cdt).contiguous()   # noqa: E731
    tensors = {"Win": sh(Win), "bin": sh(bin_), "Wo": sh(Wo), "bo": sh(bo), "W1": sh(W1), "b1": sh(b1), "W2": sh(W2), "b2": sh(b2),
               "g1": g1.detach(), "be1": be1.detach(), "g2": g2.detach(), "be2": be2.detach(),
               "tau": tau.detach().reshape(1).float().contiguous()}
    a = L.LayerArgs()
    for k, t in tensors.items():
        setattr(a, k, L.ptr(t))
    if cdt == torch.bfloat16 and TOKGEMM:
        # fragment-ordered weight image for the fused token GEMMs (refreshed by the optimizer for registered layers)
        packed = (packing.registered if stable else packing.pack_now)(Win, Wo, W1, W2)
        if packed is not None:
            tensors["packed"] = packed
            a.packed = L.ptr(packed)
    if all(t is not None for t in direct):
        # gradients accumulate straight into the flat optimizer buffer - also on a step whose shadows are stale (a torch-side weight
        # change between two optimizer steps: the operands are then cast / packed on the spot, nothing is cached)
        for k, g in zip(_GRAD_FIELDS, direct):
            setattr(a, k, L.ptr(g))
    if stable:
        # the shadows are refreshed in place by the optimizer; a changed version only matters for pointer identity
        if cache is None:
            cache = Win._gd_layer_args = {}
        cache[cdt] = (a, tensors, versions, direct[0].data_ptr())
    return a, tensors


def _call_args(base, x, wplan, pos_table, nhead, tau_min, eps, cdt, ff):
    a = L.LayerArgs.from_buffer_copy(base)
    n, d = x.shape
    a.n, a.d, a.ff, a.nhead, a.bf16 = n, d, ff, nhead, int(cdt == torch.bfloat16)
    a.eps, a.tau_min = float(eps), float(tau_min)
    nl = len(wplan.n_win)
    a.n_levels = nl
    for i in range(nl):
        a.n_win[i], a.max_tokens[i] = int(wplan.n_win[i]), int(wplan.max_tokens[i])
    a.tok_pos, a.csr_tok, a.win_start, a.win_len = (wplan.tok_pos.data_ptr(), wplan.csr_tok.data_ptr(), wplan.win_start.data_ptr(),
                                                    wplan.win_len.data_ptr())
    a.pos_table, a.x = pos_table.data_ptr(), x.data_ptr()
    return a


class EncoderLayerNativeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Win, bin_, tau, Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2, wplan, pos_table, nhead, tau_min, eps, act):
        assert act == "gelu", "hot path uses ACTIVATION gelu (gd_mae_ssl.yaml:66)"
        plist = (Win, bin_, tau, Wo, bo, W1, b1, W2, b2, g1, be1, g2, be2)
        direct = [_direct(p) for p in plist]
        x = x.float().contiguous()
        n, d = x.shape
        dev = x.device
        ff = W1.shape[0]
        cdt = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
        base, keep = _param_args(plist, cdt, direct)
        sb, fb, bb = _layer_bytes(n, d, ff, nhead, int(cdt == torch.bfloat16))
        saved = torch.empty(sb, dtype=torch.uint8, device=dev)
        scratch = torch.empty(fb, dtype=torch.uint8, device=dev)
        y = torch.empty_like(x)
        a = _call_args(base, x, wplan, pos_table, nhead, tau_min, eps, cdt, ff)
        a.y, a.saved, a.scratch = y.data_ptr(), saved.data_ptr(), scratch.data_ptr()
        L.call("gdmae_encoder_layer_fwd", _C.byref(a), L.stream())
        ctx.save_for_backward(x, saved, pos_table)
        ctx.meta = (wplan, nhead, tau_min, eps, cdt, ff, bb, base, keep, direct, [t.shape for t in plist], tau.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, saved, pos_table = ctx.saved_tensors
        wplan, nhead, tau_min, eps, cdt, ff, bb, base, keep, direct, shapes, tau_dtype = ctx.meta
        dev = x.device
        dy = dy.float().contiguous()
        dx = torch.empty_like(x)
        scratch = torch.empty(bb, dtype=torch.uint8, device=dev)
        a = _call_args(base, x, wplan, pos_table, nhead, tau_min, eps, cdt, ff)
        a.dy, a.dx, a.saved, a.scratch = dy.data_ptr(), dx.data_ptr(), saved.data_ptr(), scratch.data_ptr()
        if all(t is not None for t in direct):
            ret = [None] * 13                          # gradients accumulate straight into the flat optimizer buffer
        else:
            sizes = [int(torch.Size(s).numel()) for s in shapes]
            flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
            gl = list(torch.split(flat, sizes))
            for k, g in zip(_GRAD_FIELDS, gl):
                setattr(a, k, L.ptr(g))
            ret = [g.view(s) for g, s in zip(gl, shapes)]
            ret[2] = ret[2].to(tau_dtype)
        L.call("gdmae_encoder_layer_bwd", _C.byref(a), L.stream())
        return (dx, *ret, None, None, None, None, None, None)


# ------------------------------------------------------------------------------------------------
# stage executor: all layers of a stage (NUM_BLOCKS x 2) as ONE call per direction
# ------------------------------------------------------------------------------------------------
TOKGEMM = os.environ.get("GDMAE_TOKGEMM", "1") != "0"   # False: token GEMMs through hipBLASLt + separate row kernels (A/B)
FOLD_RESIDUAL = os.environ.get("GDMAE_FOLD_RES", "1") != "0"   # False: the block residual as its own add pass behind the stage (A/B)
STAGE = os.environ.get("GDMAE_STAGE", "1") != "0"     # False / GDMAE_STAGE=0: one call per layer (A/B reference)


def _plist(layer):
    sa = layer.win_attn.self_attn
    return (sa.in_proj_weight, sa.in_proj_bias, sa.tau, sa.out_proj.weight, sa.out_proj.bias, layer.linear1.weight,
            layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, layer.norm1.weight, layer.norm1.bias, layer.norm2.weight,
            layer.norm2.bias)


class EncoderStageFn(torch.autograd.Function):
    """x -> layer_L(... layer_1(x)) through gdmae_encoder_stage_fwd / _bwd.  Only used when every parameter lives in a
    flat optimizer buffer (gradients are accumulated there directly, so the parameters are not autograd inputs).

    ``residual``: return x + stage(x) - the block residual of SSTBlockV1 (spt_backbone.py:219-264).  On the fused path
    (csrc/layer_fused.hip) with bf16 rows the sum is written by the last layer's launch and the skip-path gradient is added by the
    first layer's backward launch (no cast / add passes around the stage); otherwise it is one ``ResidualAdd`` behind the stage."""

    @staticmethod
    def forward(ctx, x, info, residual):
        layers, wplans, pos_table, plists, directs = info
        n, d = x.shape
        dev = x.device
        nl = len(layers)
        l0 = layers[0]
        sa = l0.win_attn.self_attn
        nhead, tau_min, eps, ff = sa.num_heads, sa.tau_min, l0.norm1.eps, l0.linear1.weight.shape[0]
        cdt = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
        sb, fb, bb = _layer_bytes(n, d, ff, nhead, int(cdt == torch.bfloat16))
        arr = (L.LayerArgs * nl)()
        bases = []
        for i in range(nl):
            base, keep = _param_args(plists[i], cdt, directs[i])
            bases.append((base, keep))
            arr[i] = _call_args(base, x, wplans[i], pos_table, nhead, tau_min, eps, cdt, ff)
        saved = torch.empty(nl, sb, dtype=torch.uint8, device=dev)
        scratch = torch.empty(fb, dtype=torch.uint8, device=dev)
        for i in range(nl):
            arr[i].saved, arr[i].scratch = saved[i].data_ptr(), scratch.data_ptr()
        fused = bool(L.load().gdmae_encoder_stage_fused(arr, nl))
        folded = fused and residual and x.dtype == torch.bfloat16 and FOLD_RESIDUAL
        if fused:
            # only the last layer's output leaves the stage; x[i] == y[i-1] is the chaining contract of the entry point, the
            # intermediate rows live in the layers' saved blocks
            xin = x.contiguous() if folded else x.float().contiguous()
            out = torch.empty(n, d, dtype=torch.bfloat16 if folded else torch.float32, device=dev)
            for i in range(nl):
                arr[i].x = xin.data_ptr() if i == 0 else 1 + i     # never dereferenced for i > 0; distinct tokens keep the chain check
                arr[i].y = (2 + i) if i + 1 < nl else out.data_ptr()
            if folded:
                arr[0].x_bf16 = 1
                arr[nl - 1].res_out = out.data_ptr()
                arr[nl - 1].y = 0
            ys = None
        else:
            xin = x.float().contiguous()
            ys = torch.empty(nl, n, d, dtype=torch.float32, device=dev)
            for i in range(nl):
                arr[i].x = (xin if i == 0 else ys[i - 1]).data_ptr()
                arr[i].y = ys[i].data_ptr()
            out = ys[nl - 1]
        L.call("gdmae_encoder_stage_fwd", arr, nl, L.stream())
        ctx.save_for_backward(xin, saved, pos_table, *([] if ys is None else [ys]))
        ctx.meta = (wplans, nhead, tau_min, eps, cdt, ff, bb, bases, fused, folded, residual, x.dtype)
        if residual and not folded:
            odt = torch.bfloat16 if cdt == torch.bfloat16 else torch.float32
            res = torch.empty(n, d, dtype=odt, device=dev)
            xc = x.contiguous()
            L.call("gdmae_add3_to", L.ptr(out), L.ptr(xc), _bf(xc), None, 0, out.numel(), L.ptr(res), int(odt == torch.bfloat16), L.stream())
            return res
        return out

    @staticmethod
    def backward(ctx, dy):
        xin, saved, pos_table = ctx.saved_tensors[:3]
        ys = ctx.saved_tensors[3] if len(ctx.saved_tensors) > 3 else None
        wplans, nhead, tau_min, eps, cdt, ff, bb, bases, fused, folded, residual, x_dtype = ctx.meta
        nl = saved.shape[0]
        n, d = xin.shape
        scratch = torch.empty(bb, dtype=torch.uint8, device=xin.device)
        arr = (L.LayerArgs * nl)()
        for i in range(nl):
            a = _call_args(bases[i][0], xin, wplans[i], pos_table, nhead, tau_min, eps, cdt, ff)
            a.saved, a.scratch = saved[i].data_ptr(), scratch.data_ptr()
            if fused:
                a.x = xin.data_ptr() if i == 0 else 1 + i
                a.y = 2 + i
            else:
                a.x = (xin if i == 0 else ys[i - 1]).data_ptr()
                a.y = ys[i].data_ptr()
            arr[i] = a
        if folded:
            g = dy.contiguous() if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16).contiguous()
            dx = torch.empty(n, d, dtype=torch.bfloat16, device=xin.device)
            arr[0].x_bf16 = 1
            arr[nl - 1].dres, arr[0].dx_bf16 = g.data_ptr(), dx.data_ptr()
            L.call("gdmae_encoder_stage_bwd", arr, nl, L.stream())
            return dx, None, None
        g = dy.float().contiguous()
        dx = torch.empty(n, d, dtype=torch.float32, device=xin.device)
        arr[nl - 1].dy, arr[0].dx = g.data_ptr(), dx.data_ptr()
        L.call("gdmae_encoder_stage_bwd", arr, nl, L.stream())
        if residual:                 # the skip path carries dy unchanged
            if x_dtype == torch.float32:
                dx.add_(g)
            else:                    # same roundings as autograd's accumulation of the two branches in the input's dtype
                dx = dx.to(x_dtype) + (dy if dy.dtype == x_dtype else dy.to(x_dtype))
        return dx, None, None


def encoder_stage(blocks, x, pos_table, wplans, residual=False):
    """``blocks``: the stage's BasicShiftBlockV2 modules (layer k of a block uses window partition k % len(wplans)).
    ``residual``: return x + stage(x) (the block residual of SSTBlockV1), fp32 -> fp32, bf16 under autocast."""
    pairs = [(layer, wplans[k % len(wplans)]) for block in blocks for k, layer in enumerate(block.encoder_list)]
    layers = [p[0] for p in pairs]
    if (STAGE and IMPL == "native" and x.is_cuda and
            all(getattr(l, "fused", False) and l.activation_name == "gelu" for l in layers)):
        plists = [_plist(l) for l in layers]
        directs = [[_direct(p) for p in pl] for pl in plists]
        if all(t is not None for dl in directs for t in dl):
            return EncoderStageFn.apply(x, (layers, [p[1] for p in pairs], pos_table, plists, directs), residual)
    feat = x
    for layer, wp in pairs:
        x = layer(x, pos_table, wp)
    if residual:
        return ops.ResidualAdd.apply(x, feat) if (x.is_cuda and x.dtype == torch.float32) else feat + x
    return x


def encoder_layer(layer, x, wplan, pos_table):
    """``layer``: pcdet EncoderLayer module (parameter container); returns LN(x1 + FFN(x1)), x1 = LN(x + attn(x))."""
    sa = layer.win_attn.self_attn
    fn = EncoderLayerNativeFn if (IMPL == "native") else EncoderLayerFn
    return fn.apply(x, sa.in_proj_weight, sa.in_proj_bias, sa.tau, sa.out_proj.weight, sa.out_proj.bias,
                    layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias,
                    layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias,
                    wplan, pos_table, sa.num_heads, sa.tau_min, layer.norm1.eps, layer.activation_name)
