"""torch.autograd wrappers around the HIP entry points (the feature side of the hot path).

Token-wise GEMMs, the dense decoder convolutions and the norm layers are issued through PyTorch-ROCm
(hipBLASLt / MIOpen); everything that is gather/scatter/segment/attention/loss shaped runs in
libgdmae_hip.so.  There is no fallback path: without the library these ops raise.
"""
from __future__ import annotations

import torch

from . import lib as L
from . import timing

I32 = torch.int32


def _f32c(t):
    assert t.dtype == torch.float32
    return t.contiguous()


# ------------------------------------------------------------------------------------------------
# row gather / scatter
# ------------------------------------------------------------------------------------------------
def gather_rows_raw(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[s] = src[idx[s]] or 0 for idx < 0; idx any shape (int32), result idx.shape + (C,)."""
    src = src.contiguous()
    C = src.shape[-1]
    rb = C * src.element_size()
    out = torch.empty(tuple(idx.shape) + (C,), dtype=src.dtype, device=src.device)
    L.call("gdmae_gather_rows", L.ptr(src), L.ptr(idx.contiguous()), idx.numel(), rb, L.ptr(out), L.stream())
    return out


def scatter_rows_raw(src: torch.Tensor, idx: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    src = src.contiguous()
    rb = src.shape[-1] * src.element_size()
    L.call("gdmae_scatter_rows", L.ptr(src), L.ptr(idx.contiguous()), idx.numel(), rb, L.ptr(dst), L.stream())
    return dst


class ScatterToDense(torch.autograd.Function):
    """feat (n, C) -> zero-filled (R, C) with feat rows at idx (unique).  SparseConvTensor.dense()."""

    @staticmethod
    def forward(ctx, feat, idx, n_rows):
        ctx.save_for_backward(idx)
        dst = torch.zeros(n_rows, feat.shape[1], dtype=feat.dtype, device=feat.device)
        return scatter_rows_raw(feat, idx, dst)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return gather_rows_raw(g, idx), None, None


class GatherUnique(torch.autograd.Function):
    """dense (R, C) -> rows at idx (unique, >= 0).  Gather of decoder features at the pillar sites."""

    @staticmethod
    def forward(ctx, dense, idx):
        ctx.save_for_backward(idx)
        ctx.n_rows = dense.shape[0]
        return gather_rows_raw(dense, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        dst = torch.zeros(ctx.n_rows, g.shape[1], dtype=g.dtype, device=g.device)
        return scatter_rows_raw(g, idx, dst), None


# ------------------------------------------------------------------------------------------------
# token-wise linear layers: split-K weight gradients
# ------------------------------------------------------------------------------------------------
_BMM_OUT_DTYPE_OK = None
_GEMM_WS = {}


def _gemm_ws(dev, nbytes):
    t = _GEMM_WS.get(dev.index)
    if t is None or t.numel() < nbytes:
        t = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _GEMM_WS[dev.index] = t
    return t


def _gemm_ok(*ts):
    dt = ts[0].dtype
    return dt in (torch.bfloat16, torch.float32) and all(t.dtype == dt and t.is_contiguous() and t.dim() == 2 for t in ts)


def mm(a: torch.Tensor, b: torch.Tensor, trans_b: bool = False, bias: torch.Tensor | None = None) -> torch.Tensor:
    """a (M, K) @ b (K, N)   [or @ b^T for b (N, K) with trans_b]  (+ bias (N)), result in the operand dtype.
    Library GEMM through gdmae_gemm (hipBLASLt with one cached algorithm per shape bucket, ~10 us of host time per
    call instead of ~55 us through the framework's matmul); operands whose inner extents are not multiples of 8
    (the 11-feature DynVFE input) stay on the framework's GEMM."""
    M, K = a.shape
    N = b.shape[0] if trans_b else b.shape[1]
    if not (_gemm_ok(a, b) and K % 8 == 0 and N % 8 == 0 and M > 0 and (bias is None or (bias.dtype == a.dtype and bias.is_contiguous()))):
        y = a @ (b.t() if trans_b else b)
        return y if bias is None else y + bias
    out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    ws = _gemm_ws(a.device, L.load().gdmae_gemm_workspace_bytes())
    L.call("gdmae_gemm", L.ptr(a), L.ptr(b), L.ptr(out), M, N, K, 0, int(trans_b), int(a.dtype == torch.bfloat16), 0,
           None if bias is None else L.ptr(bias), L.ptr(ws), L.stream())
    return out


def splitk_tn_acc(a: torch.Tensor, b: torch.Tensor, dst: torch.Tensor) -> bool:
    """dst (m, n) fp32 += a^T @ b through gdmae_gemm_tn_splitk (accumulating reduce); False if the operands are not served by
    the library path (caller falls back to ``dst += splitk_tn(a, b)``)."""
    K, m = a.shape
    n = b.shape[1]
    if not (_gemm_ok(a, b) and m % 8 == 0 and n % 8 == 0 and K >= 1 and dst.is_contiguous() and dst.dtype == torch.float32
            and dst.numel() == m * n):
        return False
    ws = _gemm_ws(a.device, L.load().gdmae_gemm_tn_splitk_workspace_bytes(K, m, n))
    L.call("gdmae_gemm_tn_splitk", L.ptr(a), L.ptr(b), L.ptr(dst), K, m, n, int(a.dtype == torch.bfloat16), 1, L.ptr(ws), L.stream())
    return True


def splitk_tn(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a (K, m), b (K, n) -> a^T @ b (m, n) in fp32, with the long K (= tokens / points / sites) dimension
    split into slices.  Weight gradients of this model are (<=512 x <=2304) outputs with K = 20 k ... 1.4 M:
    as one GEMM they fill 16-72 output tiles, i.e. a few CUs of 256 (measured 130-1950 us each on MI355X);
    as S batched partial products + one fixed-order reduction every CU is busy (gdmae_gemm_tn_splitk)."""
    K, m = a.shape
    n = b.shape[1]
    if not (_gemm_ok(a, b) and m % 8 == 0 and n % 8 == 0 and K >= 1):
        return _splitk_tn_torch(a, b)
    out = torch.empty(m, n, dtype=torch.float32, device=a.device)
    ws = _gemm_ws(a.device, L.load().gdmae_gemm_tn_splitk_workspace_bytes(K, m, n))
    L.call("gdmae_gemm_tn_splitk", L.ptr(a), L.ptr(b), L.ptr(out), K, m, n, int(a.dtype == torch.bfloat16), 0, L.ptr(ws), L.stream())
    return out


def _splitk_tn_torch(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Same contraction on the framework's batched GEMM (operands the library path does not take: odd widths)."""
    global _BMM_OUT_DTYPE_OK
    K, m = a.shape
    n = b.shape[1]
    tiles = max(1, (m + 127) // 128) * max(1, (n + 127) // 128)
    S = max(1, min(K // 256, (1024 + tiles - 1) // tiles, 256))
    if S <= 1:
        return (a.t() @ b).float()
    kc = K // S
    K0 = kc * S
    a3 = a[:K0].view(S, kc, m).transpose(1, 2)
    b3 = b[:K0].view(S, kc, n)
    out = None
    if a.dtype != torch.float32 and _BMM_OUT_DTYPE_OK is not False:
        try:
            out = torch.bmm(a3, b3, out_dtype=torch.float32).sum(0)
            _BMM_OUT_DTYPE_OK = True
        except (TypeError, RuntimeError):
            _BMM_OUT_DTYPE_OK = False
    if out is None:
        out = torch.bmm(a3, b3).sum(0, dtype=torch.float32)
    if K0 < K:
        out = out + (a[K0:].t() @ b[K0:]).float()
    return out


def shadow(t: torch.Tensor, dtype) -> torch.Tensor:
    """`t` in `dtype` for GEMM use.  Parameters registered with a flat optimizer (gdmae_hip.optim) carry a bf16
    shadow that is refreshed once per optimizer step by ONE cast of the whole flat buffer, instead of one tiny cast
    kernel per weight per use; anything else is cast on the fly."""
    if t.dtype == dtype:
        return t
    base = t._base if t._base is not None else t
    sh = getattr(base, "_gd_shadow", None)
    if sh is not None and dtype == torch.bfloat16 and sh[1] == base._version:
        s16 = sh[0]
        if t is base:
            return s16
        return s16.as_strided(t.shape, t.stride(), s16.storage_offset() + t.storage_offset() - base.storage_offset())
    return t.to(dtype)


def direct_grad(p):
    """Flat-gradient view of a parameter owned by gdmae_hip.optim.FlatAdamOneCycle (zeroed at the start of the step;
    hand-written backwards accumulate into it directly) or None -> return the gradient through autograd as usual."""
    g = getattr(p, "_gd_flat_grad", None)
    return g if (g is not None and p.grad is g) else None


def colsum_f32(x2d: torch.Tensor) -> torch.Tensor:
    """Column sums (fp32) of a contiguous (R, C) fp32/bf16 matrix via gdmae_colstats (deterministic, fp64 combine)."""
    R, C = x2d.shape
    if R == 0 or (C % (8 if x2d.dtype == torch.bfloat16 else 4)) or C > 1024:
        return x2d.sum(0, dtype=torch.float32)
    out = torch.empty(2 * C, dtype=torch.float64, device=x2d.device)
    ws = torch.empty(L.load().gdmae_colstats_workspace_bytes(C), dtype=torch.uint8, device=x2d.device)
    L.call("gdmae_colstats", L.ptr(x2d), R, C, int(x2d.dtype == torch.bfloat16), L.ptr(out), L.ptr(ws), L.stream())
    return out[:C].float()


class LinearSplitK(torch.autograd.Function):
    """y = x W^T + b with bf16 compute under autocast (fp32 otherwise) and a split-K weight gradient."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        cdt = torch.bfloat16 if torch.is_autocast_enabled() else x.dtype
        xc = x.to(cdt).contiguous()
        wc = shadow(weight, cdt)
        ctx.w_t = not wc.is_contiguous() and wc.t().is_contiguous()       # weight given as the transpose of a (k, m) matrix
        if ctx.w_t:
            wc = wc.t()
        elif not wc.is_contiguous():
            wc = wc.contiguous()
        y = mm(xc, wc, trans_b=not ctx.w_t, bias=None if bias is None else shadow(bias, cdt).contiguous())
        ctx.save_for_backward(xc, wc)
        ctx.has_bias = bias is not None
        ctx.w_dtype = weight.dtype
        # weight gradient accumulated straight into the flat optimizer buffer when the parameter is given as stored
        ctx.direct_w = direct_grad(weight) if (not ctx.w_t and weight.is_contiguous()) else None
        ctx.direct_b = direct_grad(bias) if bias is not None else None
        return y

    @staticmethod
    def backward(ctx, g):
        xc, wc = ctx.saved_tensors
        g = g.contiguous().to(xc.dtype)
        dx = mm(g, wc, trans_b=ctx.w_t) if ctx.needs_input_grad[0] else None
        dw = None
        if not (ctx.direct_w is not None and splitk_tn_acc(g, xc, ctx.direct_w)):
            dw = splitk_tn(g, xc).to(ctx.w_dtype)
        db = None
        if ctx.has_bias:
            db = colsum_f32(g)
            if ctx.direct_b is not None:
                ctx.direct_b.add_(db)
                db = None
        return dx, dw, db


class FanOut(torch.autograd.Function):
    """``k`` handles on one tensor for ``k`` consumers whose gradients meet again in ONE pass: the autograd engine adds the gradients of a
    multiply-used tensor pairwise (k - 1 read-read-write passes, a bf16 rounding after each); here the backward is one
    ``gdmae_sum_bf16`` launch (k reads, one write, fp32 accumulation) when all k gradients are bf16 tensors of one dense layout, the
    engine's own sum otherwise.  Used where the reference hands one dense map to several branches (SeparateHead, center_head.py:37-45)."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k = int(k)
        ctx.set_materialize_grads(False)         # a handle nobody consumed arrives as None, not as a zero map in another layout
        return tuple(x.view_as(x) for _ in range(ctx.k))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        g0 = gs[0]
        same = all(g.is_cuda and g.dtype == torch.bfloat16 and g.shape == g0.shape and g.stride() == g0.stride() for g in gs)
        dense = same and g0.numel() % 8 == 0 and (g0.is_contiguous() or g0.is_contiguous(memory_format=torch.channels_last))
        if not dense or len(gs) > 8:
            out = gs[0]
            for g in gs[1:]:
                out = out + g
            return out, None
        out = torch.empty_like(g0)                   # preserves the (dense) strides
        L.call("gdmae_sum_bf16", L.host_ptrs_any(gs), len(gs), g0.numel(), L.ptr_any(out), L.stream())
        return out, None


class ResidualAdd(torch.autograd.Function):
    """a (fp32) + b (fp32 or bf16) as one gdmae_add3_to launch: the block residual ``feat + out`` of SSTBlockV1
    (spt_backbone.py:158), where ``feat`` is the bf16 output of the sparse-conv block under autocast.  Under autocast the sum is
    written in bf16: its only consumer is the next sparse convolution, which rounds its input rows to bf16 anyway (same
    arithmetic), and each of those rows is gathered up to nine times there - and nine times more by the weight gradient."""

    @staticmethod
    def forward(ctx, a, b):
        assert a.dtype == torch.float32 and b.dtype in (torch.float32, torch.bfloat16) and a.shape == b.shape
        ctx.b_dtype = b.dtype
        a, b = a.contiguous(), b.contiguous()
        odt = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
        out = torch.empty_like(a, dtype=odt)
        L.call("gdmae_add3_to", L.ptr(a), L.ptr(b), int(b.dtype == torch.bfloat16), None, 0, a.numel(), L.ptr(out),
               int(odt == torch.bfloat16), L.stream())
        return out

    @staticmethod
    def backward(ctx, g):
        ga = g if g.dtype == torch.float32 else g.float()
        return ga, (ga if ctx.b_dtype == torch.float32 else (g if g.dtype == ctx.b_dtype else g.to(ctx.b_dtype)))


def linear(x, weight, bias=None):
    return LinearSplitK.apply(x, weight, bias)


# ------------------------------------------------------------------------------------------------
# 2-D sparse convolution = rulebook gather (HIP) + one GEMM (hipBLASLt)
# ------------------------------------------------------------------------------------------------
class SparseConv3x3(torch.autograd.Function):
    """out[o] = sum_k W[:, ky, kx, :] . x[nbr[o, k]];  W in spconv-2.x layout (Cout, 3, 3, Cin).

    nbr (n_out, 9) maps output sites to input rows; nbr_t (n_in, 9) maps input rows to output rows
    (same tap index), used to express the input gradient as a gather as well (no atomics).
    """

    @staticmethod
    def forward(ctx, x, weight, nbr, nbr_t):
        cout, _, _, cin = weight.shape
        cdt = torch.bfloat16 if torch.is_autocast_enabled() else x.dtype      # bf16 throughput mode under autocast
        cols = gather_rows_raw(x.to(cdt), nbr).view(nbr.shape[0], 9 * cin)
        wc = shadow(weight, cdt)
        ctx.save_for_backward(cols, wc, nbr_t)
        ctx.w_dtype = weight.dtype
        return mm(cols, wc.reshape(cout, 9 * cin), trans_b=True)

    @staticmethod
    def backward(ctx, g):
        cols, wc, nbr_t = ctx.saved_tensors
        cout, _, _, cin = wc.shape
        g = g.contiguous().to(cols.dtype)
        dw = splitk_tn(g, cols).view(cout, 3, 3, cin).to(ctx.w_dtype)
        gcols = gather_rows_raw(g, nbr_t).view(nbr_t.shape[0], 9 * cout)
        wt = wc.permute(1, 2, 0, 3).reshape(9 * cout, cin)
        return mm(gcols, wt), dw, None, None


# ------------------------------------------------------------------------------------------------
# segmented max over the pillar CSR
# ------------------------------------------------------------------------------------------------
class SegmentMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pt_off, pillar_pts, inverse32):
        x = _f32c(x)
        M = pt_off.numel() - 1
        C = x.shape[1]
        out = torch.empty(M, C, dtype=torch.float32, device=x.device)
        arg = torch.empty(M, C, dtype=I32, device=x.device)
        L.call("gdmae_segment_max", L.ptr(x), L.ptr(pt_off), L.ptr(pillar_pts), M, C, L.ptr(out), L.ptr(arg), L.stream())
        ctx.save_for_backward(arg, inverse32)
        ctx.shape = x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        arg, inv = ctx.saved_tensors
        N, C = ctx.shape
        dx = torch.empty(N, C, dtype=torch.float32, device=g.device)
        L.call("gdmae_segment_max_bwd", L.ptr(_f32c(g)), L.ptr(arg), L.ptr(inv), N, C, L.ptr(dx), L.stream())
        return dx, None, None, None


def decorate_points(vox) -> torch.Tensor:
    """(N, 6+F) decorated point features of DynVFE (no gradient: raw geometry)."""
    F = vox.n_cols - 1
    out = torch.empty(vox.N, F + 6, dtype=torch.float32, device=vox.points.device)
    L.call("gdmae_decorate_points", L.ptr(vox.points), L.ptr(vox.point_coords), L.ptr(vox.inverse32),
           L.ptr(vox.pillar_mean), vox.N, vox.n_cols, L.host_f32(vox.lo), L.host_f32(vox.vs), L.ptr(out), L.stream())
    return out


# ------------------------------------------------------------------------------------------------
# fused residual add + LayerNorm
# ------------------------------------------------------------------------------------------------
class AddLayerNorm(torch.autograd.Function):
    """LayerNorm(a + b) * gamma + beta, one HBM pass (a fp32, b fp32 or bf16)."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, eps):
        ctx.a_dtype = a.dtype
        a = a.float().contiguous()        # bf16 only for the first layer after a sparse conv under autocast
        b = b.contiguous()
        assert b.dtype in (torch.float32, torch.bfloat16) and a.shape == b.shape
        n, d = a.shape
        y = torch.empty_like(a)
        stats = torch.empty(n, 2, dtype=torch.float32, device=a.device)
        L.call("gdmae_add_layernorm_fwd", L.ptr(a), L.ptr(b), int(b.dtype == torch.bfloat16), L.ptr(gamma.detach().contiguous()),
               L.ptr(beta.detach().contiguous()), n, d, float(eps), L.ptr(y), L.ptr(stats), None, L.stream())
        ctx.save_for_backward(a, b, gamma, stats)
        return y

    @staticmethod
    def backward(ctx, g):
        a, b, gamma, stats = ctx.saved_tensors
        n, d = a.shape
        dx = torch.empty_like(a)
        dgb = torch.empty(3 * d, dtype=torch.float32, device=a.device)
        ws = torch.empty(L.load().gdmae_add_layernorm_workspace_bytes(d), dtype=torch.uint8, device=a.device)
        L.call("gdmae_add_layernorm_bwd", L.ptr(a), L.ptr(b), int(b.dtype == torch.bfloat16), L.ptr(gamma.detach().contiguous()),
               L.ptr(stats), L.ptr(_f32c(g)), None, 0, n, d, L.ptr(dx), None, L.ptr(dgb), L.ptr(ws), L.stream())
        return dx.to(ctx.a_dtype), dx.to(b.dtype), dgb[:d], dgb[d:2 * d], None


def add_layer_norm(a, b, norm: torch.nn.LayerNorm):
    return AddLayerNorm.apply(a, b, norm.weight, norm.bias, norm.eps)


# ------------------------------------------------------------------------------------------------
# windowed cosine attention
# ------------------------------------------------------------------------------------------------
class WindowCosineAttention(torch.autograd.Function):
    """qk (n, 2d), v (n, d) fp32 or bf16 (both the same) -> out (n, d) in that dtype; fp32 arithmetic inside."""

    @staticmethod
    def forward(ctx, qk, v, tau, wplan, nhead, tau_min):
        assert qk.dtype == v.dtype and qk.dtype in (torch.float32, torch.bfloat16)
        qk, v = qk.contiguous(), v.contiguous()
        bf = int(qk.dtype == torch.bfloat16)
        es = qk.element_size()
        n, d = v.shape
        out = torch.empty(n, d, dtype=v.dtype, device=v.device)
        tau_flat = tau.detach().reshape(1).float().contiguous()
        # all occupancy levels through the entry the layer executor uses (bf16 rows: one launch; it leaves the rows' log-sum-exp for
        # the backward); algorithmic bytes: q,k,v rows read + out row written per token, + CSR (4 B/token + 8 B/window)
        nl = len(wplan.n_win)
        # does this call leave log-sum-exp rows?  Decided HERE, by the forward's own path: the backward must not consult the (switchable)
        # implementation flag again, it would read rows that were never written after a switch between forward and backward
        has_lse = bool(L.load().gdmae_window_attention_levels_writes_lse(bf, nl, L.host_i32(wplan.max_tokens), d, nhead))
        lse = torch.empty(n, nhead, dtype=torch.float32, device=v.device) if has_lse else None
        with timing.kernel("k_win_attn_fwd", n * (4 * d * es + 4) + 8 * sum(wplan.n_win)):
            L.call("gdmae_window_attention_levels_fwd", L.ptr(qk), L.ptr(v), L.ptr(out), bf, L.ptr(wplan.csr_tok), L.ptr(wplan.win_start),
                   L.ptr(wplan.win_len), nl, L.host_i32(wplan.n_win), L.host_i32(wplan.max_tokens), d, nhead, L.ptr(tau_flat), float(tau_min),
                   L.ptr(lse), L.stream())
        ctx.has_lse = has_lse
        # the output is saved through save_for_backward (allowed for outputs): an in-place edit of it before backward raises
        if has_lse:
            ctx.save_for_backward(qk, v, tau_flat, out, lse)
        else:
            ctx.save_for_backward(qk, v, tau_flat)
        ctx.wplan, ctx.nhead, ctx.tau_min, ctx.tau_shape, ctx.tau_dtype = wplan, nhead, tau_min, tau.shape, tau.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.has_lse:
            qk, v, tau_flat, out, lse = ctx.saved_tensors
        else:
            (qk, v, tau_flat), out, lse = ctx.saved_tensors, None, None
        wplan, H = ctx.wplan, ctx.nhead
        n, d = v.shape
        g = g.contiguous().to(v.dtype)
        bf = int(v.dtype == torch.bfloat16)
        es = v.element_size()
        dqk = torch.empty_like(qk)
        dv = torch.empty_like(v)
        n_items = [nw * H for nw in wplan.n_win]      # one partial per (window, head) at most
        part = torch.zeros(max(sum(n_items), 1), dtype=torch.float32, device=v.device)   # VALU levels fill fewer slots
        pbase = sum(n_items)
        # algorithmic bytes: q,k,v,dout rows read + dq,dk,dv rows written per token (7 d elements), + CSR
        with timing.kernel("k_win_attn_bwd", n * (7 * d * es + 4) + 8 * sum(wplan.n_win)):
            L.call("gdmae_window_attention_levels_bwd", L.ptr(qk), L.ptr(v), L.ptr(g), L.ptr(dqk), L.ptr(dv), bf, L.ptr(part), L.ptr(wplan.csr_tok),
                   L.ptr(wplan.win_start), L.ptr(wplan.win_len), len(wplan.n_win), L.host_i32(wplan.n_win), L.host_i32(wplan.max_tokens), d, H,
                   L.ptr(tau_flat), float(ctx.tau_min), L.ptr(out), L.ptr(lse), L.stream())
        dtau = torch.empty(1, dtype=torch.float32, device=v.device)
        # d clamp(tau, min)/d tau = 1 where tau >= min (torch.clamp backward), folded into the partial sum
        L.call("gdmae_sum_partials_gated", L.ptr(part), pbase, 1.0, L.ptr(dtau), L.ptr(tau_flat), float(ctx.tau_min),
               L.stream())
        return dqk, dv, dtau.view(ctx.tau_shape).to(ctx.tau_dtype), None, None, None


# ------------------------------------------------------------------------------------------------
# targets + Chamfer
# ------------------------------------------------------------------------------------------------
def group_gt_points(vox, K: int, want_index: bool = False):
    """(M, K, 3) ground-truth points relative to the pillar centre (+ optional (M, K) point ids)."""
    dev = vox.points.device
    gt = torch.empty(vox.M, K, 3, dtype=torch.float32, device=dev)
    gi = torch.empty(vox.M, K, dtype=I32, device=dev) if want_index else None
    L.call("gdmae_group_gt_points", L.ptr(vox.points), vox.n_cols, L.ptr(vox.pt_off), L.ptr(vox.pillar_pts),
           L.ptr(vox.voxel_coords), vox.M, K, L.host_f32(vox.lo), L.host_f32(vox.vs), L.ptr(gt), L.ptr(gi), L.stream())
    return (gt, gi) if want_index else gt


LAZY_SCALE = {}      # data_ptr of an UNSCALED fp32 Chamfer gradient -> (upstream gradient, 1 / sum of weights), consumed by the
                     # prediction head's backward (gdmae_hip.decoder.PredHeadFn), which applies the scale on load


class ChamferLoss(torch.autograd.Function):
    """pytorch3d-style weighted Chamfer distance; gradient w.r.t. pred only (gt carries none).

    ``lazy_scale``: pred (fp32) comes straight from the fused prediction head (decoder.pred_head): the backward hands over the
    unscaled per-point gradient with the two scalars on the side (LAZY_SCALE) - the head's input-gradient launch multiplies on
    load, so no pass over the (M, 16, 3) gradient runs here."""

    @staticmethod
    def forward(ctx, pred, gt, weights, lazy_scale=False):
        ctx.pred_dtype = pred.dtype       # bf16 rows of the prediction head under autocast: gradient returned in bf16
        ctx.lazy = bool(lazy_scale and pred.dtype == torch.float32)
        pred, gt, weights = _f32c(pred.float()), _f32c(gt), _f32c(weights)
        M, P1, _ = pred.shape
        P2 = gt.shape[1]
        term = torch.empty(max(M, 1), dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred)
        L.call("gdmae_chamfer", L.ptr(pred), L.ptr(gt), L.ptr(weights), M, P1, P2, L.ptr(term), L.ptr(dpred), L.stream())
        res = torch.empty(2, dtype=torch.float32, device=pred.device)       # {loss, 1 / sum of weights}
        L.call("gdmae_weighted_mean_finish", L.ptr(term), L.ptr(weights), M, L.ptr(res), L.stream())
        ctx.save_for_backward(dpred, res)
        return res[0]

    @staticmethod
    def backward(ctx, g):
        dpred, res = ctx.saved_tensors
        inv = res[1]
        if ctx.lazy:
            LAZY_SCALE.clear()                   # at most one pending hand-over
            LAZY_SCALE[dpred.data_ptr()] = (g.detach().reshape(1).float().contiguous(), res[1:2])
            return dpred, None, None, None
        if ctx.pred_dtype == torch.float32:
            return dpred * (g * inv), None, None, None
        out = torch.empty_like(dpred, dtype=ctx.pred_dtype)
        torch.mul(dpred, g * inv, out=out)       # scale and cast in one pass
        return out, None, None, None
