"""BatchNorm2d(train) + ReLU of DENSE channels-last maps through the row kernels of the hot path (fine-tune row f1).

The fine-tune detector (reference spt_backbone.py:282-303 deblocks / conv_out, sst_bev_backbone.py:18-40, center_head.py:20-45)
runs ``Conv2d -> BatchNorm2d -> ReLU`` on dense (B, C, Y, X) maps: 1.7 M sites x 128 channels per 8-frame KITTI-shape batch.  A
channels-last map IS a row-major (B Y X, C) matrix, i.e. exactly what the DynVFE / decoder row kernels take: one statistics pass
(gdmae_bn_fold: column sums in fp32 partials, fp64 combine, folded affine, running statistics) and one pass that applies
a x + b and the ReLU, against three MIOpen BatchNorm kernels + a ReLU pass; the backward is one statistics pass and one dx pass
(BatchNorm chain rule on column sums, ReLU mask recomputed from a x + b) against three MIOpen kernels + threshold_backward.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import lib as L
from . import ops
from .vfe import BNReLURows, BNReLURowsCat


# ------------------------------------------------------------------------------------------------
# Conv2d(k = 3, stride 1, padding = dilation) on channels-last bf16 maps: csrc/conv_dense.hip (forward, input gradient, weight gradient)
# ------------------------------------------------------------------------------------------------
def _pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def conv3x3_supported(conv: nn.Module, x: torch.Tensor) -> bool:
    """The library's dense convolution takes this layer on this input: 3 x 3, stride 1, padding = dilation in {1, 2}, one group, zero
    padding, input channels in multiples of 64 (the weight gradient's blocks); bf16 operands under autocast, the fp32-grade three-term
    form (Conv3x3DenseF32) for fp32 maps otherwise."""
    if not (isinstance(conv, nn.Conv2d) and x.is_cuda and x.dim() == 4 and (torch.is_autocast_enabled() or x.dtype == torch.float32)):
        return False
    d = conv.dilation[0]
    return (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.dilation == (d, d) and conv.padding == (d, d) and d in (1, 2)
            and conv.groups == 1 and conv.padding_mode == 'zeros' and conv.in_channels % 64 == 0 and conv.in_channels <= 1024
            and conv.out_channels <= 1024 and (d == 1 or conv.out_channels % 64 == 0) and conv.weight.dtype == torch.float32)


class Conv3x3Dense(torch.autograd.Function):
    """conv2d(x, weight, bias, padding = dil, dilation = dil) with bf16 operands / fp32 accumulation on channels-last maps (reference
    call sites: sst_bev_backbone.py:14-19, center_head.py:20-35, spt_backbone.py:289-291).  x: (B, Cin, H, W) in any layout / dtype
    (moved to channels-last bf16 if it is not there already); returns a (B, Cout, H, W) view of a channels-last bf16 map."""

    @staticmethod
    def forward(ctx, x, weight, bias, dil, direct, want_stats=False):
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        cl = _pad32(cout)
        xr = x.permute(0, 2, 3, 1)
        xr = xr.to(torch.bfloat16) if xr.dtype != torch.bfloat16 else xr
        xr = xr.contiguous()                                                   # (B, H, W, cin): no copy for a channels-last bf16 map
        w = weight.contiguous()
        packed = torch.empty(L.load().gdmae_conv3x3_dense_packed_bytes(cin, cout), dtype=torch.uint8, device=x.device)
        L.call("gdmae_conv3x3_dense_pack", L.ptr(w), cin, cout, dil, 0, L.ptr(packed), L.stream())
        bl = None
        if bias is not None:
            bl = torch.zeros(cl, dtype=torch.float32, device=x.device)
            bl[:cout] = bias.detach().float()
        y = torch.empty(B, H, W, cl, dtype=torch.bfloat16, device=x.device)
        ctx.save_for_backward(xr, w)
        ctx.meta = (dil, cout, cl, bias is not None, direct, x.dtype)
        if want_stats and cl == cout:
            # + the column sums / sums of squares of the rounded output as the convolution's epilogue: the BatchNorm that follows
            # folds from these rows (bn.fold partials) instead of reading the map again
            lib = L.load()
            part = torch.empty(lib.gdmae_conv3x3_dense_stat_rows(), 2, cl, dtype=torch.float32, device=x.device)
            ws = torch.empty(lib.gdmae_conv3x3_dense_stats_workspace_bytes(B, H, W, cl), dtype=torch.uint8, device=x.device)
            L.call("gdmae_conv3x3_dense_stats", L.ptr(xr), B, H, W, cin, cl, dil, L.ptr(packed), None if bl is None else L.ptr(bl), L.ptr(y),
                   L.ptr(part), L.ptr(ws), L.stream())
            ctx.mark_non_differentiable(part)
            ctx.set_materialize_grads(False)
            return y.permute(0, 3, 1, 2), part
        L.call("gdmae_conv3x3_dense", L.ptr(xr), B, H, W, cin, cl, dil, L.ptr(packed), None if bl is None else L.ptr(bl), L.ptr(y), L.stream())
        out = (y if cl == cout else y[..., :cout]).permute(0, 3, 1, 2)
        return (out, None) if want_stats else out

    @staticmethod
    def backward(ctx, dy, *_):
        xr, w = ctx.saved_tensors
        dil, cout, cl, has_bias, direct, x_dtype = ctx.meta
        B, H, W, cin = xr.shape
        dev = xr.device
        g = dy.permute(0, 2, 3, 1)
        if cl == cout:
            g = (g if g.dtype == torch.bfloat16 else g.to(torch.bfloat16)).contiguous()
        else:                                                                  # pad the gradient's channels to the launch's width
            gp = torch.zeros(B, H, W, cl, dtype=torch.bfloat16, device=dev)
            gp[..., :cout] = g
            g = gp
        dx = None
        if ctx.needs_input_grad[0]:
            packed = torch.empty(L.load().gdmae_conv3x3_dense_packed_bytes(cin, cout), dtype=torch.uint8, device=dev)
            L.call("gdmae_conv3x3_dense_pack", L.ptr(w), cin, cout, dil, 1, L.ptr(packed), L.stream())
            dxr = torch.empty(B, H, W, cin, dtype=torch.bfloat16, device=dev)
            addend = getattr(ctx, "dx_addend", None)       # ConvBNReLUShortcut: the shortcut's gradient joins dx in the store pass
            if addend is not None:
                assert addend.shape == dxr.shape and addend.dtype == torch.bfloat16 and addend.is_contiguous()
                L.call("gdmae_conv3x3_dense_add", L.ptr(g), B, H, W, cl, cin, dil, L.ptr(packed), None, L.ptr(addend), L.ptr(dxr), L.stream())
            else:
                L.call("gdmae_conv3x3_dense", L.ptr(g), B, H, W, cl, cin, dil, L.ptr(packed), None, L.ptr(dxr), L.stream())
            dx = dxr.permute(0, 3, 1, 2)
            if x_dtype != torch.bfloat16:
                dx = dx.to(x_dtype)
        dwd, dbd = direct if direct is not None else (None, None)
        dW = dwd if dwd is not None else torch.zeros(w.shape, dtype=torch.float32, device=dev)
        ws = torch.empty(L.load().gdmae_conv3x3_dense_dw_workspace_bytes(B, H, W, cin, cl), dtype=torch.uint8, device=dev)
        L.call("gdmae_conv3x3_dense_bwd_weight", L.ptr(xr), L.ptr(g), B, H, W, cin, cl, cin, cout, dil, L.ptr(dW), L.ptr(ws), L.stream())
        db = None
        if has_bias:
            db = ops.colsum_f32(g.view(B * H * W, cl))[:cout]
            if dbd is not None:
                dbd.add_(db)
                db = None
        return dx, (None if dwd is not None else dW), db, None, None, None


import os

FUSE_SHORTCUT = os.environ.get("GDMAE_DENSE_FUSE", "1") != "0"      # A/B switch (also read by SeparateHead's FanOut)


class _Ctx:
    """Stand-in for the autograd context when a Function's forward / backward bodies are composed inside another Function."""

    def __init__(self, needs_input_grad=()):
        self.needs_input_grad = needs_input_grad
        self.saved_tensors = ()

    def save_for_backward(self, *ts):
        self.saved_tensors = ts

    def mark_non_differentiable(self, *ts):
        pass

    def set_materialize_grads(self, flag):
        pass


class ConvBNReLUShortcut(torch.autograd.Function):
    """x + relu(bn(conv3x3(x))) for a channels-last bf16 map with as many output as input channels (the identity-shortcut blocks of
    SSTBEVBackbone, sst_bev_backbone.py:36-40) as ONE autograd node: Conv3x3Dense and BNReLURows composed, so that the gradient the
    shortcut carries to x is added inside the input-gradient convolution's store pass (gdmae_conv3x3_dense_add) - as two nodes the
    engine adds the two gradients of x in a pass of its own (3 x 439 MB per block at config D)."""

    @staticmethod
    def forward(ctx, x, weight, bias, dil, direct, gamma, beta, eps, bn):
        B, C, H, W = x.shape
        c1 = _Ctx()
        y, part = Conv3x3Dense.forward(c1, x, weight, bias, dil, direct, True)
        assert y.shape == x.shape
        rows = y.permute(0, 2, 3, 1).reshape(B * H * W, C)
        res = x.permute(0, 2, 3, 1).reshape(B * H * W, C)
        c2 = _Ctx()
        out, _, _ = BNReLURows.forward(c2, rows, gamma, beta, eps, bn, res, part)
        # the tensors the two bodies saved go through the REAL save_for_backward (version check against in-place edits, saved-tensor
        # hooks, release with the graph); the stand-ins keep the non-tensor metadata only
        t1, t2 = c1.saved_tensors, c2.saved_tensors
        c1.saved_tensors = c2.saved_tensors = ()
        ctx.save_for_backward(*t1, *t2)
        ctx.n1 = len(t1)
        ctx.c1, ctx.c2 = c1, c2
        ctx.set_materialize_grads(False)
        return out.view(B, H, W, C).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None,) * 9
        c1, c2 = ctx.c1, ctx.c2
        sv = ctx.saved_tensors
        c1.saved_tensors, c2.saved_tensors = sv[:ctx.n1], sv[ctx.n1:]
        B, H, W, C = c1.saved_tensors[0].shape
        grows = g.permute(0, 2, 3, 1).reshape(B * H * W, C)
        grows = (grows if grows.dtype == torch.bfloat16 else grows.to(torch.bfloat16)).contiguous()
        dy, dgamma, dbeta, _, _, gres, _ = BNReLURows.backward(c2, grows, None, None)
        c1.needs_input_grad = (ctx.needs_input_grad[0],)
        c1.dx_addend = gres.view(B, H, W, C) if ctx.needs_input_grad[0] else None
        dx, dW, db, _, _, _ = Conv3x3Dense.backward(c1, dy.view(B, H, W, C).permute(0, 3, 1, 2))
        c1.dx_addend = None
        c1.saved_tensors = c2.saved_tensors = ()      # (a second backward over a retained graph takes them from ctx.saved_tensors again)
        return dx, dW, db, None, None, dgamma, dbeta, None, None


def _split_bf16(x32: torch.Tensor):
    """fp32 -> three bf16 pieces (value, remainder, remainder of the remainder): x = p0 + p1 + p2 exactly up to 2^-25 |x| (3 x 8 mantissa
    bits), each piece the bf16 rounding of what the pieces before it left."""
    p0 = x32.to(torch.bfloat16)
    r = x32 - p0.float()
    p1 = r.to(torch.bfloat16)
    p2 = (r - p1.float()).to(torch.bfloat16)
    return p0, p1, p2


# products of pieces (operand piece, weight piece) that matter at fp32 accuracy: |p_i| <= 2^-8i |x|, so the six pairs with i + j <= 2 reach
# 2^-24 of a product; summed smallest first
_PAIRS = ((2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0))


def _packed_split(w: torch.Tensor, cin: int, cout: int, dil: int, transposed: int):
    """fragment-ordered images of the three bf16 pieces of W (the pack kernel rounds what it is given)."""
    n = L.load().gdmae_conv3x3_dense_packed_bytes(cin, cout)
    out = []
    r = w
    for _ in range(3):
        img = torch.empty(n, dtype=torch.uint8, device=w.device)
        L.call("gdmae_conv3x3_dense_pack", L.ptr(r), cin, cout, dil, transposed, L.ptr(img), L.stream())
        out.append(img)
        r = (r - r.to(torch.bfloat16).float()).contiguous()
    return out


def conv3x3_f32_rows(xr: torch.Tensor, w: torch.Tensor, bias, dil: int, transposed: int = 0, split=None) -> torch.Tensor:
    """fp32-accurate 3 x 3 convolution of a channels-last fp32 map xr (B, H, W, C): operands split into three bf16 pieces each, the six
    piece products that matter accumulated in fp32 across six launches of the bf16 matrix-core kernel (k_conv3x3_dense<.., OF32>);
    -> (B, H, W, pad32(out channels)) fp32.  transposed = 1: the input gradient (xr = the output gradient, w the layer's (cout, cin, 3,
    3) weight).  split: the pieces of xr when the caller has them."""
    B, H, W, C = xr.shape
    cout, cin = w.shape[0], w.shape[1]
    o_l = _pad32(cin if transposed else cout)
    xs = split if split is not None else _split_bf16(xr)
    wsp = _packed_split(w, cin, cout, dil, transposed)
    y = torch.empty(B, H, W, o_l, dtype=torch.float32, device=xr.device)
    bl = None
    if bias is not None:
        bl = torch.zeros(o_l, dtype=torch.float32, device=xr.device)
        bl[:bias.numel()] = bias.detach().float()
    for n, (i, j) in enumerate(_PAIRS):
        last = n == len(_PAIRS) - 1
        L.call("gdmae_conv3x3_dense_f32out", L.ptr(xs[i]), B, H, W, C, o_l, dil, L.ptr(wsp[j]), L.ptr(bl) if (bl is not None and last) else None,
               L.ptr(y), int(n > 0), L.stream())
    return y


class Conv3x3DenseF32(torch.autograd.Function):
    """Conv3x3Dense for fp32 maps (no autocast: the parity mode and the fp32 fine-tune chain the goldens pin): every product as six bf16
    matrix-core launches on three-piece splits of its operands with fp32 accumulation - forward, input gradient and weight gradient -
    instead of F.conv2d -> MIOpen.  A product is reproduced to 2^-24: fp32 accuracy (a two-piece split, 2^-16, moved the most
    cancellation-prone gradient of the detector golden - one attention temperature - by 40 %)."""

    @staticmethod
    def forward(ctx, x, weight, bias, dil, direct):
        B, cin, H, W = x.shape
        cout = weight.shape[0]
        xr = x.permute(0, 2, 3, 1).float().contiguous()
        xs = _split_bf16(xr)
        w = weight.detach().float().contiguous()
        y = conv3x3_f32_rows(xr, w, bias, dil, 0, xs)
        ctx.save_for_backward(*xs, w)
        ctx.meta = (dil, cout, y.shape[-1], bias is not None, direct)
        return (y if y.shape[-1] == cout else y[..., :cout]).permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, dy):
        x0, x1, x2, w = ctx.saved_tensors
        xs = (x0, x1, x2)
        dil, cout, cl, has_bias, direct = ctx.meta
        B, H, W, cin = x0.shape
        dev = x0.device
        g = dy.permute(0, 2, 3, 1).float()
        if cl == cout:
            g = g.contiguous()
        else:
            gp = torch.zeros(B, H, W, cl, dtype=torch.float32, device=dev)
            gp[..., :cout] = g
            g = gp
        gs = _split_bf16(g)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = conv3x3_f32_rows(g, w, None, dil, 1, gs).permute(0, 3, 1, 2)
        dwd, dbd = direct if direct is not None else (None, None)
        dW = dwd if dwd is not None else torch.zeros(w.shape, dtype=torch.float32, device=dev)
        ws = torch.empty(L.load().gdmae_conv3x3_dense_dw_workspace_bytes(B, H, W, cin, cl), dtype=torch.uint8, device=dev)
        for i, j in _PAIRS:
            L.call("gdmae_conv3x3_dense_bwd_weight", L.ptr(xs[i]), L.ptr(gs[j]), B, H, W, cin, cl, cin, cout, dil, L.ptr(dW), L.ptr(ws), L.stream())
        db = None
        if has_bias:
            db = ops.colsum_f32(g.view(B * H * W, cl))[:cout]
            if dbd is not None:
                dbd.add_(db)
                db = None
        return dx, (None if dwd is not None else dW), db, None, None


def conv3x3(conv: nn.Conv2d, x: torch.Tensor, want_stats: bool = False):
    """``conv(x)`` through the library's dense convolution when it qualifies (conv3x3_supported), the framework's otherwise.
    want_stats: -> (y, partial column-sum rows of y or None) - the statistics of a BatchNorm that follows, from the epilogue."""
    if not conv3x3_supported(conv, x):
        y = conv(x)
        return (y, None) if want_stats else y
    dw = ops.direct_grad(conv.weight)
    db = ops.direct_grad(conv.bias) if conv.bias is not None else None
    direct = (dw, db) if (dw is not None and (conv.bias is None or db is not None)) else None
    if torch.is_autocast_enabled():
        return Conv3x3Dense.apply(x, conv.weight, conv.bias, int(conv.dilation[0]), direct, want_stats)
    y = Conv3x3DenseF32.apply(x, conv.weight, conv.bias, int(conv.dilation[0]), direct)
    return (y, None) if want_stats else y


def rows_supported(y: torch.Tensor, bn: nn.BatchNorm2d) -> bool:
    C = y.shape[1]
    return (y.is_cuda and y.dim() == 4 and bn.training and bn.affine and C % 8 == 0 and C <= 256 and
            y.dtype in (torch.float32, torch.bfloat16) and y.is_contiguous(memory_format=torch.channels_last))


def bn_relu_2d(y: torch.Tensor, bn: nn.BatchNorm2d, shortcut: torch.Tensor | None = None, partials=None) -> torch.Tensor:
    """relu(bn(y)) [+ shortcut] for a channels-last (B, C, Y, X) map in training mode; the result is channels-last in y's dtype.
    partials: the column-sum rows y's producer left (conv3x3 want_stats) - the statistics pass over y is skipped."""
    B, C, Y, X = y.shape
    rows = y.permute(0, 2, 3, 1).reshape(B * Y * X, C)                 # a view: channels-last storage
    res = None
    if shortcut is not None:
        res = shortcut.to(y.dtype).permute(0, 2, 3, 1).reshape(B * Y * X, C)
    out, _, _ = BNReLURows.apply(rows, bn.weight, bn.bias, float(bn.eps), bn, res, partials)
    return out.view(B, Y, X, C).permute(0, 3, 1, 2)


def conv_bn_relu(block: nn.Sequential, x: torch.Tensor, shortcut: torch.Tensor | None = None) -> torch.Tensor:
    """``block`` = Sequential(conv | deconv, BatchNorm2d, ReLU): the convolution through the framework (MIOpen, NHWC), the rest through
    the row kernels whenever the map qualifies (training mode, channels-last, C % 8 == 0, C <= 256); the module sequence otherwise
    (evaluation mode uses the running statistics: a per-channel affine the framework fuses itself).  ``shortcut``: a map of the output's
    shape added after the ReLU (in the same pass on the row path)."""
    if len(block) == 3 and isinstance(block[1], nn.BatchNorm2d) and isinstance(block[2], nn.ReLU):
        conv, bn = block[0], block[1]
        if (FUSE_SHORTCUT and shortcut is x and isinstance(conv, nn.Conv2d) and torch.is_autocast_enabled() and conv3x3_supported(conv, x) and x.requires_grad
                and x.dtype == torch.bfloat16 and x.is_contiguous(memory_format=torch.channels_last) and conv.out_channels == x.shape[1]
                and conv.out_channels % 32 == 0 and rows_supported(x, bn)):
            dw = ops.direct_grad(conv.weight)
            db = ops.direct_grad(conv.bias) if conv.bias is not None else None
            direct = (dw, db) if (dw is not None and (conv.bias is None or db is not None)) else None
            return ConvBNReLUShortcut.apply(x, conv.weight, conv.bias, int(conv.dilation[0]), direct, bn.weight, bn.bias, float(bn.eps), bn)
        y, part = conv3x3(block[0], x, want_stats=True) if isinstance(block[0], nn.Conv2d) else (block[0](x), None)
        if rows_supported(y, block[1]) and (shortcut is None or shortcut.is_contiguous(memory_format=torch.channels_last)):
            return bn_relu_2d(y, block[1], shortcut, part)
        y = block[2](block[1](y))
        return y if shortcut is None else y + shortcut
    y = block(x)
    return y if shortcut is None else y + shortcut


def conv_bn_relu_cat(blocks, xs, ys=None) -> torch.Tensor:
    """cat([block_i(x_i)], dim=1) for Sequential(conv | deconv, BatchNorm2d, ReLU) blocks on maps of one spatial size (the decoder's
    three deblocks, spt_backbone.py:296-303 / spt_backbone_mae.py:125-132): BatchNorm + ReLU of every branch write their column
    slice of the channels-last result directly - no cat pass, and the backward reads its slice of the gradient in place."""
    if ys is None:                                    # ys: the blocks' convolution outputs, computed by the caller
        ys = [b[0](x) for b, x in zip(blocks, xs)]
    ok = all(len(b) == 3 and isinstance(b[1], nn.BatchNorm2d) and isinstance(b[2], nn.ReLU) and rows_supported(y, b[1])
             for b, y in zip(blocks, ys))
    ok = ok and len({(y.shape[0],) + tuple(y.shape[2:]) + (y.dtype,) for y in ys}) == 1 and len({float(b[1].eps) for b in blocks}) == 1
    if not ok:
        # whatever follows the convolution in each block, applied as the module sequence would (blocks that are not conv + BN + ReLU)
        return torch.cat([nn.Sequential(*list(b)[1:])(y) for b, y in zip(blocks, ys)], dim=1)
    B, _, Y, X = ys[0].shape
    args = []
    for b, y in zip(blocks, ys):
        args += [y.permute(0, 2, 3, 1).reshape(B * Y * X, y.shape[1]), b[1].weight, b[1].bias]
    out = BNReLURowsCat.apply(tuple(b[1] for b in blocks), float(blocks[0][1].eps), *args)
    return out.view(B, Y, X, -1).permute(0, 3, 1, 2)
