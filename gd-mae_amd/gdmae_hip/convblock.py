"""Sparse conv (k3) -> BatchNorm1d(train) -> ReLU as ONE native call per direction (csrc/conv_block.hip).

Used by pcdet.utils.spconv_utils.SparseSequential when the block's parameters live in a flat optimizer buffer
(gradients are accumulated there directly); otherwise the op-by-op path (ops.SparseConv3x3 + vfe.BNReLURows, same
arithmetic) runs.  Reference: post_act_block, pcdet/utils/spconv_utils.py:37-56.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L
from . import ops, packing


import os

IMPLICIT = os.environ.get("GDMAE_SPCONV", "1") != "0"    # False: im2col gather + library GEMMs (round-2 schedule, A/B reference)


class ConvBNReLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, info):
        weight, bn, nbr, nbr_t, direct, out_f32 = info
        cout, _, _, cin = weight.shape
        n_in, n_out = x.shape[0], nbr.shape[0]
        dev = x.device
        cdt = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
        assert x.dtype in (torch.float32, cdt)
        x = x.contiguous()
        wc = ops.shadow(weight, cdt).contiguous()
        a = L.ConvBlockArgs()
        a.n_in, a.n_out, a.cin, a.cout = n_in, n_out, cin, cout
        a.bf16, a.x_f32 = int(cdt == torch.bfloat16), int(x.dtype == torch.float32 and cdt != torch.float32)
        a.eps = float(bn.eps)
        a.x, a.nbr, a.nbr_t, a.W = L.ptr(x), L.ptr(nbr), L.ptr(nbr_t), L.ptr(wc)
        a.gamma, a.beta = L.ptr(bn.weight), L.ptr(bn.bias)
        if bn.training and bn.running_mean is not None:
            a.momentum = float(bn.momentum)
            a.running_mean, a.running_var, a.num_batches = L.ptr(bn.running_mean), L.ptr(bn.running_var), L.ptr(bn.num_batches_tracked)
        # bf16 with 128 / 256 channels: implicit GEMM over the rulebook (csrc/spconv.hip) on packed weight images that the
        # optimizer refreshes once per step - no im2col matrix is built or kept
        packed = None
        if cdt == torch.bfloat16 and IMPLICIT and packing.conv_supported(weight):
            packed = packing.conv_registered(weight)
            a.packed_fwd, a.packed_bwd = L.ptr(packed[0]), L.ptr(packed[1])
        cols = torch.empty(n_out, 9 * cin, dtype=cdt, device=dev) if packed is None else torch.empty(0, dtype=cdt, device=dev)
        y = torch.empty(n_out, cout, dtype=cdt, device=dev)
        out = torch.empty(n_out, cout, dtype=torch.float32 if out_f32 else cdt, device=dev)
        a.out_f32 = int(out_f32 and cdt != torch.float32)
        stats = torch.empty(2 * cout, dtype=torch.float64, device=dev)
        abmv = torch.empty(4 * cout, dtype=torch.float32, device=dev)
        nb = L.load().gdmae_conv_block_scratch_bytes(n_in, n_out, cin, cout, a.bf16)
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        a.cols, a.y, a.out, a.stats, a.ab, a.mv = (L.ptr(cols) if packed is None else None, L.ptr(y), L.ptr(out), L.ptr(stats),
                                                   L.ptr(abmv), L.ptr(abmv[2 * cout:]))
        a.scratch = L.ptr(scratch)
        L.call("gdmae_conv_block_fwd", C.byref(a), L.stream())
        ctx.save_for_backward(x, cols, y, stats, abmv, wc, nbr, nbr_t, bn.weight.detach())
        ctx.meta = (a, nb, direct, x.dtype)
        ctx.packed = packed
        return out

    @staticmethod
    def backward(ctx, g):
        x, cols, y, stats, abmv, wc, nbr, nbr_t, gamma = ctx.saved_tensors
        a0, nb, direct, x_dtype = ctx.meta
        a = L.ConvBlockArgs.from_buffer_copy(a0)
        g = g.contiguous()
        assert g.dtype in (torch.float32, y.dtype)
        a.g, a.g_f32 = L.ptr(g), int(g.dtype == torch.float32 and y.dtype != torch.float32)
        dx = torch.empty(x.shape[0], a.cin, dtype=y.dtype, device=x.device) if ctx.needs_input_grad[0] else None
        a.dx = L.ptr(dx) if dx is not None else None
        a.dW, a.dgamma, a.dbeta = (L.ptr(t) for t in direct)
        scratch = torch.empty(nb, dtype=torch.uint8, device=x.device)
        a.scratch = L.ptr(scratch)
        L.call("gdmae_conv_block_bwd", C.byref(a), L.stream())
        return dx, None


def conv_bn_relu(x, conv, bn, nbr, nbr_t, out_f32=False):
    """-> features (n_out, cout) or None if the block's parameters are not owned by a flat optimizer (caller falls back).
    ``out_f32``: write the block output in fp32 also in bf16 mode (it feeds the fp32 residual stream of an encoder stage)."""
    direct = [ops.direct_grad(p) for p in (conv.weight, bn.weight, bn.bias)]
    if any(t is None for t in direct) or not x.is_cuda:
        return None
    return ConvBNReLUFn.apply(x, (conv.weight, bn, nbr, nbr_t, direct, out_f32))
