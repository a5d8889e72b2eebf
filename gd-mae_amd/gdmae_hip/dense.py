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

from .vfe import BNReLURows, BNReLURowsCat


def rows_supported(y: torch.Tensor, bn: nn.BatchNorm2d) -> bool:
    C = y.shape[1]
    return (y.is_cuda and y.dim() == 4 and bn.training and bn.affine and C % 8 == 0 and C <= 256 and
            y.dtype in (torch.float32, torch.bfloat16) and y.is_contiguous(memory_format=torch.channels_last))


def bn_relu_2d(y: torch.Tensor, bn: nn.BatchNorm2d, shortcut: torch.Tensor | None = None) -> torch.Tensor:
    """relu(bn(y)) [+ shortcut] for a channels-last (B, C, Y, X) map in training mode; the result is channels-last in y's dtype."""
    B, C, Y, X = y.shape
    rows = y.permute(0, 2, 3, 1).reshape(B * Y * X, C)                 # a view: channels-last storage
    res = None
    if shortcut is not None:
        res = shortcut.to(y.dtype).permute(0, 2, 3, 1).reshape(B * Y * X, C)
    out, _, _ = BNReLURows.apply(rows, bn.weight, bn.bias, float(bn.eps), bn, res)
    return out.view(B, Y, X, C).permute(0, 3, 1, 2)


def conv_bn_relu(block: nn.Sequential, x: torch.Tensor, shortcut: torch.Tensor | None = None) -> torch.Tensor:
    """``block`` = Sequential(conv | deconv, BatchNorm2d, ReLU): the convolution through the framework (MIOpen, NHWC), the rest through
    the row kernels whenever the map qualifies (training mode, channels-last, C % 8 == 0, C <= 256); the module sequence otherwise
    (evaluation mode uses the running statistics: a per-channel affine the framework fuses itself).  ``shortcut``: a map of the output's
    shape added after the ReLU (in the same pass on the row path)."""
    if len(block) == 3 and isinstance(block[1], nn.BatchNorm2d) and isinstance(block[2], nn.ReLU):
        y = block[0](x)
        if rows_supported(y, block[1]) and (shortcut is None or shortcut.is_contiguous(memory_format=torch.channels_last)):
            return bn_relu_2d(y, block[1], shortcut)
        y = block[2](block[1](y))
        return y if shortcut is None else y + shortcut
    y = block(x)
    return y if shortcut is None else y + shortcut


def conv_bn_relu_cat(blocks, xs) -> torch.Tensor:
    """cat([block_i(x_i)], dim=1) for Sequential(conv | deconv, BatchNorm2d, ReLU) blocks on maps of one spatial size (the decoder's
    three deblocks, spt_backbone.py:296-303 / spt_backbone_mae.py:125-132): BatchNorm + ReLU of every branch write their column
    slice of the channels-last result directly - no cat pass, and the backward reads its slice of the gradient in place."""
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
