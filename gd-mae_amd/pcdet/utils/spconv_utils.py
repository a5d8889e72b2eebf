"""Sparse 2-D convolution front end with the reference's names (``pcdet/utils/spconv_utils.py``:
``spconv.SparseConvTensor``, ``SubMConv2d``, ``SparseConv2d``, ``SparseSequential``, ``post_act_block``,
``replace_feature``, ``find_all_spconv_keys``) implemented on the HIP geometry plan - no spconv.

A ``SparseConvTensor`` here is (features, stage of an ``EncoderPlan``): its active set, the rulebooks
of the convolutions that may be applied to it and its window partitions were all built up-front on the
GPU by ``gdmae_hip.plan.encoder_plan``; the convolution modules only gather rows and run one GEMM.
Weights use the spconv-2.x layout ``(Cout, kH, kW, Cin)`` so reference checkpoints load by key+shape
(reference ``pcdet/models/detectors/detector3d_template.py:361-411``).
"""
from __future__ import annotations

import math
import os
import types
from typing import Set

import torch
import torch.nn as nn

from gdmae_hip import ops


class SparseConvTensor:
    def __init__(self, features, plan, stage: int):
        self.features = features
        self._plan = plan
        self._stage = stage

    @property
    def stage_plan(self):
        return self._plan.stages[self._stage]

    @property
    def indices(self):          # (n, 3) int32 (b, y, x) like spconv
        return self.stage_plan.indices_byx()

    @property
    def spatial_shape(self):
        sp = self.stage_plan
        return [sp.Y, sp.X]

    @property
    def batch_size(self):
        return self.stage_plan.B

    def replace_feature(self, f):
        return SparseConvTensor(f, self._plan, self._stage)

    def dense(self, channels_last: bool = True):
        """(B, C, Y, X), zeros at inactive sites; stored channels-last (rows of C are what the scatter writes)."""
        sp = self.stage_plan
        flat = ops.ScatterToDense.apply(self.features, sp.tok_cell, sp.B * sp.Y * sp.X)
        return flat.view(sp.B, sp.Y, sp.X, -1).permute(0, 3, 1, 2)


class SparseModule(nn.Module):
    pass


class SparseConvolution(SparseModule):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, indice_key=None, subm=False):
        super().__init__()
        if bias or kernel_size != 3 or (not subm and (stride != 2 or padding != 1)):
            raise NotImplementedError("hot path uses SubMConv2d(k3) and SparseConv2d(k3, s2, p1) without bias only")
        self.in_channels, self.out_channels, self.subm, self.indice_key = in_channels, out_channels, subm, indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, 3, 3, in_channels))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def rulebooks(self, x: SparseConvTensor):
        """(nbr, nbr_t, output stage index) of this convolution on ``x``'s active set."""
        if self.subm:
            sp = x.stage_plan
            if getattr(sp, "_nbr_subm_t", None) is None:      # transposed rulebook of a submanifold conv = tap-reversed rulebook
                sp._nbr_subm_t = torch.flip(sp.nbr_subm, dims=[1]).contiguous()
            return sp.nbr_subm, sp._nbr_subm_t, x._stage
        nxt = x._plan.stages[x._stage + 1]
        return nxt.nbr_down, nxt.nbr_down_t, x._stage + 1

    def forward(self, x: SparseConvTensor) -> SparseConvTensor:
        nbr, nbr_t, stage = self.rulebooks(x)
        f = ops.SparseConv3x3.apply(x.features, self.weight, nbr, nbr_t)
        return x.replace_feature(f) if stage == x._stage else SparseConvTensor(f, x._plan, stage)


class SubMConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, bias=False, indice_key=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, 1, kernel_size // 2, bias, indice_key, subm=True)


class SparseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False, indice_key=None, **kw):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias, indice_key, subm=False)


class SparseSequential(SparseModule):
    fused_bn_relu = True     # conv -> BatchNorm1d(train) -> ReLU as one statistics pass + one fused row pass
    native_block = os.environ.get("GDMAE_CONV_BLOCK", "1") != "0"   # whole block as one native call when a flat optimizer owns it
    out_fp32 = False         # native block: fp32 output also under autocast (skips one bf16 rounding + the consumer's cast pass; off:
                             # the block then rounds exactly where the op-by-op path does, which the A/B test pins to 1e-6)

    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def forward(self, x):
        mods = list(self._modules.values())
        if (self.fused_bn_relu and self.training and len(mods) == 3 and isinstance(mods[1], nn.BatchNorm1d)
                and isinstance(mods[2], nn.ReLU) and mods[1].affine):
            from gdmae_hip import convblock, vfe as gvfe
            conv, bn = mods[0], mods[1]
            if self.native_block and isinstance(conv, SparseConvolution):
                nbr, nbr_t, stage = conv.rulebooks(x)
                f = convblock.conv_bn_relu(x.features, conv, bn, nbr, nbr_t, self.out_fp32)     # one native call per direction
                if f is not None:
                    return x.replace_feature(f) if stage == x._stage else SparseConvTensor(f, x._plan, stage)
            x = conv(x)
            f, _, _ = gvfe.BNReLURows.apply(x.features, bn.weight, bn.bias, bn.eps, bn)
            return x.replace_feature(f)
        for m in mods:
            x = m(x) if isinstance(m, SparseModule) else x.replace_feature(m(x.features))
        return x


# namespace object so code written as ``spconv.SubMConv2d`` keeps working
spconv = types.SimpleNamespace(SparseConvTensor=SparseConvTensor, SparseModule=SparseModule, SubMConv2d=SubMConv2d,
                               SparseConv2d=SparseConv2d, SparseSequential=SparseSequential,
                               conv=types.SimpleNamespace(SparseConvolution=SparseConvolution))


def find_all_spconv_keys(model: nn.Module, prefix="") -> Set[str]:
    found: Set[str] = set()
    for name, child in model.named_children():
        p = f"{prefix}.{name}" if prefix else name
        if isinstance(child, SparseConvolution):
            found.add(f"{p}.weight")
        found.update(find_all_spconv_keys(child, prefix=p))
    return found


def replace_feature(out, new_features):
    return out.replace_feature(new_features)


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm',
                   norm_fn=None, dim=2):
    if dim != 2:
        raise NotImplementedError("only 2-D sparse convolutions are on the GD-MAE hot path")
    if conv_type == 'subm':
        conv = SubMConv2d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = SparseConv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                            indice_key=indice_key)
    else:
        raise NotImplementedError(conv_type)
    return SparseSequential(conv, norm_fn(out_channels), nn.ReLU())
