"""Cosine multi-head attention parameters + windowed forward on the HIP kernel.

Parameter names/shapes follow ``CosineMultiheadAttention(nn.MultiheadAttention)`` of the reference
(pcdet/models/model_utils/cosine_msa.py:441-459): ``in_proj_weight (3d, d)``, ``in_proj_bias``,
``out_proj.{weight,bias}``, learnable ``tau (1,1,1)`` clamped at ``tau_min``.  The forward differs by
design: it consumes flat tokens + a window CSR (no padded (T, nW, d) tensors, no key-padding mask).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from gdmae_hip import ops


class CosineMultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0., bias=True, batch_first=False, cosine=True, tau_min=0.01,
                 non_shared_tau=False, **kw):
        super().__init__()
        if dropout != 0. or not cosine or non_shared_tau or not bias:
            raise NotImplementedError("hot path: cosine attention, shared tau, dropout 0 (gd_mae_ssl.yaml:70-75)")
        self.embed_dim, self.num_heads, self.tau_min = embed_dim, num_heads, tau_min
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        self.tau = nn.Parameter(torch.ones(1, 1, 1))
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.)

    def forward(self, x, pos, wplan):
        """x (n, d) tokens, pos (n, d) positional embedding of the tokens, wplan: gdmae_hip.plan.WindowPlan.
        q = k = x + pos, v = x with separate q/k/v slices of the packed projection (cosine_msa.py:56-62)."""
        d = self.embed_dim
        qk = ops.linear(x + pos, self.in_proj_weight[:2 * d], self.in_proj_bias[:2 * d])
        v = ops.linear(x, self.in_proj_weight[2 * d:], self.in_proj_bias[2 * d:])
        o = ops.WindowCosineAttention.apply(qk, v, self.tau, wplan, self.num_heads, self.tau_min)
        return ops.linear(o, self.out_proj.weight, self.out_proj.bias)
