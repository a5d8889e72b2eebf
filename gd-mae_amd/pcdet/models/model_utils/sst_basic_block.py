"""Encoder layer stack with the reference's module tree (pcdet/models/model_utils/sst_basic_block.py):
``BasicShiftBlockV2.encoder_list[k]`` = post-norm ``EncoderLayer`` with ``win_attn.self_attn``,
``linear1/linear2``, ``norm1/norm2``; layer 0 uses the un-shifted window partition, layer 1 the
shifted one (:100-114)."""
import torch.nn as nn
import torch.nn.functional as F

from .cosine_msa import CosineMultiheadAttention
import torch

from gdmae_hip import encoder as genc, ops


class WindowAttention(nn.Module):
    def __init__(self, d_model, nhead, dropout, batch_first=False, layer_cfg=None):
        super().__init__()
        layer_cfg = layer_cfg or {}
        if not layer_cfg.get('cosine', False):
            raise NotImplementedError("GD-MAE uses LAYER_CFG.cosine = True")
        self.nhead = nhead
        self.self_attn = CosineMultiheadAttention(d_model, nhead, dropout=dropout, tau_min=layer_cfg.get('tau_min', 0.01),
                                                  non_shared_tau=layer_cfg.get('non_shared_tau', False))

    def forward(self, feat_2d, pos, wplan):
        return self.self_attn(feat_2d, pos, wplan)


class EncoderLayer(nn.Module):
    # True: hand-written forward/backward of the whole layer (gdmae_hip/encoder.py, ~55 launches);
    # False: the same arithmetic op by op through autograd (~140 launches) - kept as the A/B reference.
    fused = True

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", batch_first=False,
                 mlp_dropout=0, layer_cfg=None):
        super().__init__()
        if mlp_dropout != 0:
            raise NotImplementedError
        self.win_attn = WindowAttention(d_model, nhead, dropout, layer_cfg=layer_cfg)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model)
        self.norm2 = nn.LayerNorm(d_model)
        if activation not in ("gelu", "relu"):
            raise RuntimeError(f"activation should be relu/gelu, not {activation}.")
        self.activation = F.gelu if activation == "gelu" else F.relu
        self.activation_name = activation

    def forward(self, src, pos_table, wplan):
        if self.fused and self.activation_name == "gelu":
            return genc.encoder_layer(self, src, wplan, pos_table)
        pos = torch.index_select(pos_table, 0, wplan.tok_pos)
        src = ops.add_layer_norm(src, self.win_attn(src, pos, wplan), self.norm1)
        h = self.activation(ops.linear(src, self.linear1.weight, self.linear1.bias))
        return ops.add_layer_norm(src, ops.linear(h, self.linear2.weight, self.linear2.bias), self.norm2)


class BasicShiftBlockV2(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation="relu", batch_first=False,
                 layer_cfg=None):
        super().__init__()
        self.encoder_list = nn.ModuleList([
            EncoderLayer(d_model, nhead, dim_feedforward, dropout, activation, batch_first, layer_cfg=layer_cfg)
            for _ in range(2)])

    def forward(self, src, pos_table, wplan_list):
        for i, layer in enumerate(self.encoder_list):
            src = layer(src, pos_table, wplan_list[i % len(wplan_list)])
        return src
