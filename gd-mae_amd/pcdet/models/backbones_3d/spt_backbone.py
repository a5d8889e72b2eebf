"""Sparse-regional-transformer blocks on the HIP geometry plan.

Module tree and parameter names follow the reference (pcdet/models/backbones_3d/spt_backbone.py:
``SSTInputLayer`` :11-194, ``SSTBlockV1`` :197-264, ``SPTBackbone`` :267-347) so checkpoints load by key;
the data flow does not: window ids, occupancy levels, dense window indices and in-window ranks come
precomputed from ``gdmae_hip.plan.encoder_plan`` (one wavefront per 8x8 window, no atomics / host
syncs), padded per-level tensors are never built, and the 2-D position embedding is a 64-row table
indexed by the token's in-window cell.
"""
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from ..model_utils.sst_basic_block import BasicShiftBlockV2
from ...utils.spconv_utils import post_act_block, replace_feature, SparseConvTensor
from gdmae_hip import dense as gdense
from gdmae_hip import encoder as genc
from gdmae_hip import ops, plan as gplan


class SSTInputLayer(nn.Module):
    """Holds the window / drop-level configuration of a stage and the position-embedding table
    (reference ``get_pos_embed``, spt_backbone.py:137-181: interleaved sin/cos of (x - wx/2, y - wy/2)
    over inv_freq = T^(2*floor(i/2)/(d/2)), concatenated [embed_x | embed_y])."""

    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.window_shape = list(model_cfg.WINDOW_SHAPE)
        if model_cfg.SHUFFLE_VOXELS:
            raise NotImplementedError("SHUFFLE_VOXELS (off in every shipped config)")
        drop_info = model_cfg.DROP_INFO['train' if self.training else 'test']   # always 'train' at construction
        self.drop_info = {int(k): v for k, v in drop_info.items()}
        self.pos_temperature = model_cfg.POS_TEMPERATURE
        self.normalize_pos = model_cfg.NORMALIZE_POS
        assert self.window_shape[2] == 1
        self._pos_cache = {}

    def pos_table(self, feat_dim, device, dtype=torch.float32):
        key = (feat_dim, str(device), dtype)
        if key not in self._pos_cache:
            wx, wy = self.window_shape[0], self.window_shape[1]
            cell = torch.arange(wx * wy, device=device)
            y = (cell // wx).float() - wy / 2
            x = (cell % wx).float() - wx / 2
            if self.normalize_pos:
                x = x / wx * 2 * 3.1415
                y = y / wy * 2 * 3.1415
            pos_length = feat_dim // 2
            assert pos_length * 2 == feat_dim
            i = torch.arange(pos_length, dtype=torch.float32, device=device)
            inv_freq = self.pos_temperature ** (2 * torch.div(i, 2, rounding_mode='floor') / pos_length)
            ex, ey = x[:, None] / inv_freq[None, :], y[:, None] / inv_freq[None, :]
            ex = torch.stack([ex[:, ::2].sin(), ex[:, 1::2].cos()], dim=-1).flatten(1)
            ey = torch.stack([ey[:, ::2].sin(), ey[:, 1::2].cos()], dim=-1).flatten(1)
            self._pos_cache[key] = torch.cat([ex, ey], dim=-1).to(dtype).contiguous()
        return self._pos_cache[key]


class SSTBlockV1(nn.Module):
    def __init__(self, model_cfg, input_channels, indice_key, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        enc = model_cfg.ENCODER
        d_model, stride = enc.D_MODEL, enc.STRIDE
        self.d_model, self.stride = d_model, stride
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.conv_down = post_act_block(input_channels, d_model, 3, norm_fn=norm_fn, stride=stride, padding=1,
                                        indice_key=f'{indice_key}_spconv', conv_type='spconv', dim=2) if stride > 1 else None
        self.sst_input_layer = SSTInputLayer(model_cfg.PREPROCESS)
        self.encoder_blocks = nn.ModuleList([
            BasicShiftBlockV2(d_model, enc.NHEAD, enc.DIM_FEEDFORWARD, enc.DROPOUT, enc.ACTIVATION, batch_first=False,
                              layer_cfg=enc.LAYER_CFG) for _ in range(enc.NUM_BLOCKS)])
        self.conv_out = post_act_block(d_model, d_model, 3, norm_fn=norm_fn, indice_key=f'{indice_key}_subm', dim=2)

    def forward(self, sp_tensor: SparseConvTensor) -> SparseConvTensor:
        if self.conv_down is not None:
            sp_tensor = self.conv_down(sp_tensor)
        feat = sp_tensor.features
        wplans = sp_tensor.stage_plan.windows
        table = self.sst_input_layer.pos_table(feat.shape[1], feat.device)
        # feat + block_k(... block_1(feat)): one call per stage, the block residual folded into it where the fused layer path runs
        res = genc.encoder_stage(self.encoder_blocks, feat, table, wplans, residual=True)
        sp_tensor = replace_feature(sp_tensor, res)             # token drop is the identity (no un-shuffle)
        return self.conv_out(sp_tensor)


def stage_plan_args(sst_block_list):
    """(strides, window shapes, drop infos) of a block list for ``gdmae_hip.plan.encoder_plan``."""
    return ([int(b.ENCODER.STRIDE) for b in sst_block_list], [list(b.PREPROCESS.WINDOW_SHAPE) for b in sst_block_list],
            [dict(b.PREPROCESS.DROP_INFO['train']) for b in sst_block_list])


def build_decoder(model_cfg):
    """deblocks + conv_out shared by SPTBackbone / SPTBackboneMAE (spt_backbone.py:282-303)."""
    deblocks, tot = nn.ModuleList(), 0
    for src in model_cfg.FEATURES_SOURCE:
        c = model_cfg.FUSE_LAYER[src]
        deblocks.append(nn.Sequential(
            nn.ConvTranspose2d(c.NUM_FILTER, c.NUM_UPSAMPLE_FILTER, c.UPSAMPLE_STRIDE, stride=c.UPSAMPLE_STRIDE, bias=False),
            nn.BatchNorm2d(c.NUM_UPSAMPLE_FILTER, eps=1e-3, momentum=0.01), nn.ReLU(inplace=True)))
        tot += c.NUM_UPSAMPLE_FILTER
    out_c = tot // len(deblocks)
    conv_out = nn.Sequential(nn.Conv2d(tot, out_c, 3, padding=1, bias=False),
                             nn.BatchNorm2d(out_c, eps=1e-3, momentum=0.01), nn.ReLU(inplace=True))
    return deblocks, conv_out, out_c


def deconv_map(h, deconv, Y: int, X: int):
    """ConvTranspose2d(k = s, stride s, no bias) of the DENSE map of a sparse stage output, without the dense product: a zero site
    maps to an s x s block of zeros, so the result is the token rows' product P (n s^2, cout) (gdmae_hip.decoder.deconv_rows: the
    library's row GEMM) scattered to the (token, dy, dx) sites of a zero map - what F.conv_transpose2d (MIOpen) computed on
    B Y X sites, 97 % of them empty (reference spt_backbone.py:296-299).  -> (B, cout, Y, X) view of a channels-last map."""
    from gdmae_hip import decoder as gdec
    sp = h.stage_plan
    s = int(deconv.stride[0])
    P = gdec.deconv_rows(h.features, deconv)
    sites = gdec.upsampled_sites(sp, s, Y, X)
    flat = ops.ScatterToDense.apply(P, sites, sp.B * Y * X)
    return flat.view(sp.B, Y, X, -1).permute(0, 3, 1, 2)


def run_decoder(model_cfg, deblocks, conv_out, hidden):
    """Densify each source stage -> ConvTranspose2d(k=s)+BN+ReLU -> cat -> Conv2d 3x3+BN+ReLU
    (spt_backbone_mae.py:125-133).  Dense maps are channels-last in memory.  Under autocast no library convolution runs: the
    deconvolutions are row products on the active tokens (deconv_map), conv_out is csrc/conv_dense.hip."""
    srcs = [hidden[int(src[-1]) - 1] for src in model_cfg.FEATURES_SOURCE]
    strides = [int(b[0].stride[0]) for b in deblocks]
    Y, X = srcs[0].stage_plan.Y * strides[0], srcs[0].stage_plan.X * strides[0]
    rows_ok = (torch.is_autocast_enabled() and all(
        isinstance(b[0], nn.ConvTranspose2d) and b[0].kernel_size == (s, s) and b[0].stride == (s, s) and b[0].bias is None and
        b[0].padding == (0, 0) and b[0].output_padding == (0, 0) and h.features.is_cuda and h.stage_plan.Y * s == Y and h.stage_plan.X * s == X
        for b, s, h in zip(deblocks, strides, srcs)))
    if rows_ok:
        ys = [deconv_map(h, b[0], Y, X) for b, h in zip(deblocks, srcs)]
        return gdense.conv_bn_relu(conv_out, gdense.conv_bn_relu_cat(list(deblocks), None, ys=ys))
    xs = [h.dense() for h in srcs]
    y = gdense.conv_bn_relu(conv_out, gdense.conv_bn_relu_cat(list(deblocks), xs))
    return y


class SPTBackbone(nn.Module):
    """Fine-tune variant without masking (reference spt_backbone.py:267-347)."""

    def __init__(self, model_cfg, input_channels, grid_size, voxel_size, point_cloud_range, **kwargs):
        super().__init__()
        self.model_cfg, self.grid_size, self.voxel_size, self.point_cloud_range = model_cfg, grid_size, voxel_size, point_cloud_range
        self.sparse_shape = grid_size[[1, 0]]
        c = input_channels
        self.sst_blocks = nn.ModuleList()
        for b in model_cfg.SST_BLOCK_LIST:
            self.sst_blocks.append(SSTBlockV1(b, c, b.NAME))
            c = b.ENCODER.D_MODEL
        self.deblocks, self.conv_out, self.num_point_features = build_decoder(model_cfg)

    def forward(self, batch_dict):
        vox = batch_dict['_gdmae_vox']
        ep = gplan.encoder_plan(vox, *stage_plan_args(self.model_cfg.SST_BLOCK_LIST), keep_frac=None)
        x = SparseConvTensor(batch_dict['voxel_features'], ep, 0)
        hidden = []
        for blk in self.sst_blocks:
            x = blk(x)
            hidden.append(x)
        feats, strides = {}, {}
        for i, h in enumerate(hidden):
            feats[f'x_conv{i + 1}'] = h
            strides[f'x_conv{i + 1}'] = 2 ** (i + 1)
        sf = run_decoder(self.model_cfg, self.deblocks, self.conv_out, hidden)
        batch_dict.update({'encoded_spconv_tensor': hidden[-1], 'encoded_spconv_tensor_stride': 2 ** len(hidden),
                           'multi_scale_3d_features': feats, 'multi_scale_3d_strides': strides, 'spatial_features': sf,
                           'spatial_features_stride': strides[self.model_cfg.FEATURES_SOURCE[0]] //
                           self.model_cfg.FUSE_LAYER[self.model_cfg.FEATURES_SOURCE[0]].UPSAMPLE_STRIDE})
        return batch_dict
