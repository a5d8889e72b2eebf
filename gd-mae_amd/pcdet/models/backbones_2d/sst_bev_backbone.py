"""BEV backbone of the GD-MAE fine-tune detector: a stack of Conv2d(3x3, optional dilation) + BatchNorm2d(eps 1e-3,
momentum 0.01) + ReLU blocks with identity shortcuts where shapes allow.  Parameter names / shapes and the
``spatial_features -> spatial_features_2d`` contract follow the reference ``SSTBEVBackbone``
(pcdet/models/backbones_2d/sst_bev_backbone.py:6-42) so fine-tune checkpoints load by key.  The maps here are dense (every
pillar is a token in the fine-tune path), so the convolutions are plain dense contractions issued through PyTorch-ROCm; BatchNorm +
ReLU of their channels-last outputs run through the row kernels of the hot path (gdmae_hip.dense)."""
import torch.nn as nn

from gdmae_hip import dense as gdense


class SSTBEVBackbone(nn.Module):
    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        c = model_cfg.NUM_FILTER
        self.conv_shortcut = list(model_cfg.CONV_SHORTCUT)
        self.conv_layer = nn.ModuleList()
        for kw in model_cfg.CONV_KWARGS:
            kw = dict(kw)
            self.conv_layer.append(nn.Sequential(nn.Conv2d(c, bias=False, **kw),
                                                 nn.BatchNorm2d(kw['out_channels'], eps=1e-3, momentum=0.01), nn.ReLU(inplace=True)))
            c = kw['out_channels']
        self.num_bev_features = c

    def forward(self, data_dict):
        x = data_dict['spatial_features']
        for i, block in enumerate(self.conv_layer):
            conv = block[0]
            same = (conv.out_channels == x.shape[1] and tuple(conv.stride) == (1, 1) and
                    all(2 * p == d * (k - 1) for p, d, k in zip(conv.padding, conv.dilation, conv.kernel_size)))
            x = gdense.conv_bn_relu(block, x, shortcut=x if (same and i in self.conv_shortcut) else None)
        data_dict['spatial_features_2d'] = x
        return data_dict
