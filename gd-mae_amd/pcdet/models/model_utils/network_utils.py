"""``make_fc_layers`` with the reference's module layout (pcdet/models/model_utils/network_utils.py:7-21):
[Linear(no bias) -> norm -> ReLU] per entry, so ``dvfe_mlps.0.{0,1,3,4}`` state_dict keys line up."""
import torch.nn as nn


def make_fc_layers(fc_cfg, input_channels, output_channels=None, linear=True, norm_fn=None):
    layers, c_in = [], input_channels
    for c in fc_cfg:
        layers += [nn.Linear(c_in, c, bias=False) if linear else nn.Conv1d(c_in, c, kernel_size=1, bias=False),
                   nn.BatchNorm1d(c) if norm_fn is None else norm_fn(c), nn.ReLU()]
        c_in = c
    if output_channels is not None:
        layers.append(nn.Linear(c_in, output_channels) if linear else nn.Conv1d(c_in, output_channels, kernel_size=1))
    return nn.Sequential(*layers)
