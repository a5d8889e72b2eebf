"""Dynamic pillar VFE on the HIP voxelizer (interface of the reference ``DynVFE``,
pcdet/models/backbones_3d/vfe/dyn_vfe.py:11-124; parameters ``dvfe_mlps.0.{0,1,3,4}``).

forward: one ``gdmae_voxelize`` call (range filter, pillar table, canonical point CSR, per-pillar mean),
one decoration kernel, the two Linear/BN/ReLU layers as token GEMMs, and a segmented max over the CSR.
"""
import os
from functools import partial

import torch
import torch.nn as nn

from .vfe_template import VFETemplate
from ...model_utils.network_utils import make_fc_layers
from gdmae_hip import ops, plan as gplan, vfe as gvfe


# both fused layers as one node with fp16 rows between them (round 6); 0: two nodes, bf16 rows (A/B reference, tools/ab_env.sh)
VFE_F16 = os.environ.get("GDMAE_VFE_F16", "1") != "0"


class DynVFE(VFETemplate):
    fused = True      # fused BN+ReLU(+max) row kernels; False = torch BatchNorm1d / ReLU modules + segment max
    point_layer = os.environ.get("GDMAE_VFE_POINT_LAYER", "1") != "0"    # first layer as gdmae_vfe_point_layer_*
    max_layer = os.environ.get("GDMAE_VFE_MAX_LAYER", "1") != "0"        # last layer as gdmae_vfe_max_layer_* (bf16 mode)

    def __init__(self, model_cfg, num_point_features, voxel_size, point_cloud_range, grid_size, **kwargs):
        super().__init__(model_cfg=model_cfg)
        self.sample_type = model_cfg.get('TYPE', 'mean')
        mlps = model_cfg.get('MLPS', None)
        if self.sample_type != 'mean' or mlps is None or len(mlps) != 1 or model_cfg.get('AGGREGATION_MLPS', None):
            raise NotImplementedError("hot path: TYPE mean, one MLPS entry, no aggregation MLP (gd_mae_ssl.yaml:49-55)")
        if model_cfg.WITH_DISTANCE or not model_cfg.USE_ABSLOTE_XYZ or not model_cfg.USE_CLUSTER_XYZ:
            raise NotImplementedError("hot path: USE_ABSLOTE_XYZ and USE_CLUSTER_XYZ on, WITH_DISTANCE off")
        self.with_distance, self.use_absolute_xyz, self.use_cluster_xyz = False, True, True
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        self.dvfe_mlps = nn.ModuleList([make_fc_layers(mlps[0], num_point_features + 6, norm_fn=norm_fn)])
        self.aggregation_mlp = None
        self.num_point_features = mlps[0][-1]
        self.voxel_size, self.point_cloud_range, self.grid_size = voxel_size, point_cloud_range, grid_size

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        vox = batch_dict.get('_gdmae_vox', None)          # prefetched geometry plan (gdmae_hip.plan.PlanPrefetch)
        if vox is None:
            vox = gplan.voxelize(batch_dict['points'], self.point_cloud_range, self.voxel_size, self.grid_size,
                                 int(batch_dict['batch_size']))
        mlp = self.dvfe_mlps[0]
        if self.fused and self.training:
            nl = len(mlp) // 3
            # first layer: decoration + Linear + BatchNorm + ReLU in one call, nothing but its output stored
            first = (self.point_layer and nl > 1 and mlp[0].bias is None and mlp[0].out_features == 64
                     and 3 <= vox.n_cols - 1 <= 5 and vox.N > 0)
            # last layer (bf16 mode): Linear + BatchNorm + ReLU + pillar max with the pre-activation never stored; it wants
            # the rows of the first layer in pillar-major order
            last = (first and self.max_layer and nl == 2 and torch.is_autocast_enabled() and mlp[3].bias is None
                    and mlp[3].weight.shape[1] == 64 and mlp[3].weight.shape[0] in (128, 256) and vox.points_pm is not None)
            x = None if first else ops.decorate_points(vox)
            if first and last and VFE_F16:
                # both layers as one autograd node with fp16 rows between them (gdmae_hip.vfe.PointLayers12Max, round 6)
                l1, bn1, l2, bn2 = mlp[0], mlp[1], mlp[3], mlp[4]
                x = gvfe.PointLayers12Max.apply(vox, l1.weight, bn1.weight, bn1.bias, bn1.eps, bn1, l2.weight, bn2.weight, bn2.bias, bn2.eps, bn2)
                nl = 0
            for k in range(nl):
                lin, bn = mlp[3 * k], mlp[3 * k + 1]
                if k == 0 and first:
                    x, _, _ = gvfe.PointLayer1.apply(vox, lin.weight, bn.weight, bn.bias, bn.eps, bn, last)
                    continue
                if k == nl - 1 and last:
                    x = gvfe.point_layer2_max(x, vox.row_pillar, lin.weight, bn, vox.pt_off)
                    continue
                x = ops.linear(x, lin.weight, lin.bias)
                if k < nl - 1:
                    x, _, _ = gvfe.BNReLURows.apply(x, bn.weight, bn.bias, bn.eps, bn)
                else:
                    x, _, _ = gvfe.BNReLUSegmentMax.apply(x, bn.weight, bn.bias, bn.eps, vox.pt_off, vox.pillar_pts,
                                                          vox.inverse32, bn)
        else:
            x = ops.decorate_points(vox)
            for m in mlp:                                # Linear(no bias) -> BN1d -> ReLU, twice
                x = ops.linear(x, m.weight, m.bias) if isinstance(m, nn.Linear) else m(x)
            x = ops.SegmentMax.apply(x.float(), vox.pt_off, vox.pillar_pts, vox.inverse32)
        batch_dict.update({'points': vox.points, 'point_coords': vox.point_coords,
                           'point_inverse_indices': vox.inverse, 'voxel_coords': vox.voxel_coords,
                           'pillar_features': x, 'voxel_features': x, '_gdmae_vox': vox})
        return batch_dict
