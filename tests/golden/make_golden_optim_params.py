"""Which parameters does the REFERENCE's build_optimizer('adam_onecycle') actually optimise?  Builds the reference
DynVFE + SPTBackboneMAE (ref_harness) and the reference OptimWrapper exactly as tools/train_utils/optimization/__init__.py
does, and stores the optimised / skipped parameter names.  Run in the build container."""
import json
import os
import sys
from functools import partial

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd"), HERE]
import ref_harness as rh  # noqa: E402
from gdmae_hip import configs  # noqa: E402

ycfg = rh.load_yaml_cfg("cfgs/waymo_models/gd_mae_ssl.yaml")
mc = ycfg.MODEL
ds = configs.SyntheticDatasetInfo(point_cloud_range=[-74.88, -74.88, -2, 74.88, 74.88, 4], voxel_size=[0.32, 0.32, 6],
                                  num_point_features=5, class_names=["Vehicle", "Pedestrian", "Cyclist"])
dyn = rh.ref("pcdet.models.backbones_3d.vfe.dyn_vfe")
mae = rh.ref("pcdet.models.backbones_3d.spt_backbone_mae")
vfe = dyn.DynVFE(model_cfg=mc.VFE, num_point_features=5, voxel_size=ds.voxel_size, point_cloud_range=ds.point_cloud_range,
                 grid_size=ds.grid_size)
bb = mae.SPTBackboneMAE(model_cfg=mc.BACKBONE_3D, input_channels=vfe.get_output_feature_dim(), grid_size=ds.grid_size,
                        voxel_size=ds.voxel_size, point_cloud_range=ds.point_cloud_range)


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.vfe, self.backbone_3d = vfe, bb


net = Net()
sys.path.insert(0, os.path.join(rh.REF, "tools"))
from train_utils.optimization import fastai_optim  # noqa: E402

flatten_model = lambda m: sum(map(flatten_model, m.children()), []) if len(list(m.children())) else [m]  # noqa: E731
opt = fastai_optim.OptimWrapper.create(partial(torch.optim.Adam, betas=(0.9, 0.99)), 3e-3, [nn.Sequential(*flatten_model(net))],
                                       wd=0.01, true_wd=True, bn_wd=True)
ids = [{id(p) for p in g["params"]} for g in opt.opt.param_groups]
names = {id(p): n for n, p in net.named_parameters()}
out = {"optimised_non_bn": sorted(names[i] for i in ids[0]), "optimised_bn": sorted(names[i] for i in ids[1]),
       # order of the tensors inside the two torch.optim.Adam param groups (= index space of optimizer_state['state'])
       "group_order": [[names[id(p)] for p in g["params"]] for g in opt.opt.param_groups],
       "group_keys": sorted(k for k in opt.opt.state_dict()["param_groups"][0] if k != "params"),
       "skipped": sorted(n for n, p in net.named_parameters() if id(p) not in ids[0] | ids[1]),
       "numel_skipped": int(sum(p.numel() for p in net.parameters() if id(p) not in ids[0] | ids[1])),
       "numel_total": int(sum(p.numel() for p in net.parameters()))}
json.dump(out, open(os.path.join(HERE, "optimizer_params.json"), "w"), indent=0)
print({k: (len(v) if isinstance(v, list) else v) for k, v in out.items()})
