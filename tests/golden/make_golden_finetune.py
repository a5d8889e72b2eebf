"""Golden for the fine-tune backbone (SURVEY next row f1, encoder part): reference DynVFE + SPTBackbone (no masking, all
pillars are tokens; pcdet/models/backbones_3d/spt_backbone.py:267-347) forward + backward on a seeded KITTI-shape batch,
run from the unmodified reference modules through ref_harness.  Run in the build container."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd"), HERE]
import make_golden as mg  # noqa: E402  (installs the reference namespace + this repo's pcdet.config for gdmae_hip.configs)

rh, configs, synth, orc = mg.rh, mg.configs, mg.synth, mg.orc


def main():
    seed, B, F = 21, 2, 4
    ycfg = rh.load_yaml_cfg("cfgs/kitti_models/gd_mae.yaml")
    mc = ycfg.MODEL
    ours = configs.gdmae_finetune_backbone_cfg()
    assert mg.to_plain(mc.BACKBONE_3D) == mg.to_plain(ours), "fine-tune backbone config drifted from the reference yaml"
    ds = configs.SyntheticDatasetInfo(**configs.KITTI)
    points = torch.from_numpy(synth.synth_batch(seed, B, ds.point_cloud_range, beams=24, azimuths=300, extra=500, features=F))
    dyn = rh.ref("pcdet.models.backbones_3d.vfe.dyn_vfe")
    spt = rh.ref("pcdet.models.backbones_3d.spt_backbone")
    vfe = dyn.DynVFE(model_cfg=mc.VFE, num_point_features=F, voxel_size=ds.voxel_size, point_cloud_range=ds.point_cloud_range,
                     grid_size=ds.grid_size)
    bb = spt.SPTBackbone(model_cfg=mc.BACKBONE_3D, input_channels=vfe.get_output_feature_dim(), grid_size=ds.grid_size,
                         voxel_size=ds.voxel_size, point_cloud_range=ds.point_cloud_range)

    class Wrap(torch.nn.Module):
        def __init__(s):
            super().__init__()
            s.vfe, s.backbone_3d = vfe, bb
    net = Wrap()
    shapes = {k: tuple(v.shape) for k, v in net.named_parameters()}
    sd = orc.seeded_state_dict(shapes, seed=seed)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(("running_" in m or "num_batches" in m) for m in missing)
    net.train()
    bd = bb(vfe({"points": points.clone(), "batch_size": B}))
    sf = bd["spatial_features"]
    wgt = torch.randn(sf.shape, generator=torch.Generator().manual_seed(seed + 1))
    loss = (sf * wgt).sum() / sf.numel()
    loss.backward()
    names = sorted(shapes)
    g = dict(net.named_parameters())
    z = {"seed": np.int64(seed), "batch_size": np.int64(B), "num_point_features": np.int64(F), "points": points.numpy(),
         "voxel_coords": bd["voxel_coords"].numpy().astype(np.int32), "loss": np.float64(float(loss)),
         "param_names": np.array(names), "param_shapes": np.array([list(shapes[k]) + [0] * (4 - len(shapes[k])) for k in names]),
         "grad_norm": np.array([float(g[k].grad.double().norm()) for k in names])}
    z["spatial_features_s"], z["spatial_features_c"] = mg.sample(sf)
    for i in range(3):
        t = bd["multi_scale_3d_features"][f"x_conv{i + 1}"]
        z[f"st{i}_indices"] = t.indices.numpy().astype(np.int32)
        z[f"st{i}_features_s"], z[f"st{i}_features_c"] = mg.sample(t.features)
    path = os.path.join(HERE, "finetune_kitti_b2.npz")
    np.savez_compressed(path, **z)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; M =", bd["voxel_coords"].shape[0], "tokens per stage",
          [int(bd["multi_scale_3d_features"][f"x_conv{i + 1}"].features.shape[0]) for i in range(3)], "loss", float(loss))


if __name__ == "__main__":
    main()
