"""End-to-end golden of the fine-tune detector (BASELINE config D's module chain; SURVEY next row f1): the reference's DynVFE ->
SPTBackbone -> SSTBEVBackbone -> CenterHead (pcdet/models/backbones_3d/vfe/dyn_vfe.py, backbones_3d/spt_backbone.py:267-347,
backbones_2d/sst_bev_backbone.py:6-42, dense_heads/center_head.py:48-392) run UNMODIFIED through ref_harness on one seeded
KITTI-shape batch with seeded ground-truth boxes: forward, CenterHead loss (focal heat map + L1 regression), backward through the
whole chain.  Build container only (reads /root/reference); the component goldens (finetune_kitti_b2: backbone alone,
center_head_b2: BEV backbone + head on a random map) pin the two halves, this one pins their composition."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd"), HERE, os.path.join(REPO, "tests")]
import make_golden as mg  # noqa: E402
import make_golden_head as mh  # noqa: E402

rh, configs, synth, orc = mg.rh, mg.configs, mg.synth, mg.orc


def main():
    mh.install_head_stubs()
    from head_seed import seeded_head_state
    seed, B, F = 41, 2, 4
    y3 = rh.load_yaml_cfg("cfgs/kitti_models/gd_mae.yaml").MODEL
    y2 = rh.load_yaml_cfg("cfgs/waymo_models/gd_mae.yaml").MODEL          # BACKBONE_2D / CenterHead sections (config D: BASELINE-defined)
    ds = configs.SyntheticDatasetInfo(**configs.KITTI)
    class_names = ['Vehicle', 'Pedestrian', 'Cyclist']
    pcr = np.asarray(ds.point_cloud_range, dtype=np.float32)
    points = torch.from_numpy(synth.synth_batch(seed, B, ds.point_cloud_range, beams=24, azimuths=300, extra=500, features=F))
    dyn = rh.ref("pcdet.models.backbones_3d.vfe.dyn_vfe")
    spt = rh.ref("pcdet.models.backbones_3d.spt_backbone")
    bev = rh.ref("pcdet.models.backbones_2d.sst_bev_backbone")
    ch = rh.ref("pcdet.models.dense_heads.center_head")
    vfe = dyn.DynVFE(model_cfg=y3.VFE, num_point_features=F, voxel_size=ds.voxel_size, point_cloud_range=ds.point_cloud_range,
                     grid_size=ds.grid_size)
    bb = spt.SPTBackbone(model_cfg=y3.BACKBONE_3D, input_channels=vfe.get_output_feature_dim(), grid_size=ds.grid_size,
                         voxel_size=ds.voxel_size, point_cloud_range=ds.point_cloud_range)
    b2d = bev.SSTBEVBackbone(model_cfg=y2.BACKBONE_2D, input_channels=128)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        head = ch.CenterHead(model_cfg=y2.DENSE_HEAD, input_channels=b2d.num_bev_features, num_class=3, class_names=class_names,
                             grid_size=np.asarray(ds.grid_size), point_cloud_range=pcr, voxel_size=list(ds.voxel_size),
                             predict_boxes_when_training=False)
    finally:
        torch.Tensor.cuda = real_cuda

    class Front(torch.nn.Module):
        def __init__(s):
            super().__init__()
            s.vfe, s.backbone_3d = vfe, bb

    class Back(torch.nn.Module):
        def __init__(s):
            super().__init__()
            s.backbone_2d, s.dense_head = b2d, head
    front, back = Front(), Back()
    shapes = {k: tuple(v.shape) for k, v in front.named_parameters()}
    front.load_state_dict(orc.seeded_state_dict(shapes, seed=seed), strict=False)
    back.load_state_dict(seeded_head_state(back, seed), strict=False)
    front.train(), back.train()
    rng = np.random.default_rng(seed)
    gt = torch.from_numpy(mh.synth_boxes(rng, B, 24, pcr, 3))
    bd = bb(vfe({"points": points.clone(), "batch_size": B}))
    bd["gt_boxes"] = gt.clone()
    bd = head(b2d(bd))
    loss, tb = head.get_loss()
    loss.backward()
    g = {**{k: v for k, v in front.named_parameters()}, **{k: v for k, v in back.named_parameters()}}
    names = sorted(g)
    z = {"seed": np.int64(seed), "batch_size": np.int64(B), "num_point_features": np.int64(F), "points": points.numpy(),
         "gt_boxes": gt.numpy(), "loss": np.float64(float(loss)), "hm_loss": np.float64(tb["hm_loss_head_0"]),
         "loc_loss": np.float64(tb["loc_loss_head_0"]), "voxel_coords": bd["voxel_coords"].numpy().astype(np.int32),
         "front_names": np.array(sorted(shapes)),
         "front_shapes": np.array([list(shapes[k]) + [0] * (4 - len(shapes[k])) for k in sorted(shapes)]),
         "param_names": np.array(names), "grad_norm": np.array([float(g[k].grad.double().norm()) for k in names])}
    z["spatial_features_s"], z["spatial_features_c"] = mg.sample(bd["spatial_features"])
    z["feat2d_s"], z["feat2d_c"] = mg.sample(bd["spatial_features_2d"])
    path = os.path.join(HERE, "detector_kitti_b2.npz")
    np.savez_compressed(path, **z)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB; pillars", bd["voxel_coords"].shape[0], "loss", float(loss), tb,
          "positives", int(head.forward_ret_dict["target_dicts"]["masks"][0].sum()))


if __name__ == "__main__":
    main()
