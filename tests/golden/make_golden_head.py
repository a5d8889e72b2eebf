"""Golden for the fine-tune head stack (SURVEY next row f1): the reference's SSTBEVBackbone
(pcdet/models/backbones_2d/sst_bev_backbone.py:6-42) + CenterHead (pcdet/models/dense_heads/center_head.py:48-392:
shared conv, separate heads, Gaussian target assignment, focal + L1 losses) run UNMODIFIED on a seeded dense BEV map and
seeded ground-truth boxes, forward + loss + backward.  Build container only (reads /root/reference).

Stand-ins (never executed by the training-loss path that is captured): ``numba`` (decorator no-op; only circle_nms uses it),
``pcdet.ops.iou3d_nms.iou3d_nms_utils`` and ``pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils`` (CUDA extensions, imported
by model_nms_utils / loss_utils / box_utils at module level), and ``Tensor.cuda`` as the identity while the head is
constructed (center_head.py:67 moves an index tensor to the GPU in __init__)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd"), HERE]
import make_golden as mg  # noqa: E402

rh, configs = mg.rh, mg.configs


def install_head_stubs():
    for name in ["pcdet.models.dense_heads", "pcdet.models.backbones_2d", "pcdet.ops.iou3d_nms", "pcdet.ops.roiaware_pool3d"]:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [os.path.join(rh.REF, *name.split("."))]
            sys.modules[name] = m
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    sys.modules["numba"] = nb
    for name in ["pcdet.ops.iou3d_nms.iou3d_nms_utils", "pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils"]:
        sys.modules[name] = types.ModuleType(name)
    sys.modules["pcdet.ops.iou3d_nms"].iou3d_nms_utils = sys.modules["pcdet.ops.iou3d_nms.iou3d_nms_utils"]
    sys.modules["pcdet.ops.roiaware_pool3d"].roiaware_pool3d_utils = sys.modules["pcdet.ops.roiaware_pool3d.roiaware_pool3d_utils"]


def synth_boxes(rng, B, n_max, pcr, n_class):
    """(B, n_max, 8) [x, y, z, dx, dy, dz, heading, class 1..n_class], zero rows = padding (class 0), as the reference's
    collate_batch pads gt_boxes (dataset.py:188-193)."""
    out = np.zeros((B, n_max, 8), dtype=np.float32)
    for b in range(B):
        n = int(rng.integers(n_max // 2, n_max - 1))
        out[b, :n, 0] = rng.uniform(pcr[0] + 1, pcr[3] - 1, n)
        out[b, :n, 1] = rng.uniform(pcr[1] + 1, pcr[4] - 1, n)
        out[b, :n, 2] = rng.uniform(-1.5, 0.5, n)
        cls = rng.integers(1, n_class + 1, n)
        size = {1: (4.2, 1.8, 1.6), 2: (0.8, 0.7, 1.7), 3: (1.8, 0.7, 1.6)}
        for i in range(n):
            s = size[min(int(cls[i]), 3)]
            out[b, i, 3:6] = np.array(s) * rng.uniform(0.8, 1.3, 3)
        out[b, :n, 6] = rng.uniform(-np.pi, np.pi, n)
        out[b, :n, 7] = cls
        # edge cases: a box on the map border, one outside the range (clamped by the assigner), two boxes in one cell
        out[b, 0, 0:2] = [pcr[3] - 0.01, pcr[4] - 0.01]
        out[b, 1, 0:2] = [pcr[0] - 3.0, pcr[1] + 5.0]
        out[b, 3, 0:2] = out[b, 2, 0:2] + 0.05
    return out


def main():
    install_head_stubs()
    seed, B = 31, 2
    ycfg = rh.load_yaml_cfg("cfgs/waymo_models/gd_mae.yaml")
    mc = ycfg.MODEL
    ours_b2d, ours_head = configs.sst_bev_backbone_cfg(), configs.center_head_cfg()
    assert mg.to_plain(mc.BACKBONE_2D) == mg.to_plain(ours_b2d), "BACKBONE_2D config drifted from the reference yaml"
    assert mg.to_plain(mc.DENSE_HEAD) == mg.to_plain(ours_head), "DENSE_HEAD config drifted from the reference yaml"
    # config D geometry: KITTI range at 0.16 m pillars (432 x 496), a 1/4-size crop of it keeps the fixture small
    pcr = np.array([0, -19.84, -3, 34.56, 19.84, 1], dtype=np.float32)
    vs = [0.16, 0.16, 4]
    grid = np.round((pcr[3:6] - pcr[0:3]) / np.array(vs)).astype(np.int64)      # 216 x 248 x 1
    class_names = ['Vehicle', 'Pedestrian', 'Cyclist']
    bev = rh.ref("pcdet.models.backbones_2d.sst_bev_backbone")
    ch = rh.ref("pcdet.models.dense_heads.center_head")
    b2d = bev.SSTBEVBackbone(model_cfg=mc.BACKBONE_2D, input_channels=128)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        head = ch.CenterHead(model_cfg=mc.DENSE_HEAD, input_channels=b2d.num_bev_features, num_class=3, class_names=class_names,
                             grid_size=grid, point_cloud_range=pcr, voxel_size=vs, predict_boxes_when_training=False)
    finally:
        torch.Tensor.cuda = real_cuda

    class Wrap(torch.nn.Module):
        def __init__(s):
            super().__init__()
            s.backbone_2d, s.dense_head = b2d, head
    net = Wrap()
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from head_seed import seeded_head_state
    sd = seeded_head_state(net, seed)
    g = torch.Generator().manual_seed(seed + 1)
    net.load_state_dict(sd, strict=False)
    net.train()
    H, W = int(grid[1]), int(grid[0])
    sf = torch.randn(B, 128, H, W, generator=g) * 0.5
    sf.requires_grad_(True)
    rng = np.random.default_rng(seed)
    gt = torch.from_numpy(synth_boxes(rng, B, 24, pcr, 3))
    dd = {"spatial_features": sf, "gt_boxes": gt.clone(), "batch_size": B}
    dd = head(b2d(dd))
    td = head.forward_ret_dict["target_dicts"]
    loss, tb = head.get_loss()
    loss.backward()
    pd = head.forward_ret_dict["pred_dicts"][0]
    names = sorted(k for k, _ in net.named_parameters())
    gp = dict(net.named_parameters())
    z = {"seed": np.int64(seed), "batch_size": np.int64(B), "point_cloud_range": pcr, "voxel_size": np.array(vs, dtype=np.float64),
         "grid_size": grid, "gt_boxes": gt.numpy(), "loss": np.float64(float(loss)),
         "hm_loss": np.float64(tb["hm_loss_head_0"]), "loc_loss": np.float64(tb["loc_loss_head_0"]),
         "heatmap": td["heatmaps"][0].numpy().astype(np.float32), "target_boxes": td["target_boxes"][0].numpy(),
         "inds": td["inds"][0].numpy().astype(np.int64), "masks": td["masks"][0].numpy().astype(np.int64),
         "param_names": np.array(names), "grad_norm": np.array([float(gp[k].grad.double().norm()) for k in names]),
         "state_keys": np.array(sorted(net.state_dict().keys())),
         "state_shapes": np.array([list(net.state_dict()[k].shape) + [0] * (4 - net.state_dict()[k].dim()) for k in sorted(net.state_dict().keys())])}
    z["sf_grad_s"], z["sf_grad_c"] = mg.sample(sf.grad)
    z["feat2d_s"], z["feat2d_c"] = mg.sample(dd["spatial_features_2d"])
    for k in ("hm", "center", "center_z", "dim", "rot"):
        z[f"pred_{k}_s"], z[f"pred_{k}_c"] = mg.sample(pd[k])
    path = os.path.join(HERE, "center_head_b2.npz")
    np.savez_compressed(path, **z)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB  loss", float(loss), tb, "positives", int(td["masks"][0].sum()))


if __name__ == "__main__":
    main()
