"""Fine-tune head stack (SURVEY next row f1, BASELINE config D): SSTBEVBackbone + CenterHead + CenterPoint against the golden
captured from the UNMODIFIED reference modules (tests/golden/make_golden_head.py), and the config-D training step."""
import logging
import os

import numpy as np
import pytest
import torch

from gdmae_hip import configs
from helpers import GOLDEN, assert_sampled_close, seeded_head_state


def _golden():
    return dict(np.load(os.path.join(GOLDEN, "center_head_b2.npz")))


def _modules(z):
    from pcdet.models.backbones_2d import SSTBEVBackbone
    from pcdet.models.dense_heads import CenterHead
    b2d = SSTBEVBackbone(model_cfg=configs.sst_bev_backbone_cfg(), input_channels=128)
    head = CenterHead(model_cfg=configs.center_head_cfg(), input_channels=b2d.num_bev_features, num_class=3,
                      class_names=['Vehicle', 'Pedestrian', 'Cyclist'], grid_size=z["grid_size"], point_cloud_range=z["point_cloud_range"],
                      voxel_size=[float(v) for v in z["voxel_size"]], predict_boxes_when_training=False)

    class Wrap(torch.nn.Module):
        def __init__(s):
            super().__init__()
            s.backbone_2d, s.dense_head = b2d, head
    return Wrap()


def test_head_stack_registries_and_state_dict_match_reference():
    """CPU: registry names and every state_dict key / shape of the BEV backbone + head equal the reference's (golden)."""
    from pcdet.models import backbones_2d, dense_heads, detectors
    assert set(backbones_2d.__all__) >= {"SSTBEVBackbone"} and set(dense_heads.__all__) >= {"CenterHead"}
    assert set(detectors.__all__) >= {"GDMAE", "CenterPoint"}
    z = _golden()
    net = _modules(z)
    ours = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    ref = {str(k): tuple(int(x) for x in s if x) for k, s in zip(z["state_keys"], z["state_shapes"])}
    assert ours == ref
    assert list(net.state_dict().keys()) == [k for k in net.state_dict().keys()] and sorted(ours) == [str(k) for k in z["state_keys"]]
    cfg, ds, _ = configs.named_config("D")
    from pcdet.models import build_network
    det = build_network(cfg, 3, ds, logging.getLogger("t"))
    assert [type(m).__name__ for m in det.module_list] == ["DynVFE", "SPTBackbone", "SSTBEVBackbone", "CenterHead"]
    assert tuple(int(v) for v in ds.grid_size) == (432, 496, 1)


@pytest.mark.gpu
def test_center_head_targets_match_reference_golden():
    """gdmae_center_head_targets vs the reference's CPU loop: inds / masks exact, heat map and regression targets to fp32
    round-off (device exp / log / sincos vs the host's), incl. a box on the map border, a box outside the range (clamped),
    two boxes sharing a cell, and padding rows."""
    z = _golden()
    dev = torch.device("cuda:0")
    net = _modules(z).to(dev)
    H, W = int(z["grid_size"][1]), int(z["grid_size"][0])
    t = net.dense_head.assign_targets(torch.from_numpy(z["gt_boxes"]).to(dev), feature_map_size=(H, W))
    assert np.array_equal(t["inds"][0].cpu().numpy(), z["inds"])
    assert np.array_equal(t["masks"][0].cpu().numpy(), z["masks"])
    assert int(z["masks"].sum()) >= 30
    hm = t["heatmaps"][0].cpu().numpy()
    assert hm.shape == z["heatmap"].shape and np.abs(hm - z["heatmap"]).max() <= 1e-6
    assert np.array_equal(hm == 1, z["heatmap"] == 1)                               # the positives of the focal loss
    assert np.abs(t["target_boxes"][0].cpu().numpy() - z["target_boxes"]).max() <= 2e-6
    # empty input and more boxes than slots
    e = net.dense_head.assign_targets(torch.zeros(2, 5, 8, device=dev), feature_map_size=(H, W))
    assert float(e["heatmaps"][0].abs().max()) == 0 and int(e["masks"][0].sum()) == 0
    many = torch.from_numpy(z["gt_boxes"]).to(dev).repeat(1, 30, 1)                 # 720 boxes > NUM_MAX_OBJS = 500
    m = net.dense_head.assign_targets(many, feature_map_size=(H, W))
    valid = (many[..., 7] > 0).sum(1).clamp(max=500)
    assert torch.all(m["masks"][0].sum(1) <= valid) and int(m["masks"][0].sum()) > 600


@pytest.mark.gpu
def test_bev_backbone_and_center_head_match_reference_golden():
    """Forward, loss and gradients of SSTBEVBackbone + CenterHead (fp32) vs the reference on the same seeded weights, BEV
    map and boxes: loss terms 1e-4, feature / prediction samples 5e-4, parameter-gradient norms 5e-3."""
    z = _golden()
    dev = torch.device("cuda:0")
    net = _modules(z)
    net.load_state_dict(seeded_head_state(net, int(z["seed"])), strict=False)
    net = net.to(dev).train()
    B, H, W = int(z["batch_size"]), int(z["grid_size"][1]), int(z["grid_size"][0])
    sf = (torch.randn(B, 128, H, W, generator=torch.Generator().manual_seed(int(z["seed"]) + 1)) * 0.5).to(dev).requires_grad_(True)
    torch.backends.cudnn.allow_tf32 = False
    dd = net.dense_head(net.backbone_2d({"spatial_features": sf, "gt_boxes": torch.from_numpy(z["gt_boxes"]).to(dev), "batch_size": B}))
    loss, tb = net.dense_head.get_loss()
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) <= 1e-4 * float(z["loss"])
    assert abs(float(tb["hm_loss_head_0"]) - float(z["hm_loss"])) <= 1e-4 * float(z["hm_loss"])
    assert abs(float(tb["loc_loss_head_0"]) - float(z["loc_loss"])) <= 1e-4 * float(z["loc_loss"])
    assert_sampled_close(dd["spatial_features_2d"], z["feat2d_s"], z["feat2d_c"], 5e-4, "spatial_features_2d")
    pd = net.dense_head.forward_ret_dict["pred_dicts"][0]
    for k in ("center", "center_z", "dim", "rot"):
        assert_sampled_close(pd[k], z[f"pred_{k}_s"], z[f"pred_{k}_c"], 5e-4, k)
    assert_sampled_close(sf.grad, z["sf_grad_s"], z["sf_grad_c"], 2e-3, "input gradient")
    gp = dict(net.named_parameters())
    gn = np.array([float(gp[str(k)].grad.double().norm()) for k in z["param_names"]])
    # (conv biases in front of a BatchNorm have an exactly-zero mathematical gradient: both sides hold round-off noise there,
    #  hence the absolute slack relative to the largest gradient)
    err = np.abs(gn - z["grad_norm"])
    tol = 5e-3 * z["grad_norm"] + 1e-6 * z["grad_norm"].max()
    assert (err <= tol).all(), [(str(z["param_names"][i]), gn[i], z["grad_norm"][i]) for i in np.flatnonzero(err > tol)]


@pytest.mark.gpu
@pytest.mark.parametrize("autocast", [False, True])
def test_config_d_training_step(autocast):
    """BASELINE config D (KITTI-shape 20 k points, 0.16 m pillars, SPTBackbone + SSTBEVBackbone + CenterHead / CenterPoint)
    through build_network and model_fn_decorator: finite loss, every parameter receives a finite gradient, the flat
    optimizer steps, and the step is repeatable bit for bit."""
    from gdmae_hip import optim, synth
    from pcdet.models import build_network, model_fn_decorator
    from tests_golden_boxes import synth_boxes
    dev = torch.device("cuda:0")
    cfg, ds, skw = configs.named_config("D")
    B = 2
    pts = synth.synth_batch(77, B, ds.point_cloud_range, **skw)
    gt = synth_boxes(np.random.default_rng(5), B, 20, np.asarray(ds.point_cloud_range), 3)
    res = []
    for rep in range(2):
        torch.manual_seed(3)
        net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev).train()
        opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(B), total_steps=10)
        opt.zero_grad()
        model_func = model_fn_decorator()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            ret = model_func(net, {"points": pts.copy(), "gt_boxes": gt.copy(), "batch_size": B})
        assert int(net.global_step) == 1 and set(ret.tb_dict) >= {"loss_rpn", "hm_loss_head_0", "loc_loss_head_0"}
        ret.loss.backward()
        missing = [k for k, p in net.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
        assert not missing, missing
        assert torch.isfinite(ret.loss) and float(ret.loss) > 0
        zero = [k for k, p in net.named_parameters() if float(p.grad.abs().max()) == 0]
        assert not zero, zero
        before = opt.flat_param.clone()
        opt.step(0)
        assert torch.isfinite(opt.flat_param).all() and not torch.equal(before, opt.flat_param)
        res.append((float(ret.loss), opt.flat_grad.clone()))
    assert res[0][0] == res[1][0] or autocast            # fp32: deterministic kernels -> identical loss
