"""Kernel timing of the first DynVFE point layer (gdmae_vfe_point_layer_fwd / _bwd) on a config-B batch; prints the
forward / backward call time and the effective HBM rate against the layer's algorithmic bytes."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch  # noqa: E402

from gdmae_hip import configs, plan as gplan, synth, vfe as gvfe  # noqa: E402

dev = torch.device("cuda:0")
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
B = 8
pts = torch.from_numpy(synth.synth_batch(0, B, ds.point_cloud_range, **skw)).to(dev)
vfe_cfg = cfg.VFE if hasattr(cfg, "VFE") else cfg.MODEL.VFE
vox = gplan.voxelize(pts, ds.point_cloud_range, ds.voxel_size, ds.grid_size, B)
N, D = int(vox.N), vox.n_cols + 5
torch.manual_seed(0)
W = (torch.randn(64, D, device=dev) * 0.1).requires_grad_()
gamma = torch.ones(64, device=dev, requires_grad=True)
beta = torch.zeros(64, device=dev, requires_grad=True)
res = {"points": N}
for bf in (True, False):
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf):
        y, _, _ = gvfe.PointLayer1.apply(vox, W, gamma, beta, 1e-3, None)
    g = torch.randn_like(y)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for i in range(13):
        for p in (W, gamma, beta):
            p.grad = None
        ev[0].record()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf):
            y, _, _ = gvfe.PointLayer1.apply(vox, W, gamma, beta, 1e-3, None)
        ev[1].record()
        y.backward(g)
        ev[2].record()
        torch.cuda.synchronize()
        if i >= 3:
            tf += ev[0].elapsed_time(ev[1]) * 100
            tb += ev[1].elapsed_time(ev[2]) * 100
    es = 2 if bf else 4
    geo = N * (4 * vox.n_cols + 32 + 4)                     # points + coords + inverse (pillar means stay in cache)
    res["bf16" if bf else "fp32"] = {"fwd_us": round(tf, 1), "bwd_us": round(tb, 1),
                                     "fwd_GBs": round((2 * geo + N * 64 * es) / tf / 1e3, 1),
                                     "bwd_GBs": round((2 * geo + 2 * N * 64 * es) / tb / 1e3, 1)}
print(json.dumps(res))
