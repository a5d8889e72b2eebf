"""Bench-mode (bf16) forward + backward on the reference goldens: loss / gradient-norm / tau deviations from the fp32 reference values.
Usage: python tools/bench_vs_golden.py [case ...]   (env GDMAE_PRED_F32=1: prediction head in torch fp32 - a probe, not a product path)"""
import logging
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gd-mae_amd"), os.path.join(ROOT, "tests")]
from helpers import load_case                    # noqa: E402
from oracle import gdmae_oracle as orc           # noqa: E402  (checker side only: seeded weights of the goldens)


def run(name):
    from gdmae_hip import configs, optim
    from pcdet.models import build_network
    dev = torch.device("cuda:0")
    z, ds, cfg, shapes = load_case(name)
    torch.manual_seed(0)
    net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev)
    net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
    net.train()
    opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)
    opt.zero_grad()
    bd = {"points": torch.from_numpy(z["points"]).to(dev), "batch_size": int(z["batch_size"]), "mae_noise": torch.from_numpy(z["noise"]).to(dev)}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ret, _, _ = net(bd)
    ret["loss"].backward()
    names = sorted(shapes)
    params = dict(net.named_parameters())
    gn = np.array([float(params[k].grad.double().norm()) for k in names])
    rel = np.abs(gn - z["grad_norm"]) / (z["grad_norm"] + 1e-12)
    nt = np.array([not k.endswith("tau") for k in names])
    tau_ref = z["grad_norm"][~nt]
    tau_abs = np.abs(gn[~nt] - tau_ref).max() / tau_ref.max()
    lrel = abs(float(ret["loss"]) - float(z["loss"])) / float(z["loss"])
    worst = names[int(np.argmax(np.where(nt, rel, 0)))]
    print(f"{name}: loss rel {lrel:.3e}  worst grad-norm dev {rel[nt].max():.3e} ({worst})  tau | |g| - |g_ref| | / max|g_ref| {tau_abs:.3e}", flush=True)


if __name__ == "__main__":
    for n in (sys.argv[1:] or ["kitti_b2", "kitti_b2_m75", "waymo_b1", "once_e_b1"]):
        run(n)
