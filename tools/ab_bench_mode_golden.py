"""Bench-mode (bf16) loss / gradient norms of a golden case vs the reference golden: what an A/B switch (env) moves.  Usage: python tools/ab_bench_mode_golden.py kitti_b2"""
import logging, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd"), os.path.join(REPO, "tests")]
import numpy as np, torch
from helpers import load_case
from oracle import gdmae_oracle as orc
from gdmae_hip import configs, optim
from pcdet.models import build_network
name = sys.argv[1] if len(sys.argv) > 1 else "kitti_b2"
z, ds, cfg, shapes = load_case(name)
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev)
net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
net.train()
opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)
opt.zero_grad()
bd = {"points": torch.from_numpy(z["points"]).to(dev), "batch_size": int(z["batch_size"]), "mae_noise": torch.from_numpy(z["noise"]).to(dev)}
with torch.autocast("cuda", dtype=torch.bfloat16):
    ret, _, _ = net(bd)
loss_rel = abs(float(ret["loss"].detach()) - float(z["loss"])) / float(z["loss"])
ret["loss"].backward()
names = sorted(shapes)
params = dict(net.named_parameters())
gn = np.array([float(params[k].grad.double().norm()) for k in names])
rel = np.abs(gn - z["grad_norm"]) / (z["grad_norm"] + 1e-12)
nt = np.array([not k.endswith("tau") for k in names])
order = np.argsort(-rel * nt)[:4]
print(f"{name}: loss {float(ret['loss']):.6f} rel {loss_rel:.3e}; worst norm deviations: " + ", ".join(f"{names[i].split('.')[-3:]} {rel[i]:.3f}" for i in order))
