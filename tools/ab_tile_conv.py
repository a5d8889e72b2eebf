"""A/B of the decoder conv implementations in bf16 mode: per-parameter gradient differences (diagnostic)."""
import logging, sys, os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "gd-mae_amd"), os.path.join(REPO, "tests")):
    sys.path.insert(0, p)
import torch
from helpers import load_case
from oracle import gdmae_oracle as orc
from pcdet.models import build_network
name = sys.argv[1] if len(sys.argv) > 1 else "kitti_b2_m75"
z, ds, cfg, shapes = load_case(name)
dev = torch.device("cuda:0")
res = {}
for impl in ("tiles", "dense", "fp32"):
    torch.manual_seed(0)
    net = build_network(cfg, 3, ds, logging.getLogger("t")).to(dev)
    net.load_state_dict(orc.seeded_state_dict(shapes, seed=5), strict=False)
    net.backbone_3d.decoder_conv_impl = impl if impl != "fp32" else "dense"
    net.train()
    bd = {"points": torch.from_numpy(z["points"]).to(dev), "batch_size": int(z["batch_size"]), "mae_noise": torch.from_numpy(z["noise"]).to(dev)}
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=impl != "fp32"):
        ret, _, _ = net(bd)
    ret["loss"].backward()
    res[impl] = (float(ret["loss"]), {k: p.grad.detach().double().cpu() for k, p in net.named_parameters()})
print("loss", {k: v[0] for k, v in res.items()})
gt, gd, gf = res["tiles"][1], res["dense"][1], res["fp32"][1]
rows = []
for k in gt:
    n = gf[k].norm() + 1e-30
    rows.append((float((gt[k] - gd[k]).norm() / n), float((gt[k] - gf[k]).norm() / n), float((gd[k] - gf[k]).norm() / n), k))
rows.sort(reverse=True)
print("rel |tiles-dense|, |tiles-fp32|, |dense-fp32|  (relative to the fp32 gradient norm)")
for r in rows[:25]:
    print("%.3e %.3e %.3e %s" % r)
import numpy as np
a = np.array([r[:3] for r in rows if not r[3].endswith("tau")])
print("median", np.median(a, 0), "max", a.max(0))
