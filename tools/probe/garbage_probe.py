"""Does any kernel of the bench step read memory it did not write?  Runs the kitti_b2_m75 golden step (native / op-by-op conv block)
twice: from fresh allocations, and after the caching allocator's free blocks were filled with NaN / huge values."""
import logging, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd"), os.path.join(REPO, "tests")]
import numpy as np, torch
from helpers import load_case
from oracle import gdmae_oracle as orc
from gdmae_hip import configs, optim
from pcdet.models import build_network
from pcdet.utils.spconv_utils import SparseSequential
dev = torch.device("cuda:0")
z, ds, cfg, shapes = load_case(sys.argv[1] if len(sys.argv) > 1 else "kitti_b2_m75")
def run(native):
    SparseSequential.native_block = native
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
    torch.cuda.synchronize()
    return float(ret["loss"]), opt.flat_grad.clone()
def poison(val):
    bufs = [torch.full((n,), val, dtype=torch.float32, device=dev) for n in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 1 << 18, 1 << 16) for _ in range(3)]
    torch.cuda.synchronize()
    del bufs
base = {n: run(n) for n in (True, False)}
print("fresh      :", {k: v[0] for k, v in base.items()}, "grad diff native vs op-by-op", float((base[True][1] - base[False][1]).norm() / base[False][1].norm()))
for val in (float("nan"), 3e38, -7.0):
    poison(val)
    r = {n: run(n) for n in (True, False)}
    print(f"poison {val}:", {k: v[0] for k, v in r.items()}, {k: bool(torch.equal(r[k][1], base[k][1])) for k in r},
          {k: float((r[k][1] - base[k][1]).norm() / base[k][1].norm()) for k in r})
