"""CPU experiment behind DESIGN section 5 (bench mode vs reference loss): the fp32 ORACLE with bf16 rounding injected at one decoder tensor at
a time - which rounding point moves the Chamfer loss by how much (relative to the fp32 loss).  python tools/oracle_rounding_injection.py [case]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gd-mae_amd"), os.path.join(ROOT, "tests")]
import torch, torch.nn.functional as F
from helpers import load_case
from oracle import gdmae_oracle as orc
torch.set_num_threads(16)
def r(x): return x.to(torch.bfloat16).float()
name = sys.argv[1] if len(sys.argv) > 1 else "kitti_b2"
z, ds, cfg, shapes = load_case(name)
sd = orc.seeded_state_dict(shapes, seed=int(z["seed"]))
pts = torch.from_numpy(z["points"]); noise = torch.from_numpy(z["noise"])
def run():
    with torch.no_grad():
        o = orc.forward(pts, int(z["batch_size"]), cfg, sd, ds.point_cloud_range, ds.voxel_size, ds.grid_size, noise=noise)
    return float(o["loss"])
base = run()
print("base", base, "golden", float(z["loss"]))
conv2d, relu, linear, convT = F.conv2d, F.relu, F.linear, F.conv_transpose2d
# 1: conv_out output rounded
F.conv2d = lambda x, w, **k: r(conv2d(x, w, **k)); l = run(); F.conv2d = conv2d
print("conv_out OUTPUT bf16:", (l - base) / base)
F.conv2d = lambda x, w, **k: conv2d(r(x), r(w), **k); l = run(); F.conv2d = conv2d
print("conv_out OPERANDS bf16:", (l - base) / base)
F.conv_transpose2d = lambda x, w, **k: r(convT(r(x), r(w), **k)); l = run(); F.conv_transpose2d = convT
print("deconv operands+output bf16:", (l - base) / base)
F.linear = lambda x, w, b=None: linear(x, w, b) if w.shape[0] != 48 else r(linear(r(x), r(w), b)); l = run(); F.linear = linear
print("pred head operands+output bf16:", (l - base) / base)
F.linear = lambda x, w, b=None: linear(x, w, b) if w.shape[0] != 48 else linear(r(x), w, b); l = run(); F.linear = linear
print("pred head INPUT rows bf16:", (l - base) / base)
