"""Per-phase time line of one workgroup of k_layer_fwd_v3 (variant library built with -DV3_TRACE=<workgroup>): runs a few bench steps and
prints the stamps of the last launch.  Usage (GPU box): GDMAE_LIB=gd-mae_amd/csrc/variants/lib_v3trace.so GDMAE_LAYER_V3=1 python tools/v3_trace.py"""
import ctypes as C, os, sys, logging
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from gdmae_hip import lib as L
if os.environ.get("GDMAE_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["GDMAE_LIB"])
from gdmae_hip import configs, optim, synth
from pcdet.models import build_network
dev = torch.device("cuda:0")
NB = 8
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
torch.manual_seed(1234)
net = build_network(cfg, 3, ds, logging.getLogger("p")).to(dev).train()
net.backbone_3d.dense_spatial_features = False
net.sync_loss_scalar = False
b = torch.from_numpy(synth.synth_batch(5, NB, ds.point_cloud_range, **skw)).to(dev)
for it in range(3):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ret, _, _ = net({"points": b, "batch_size": NB})
torch.cuda.synchronize()
buf = (C.c_ulonglong * 256)()
lib = C.CDLL(L.LIB_PATH)
assert lib.gdmae_debug_v3_trace(buf) == 0
print("raw", [buf[i] for i in range(8)], L.LIB_PATH, os.environ.get("GDMAE_LAYER_V3"))
for w in range(4):
    t = [buf[w * 64 + i] for i in range(64)]
    t0 = t[0]
    print("wave", w, " ".join("%d:%.2f" % (i, (t[i] - t0) / 100.0) for i in range(64) if t[i]))
