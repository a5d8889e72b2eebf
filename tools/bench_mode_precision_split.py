"""Where does the 16-bit mode's loss scatter on the small cases come from?  The bench-mode forward of a golden case with ONE part of the model
taken out of the autocast region (that part then runs this library's fp32 kernels on fp32 rows), over K masking-noise seeds against the
fp32 oracle: mean / standard deviation of the relative loss deviation per variant.  Diagnosis only: the module forwards are wrapped from
here, the product code is not touched.   python tools/bench_mode_precision_split.py [case] [--seeds 8]"""
import logging, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gd-mae_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
from helpers import load_case
from oracle import gdmae_oracle as orc
from gdmae_hip import configs, optim, decoder as gdec
from pcdet.models import build_network
from pcdet.utils.spconv_utils import replace_feature
K = int(sys.argv[sys.argv.index("--seeds") + 1]) if "--seeds" in sys.argv else 8
name = next((a for a in sys.argv[1:] if not a.startswith("--") and not a.isdigit()), "kitti_b2")
dev = torch.device("cuda:0")
torch.set_num_threads(32)
z, ds, cfg, shapes = load_case(name)
sd = orc.seeded_state_dict(shapes, seed=int(z["seed"]))
pts = torch.from_numpy(z["points"])
B, M = int(z["batch_size"]), int(z["noise"].shape[0])
torch.manual_seed(0)
net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev)
net.load_state_dict(sd, strict=False)
net.train()
opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)
bb = net.backbone_3d
FP32 = set()


def no_autocast(fn, tag, cast_in=None, cast_16=None):
    def wrapped(*a, **k):
        if tag in FP32:
            with torch.autocast("cuda", enabled=False):
                if cast_in is not None:
                    a, k = cast_in(a, k)
                return fn(*a, **k)
        if cast_16 is not None:          # a 16-bit part behind an fp32 part: hand it the bf16 rows its product path expects
            a, k = cast_16(a, k)
        return fn(*a, **k)
    return wrapped


def cast_sp(a, k):
    x = a[0]
    return (replace_feature(x, x.features.float()),) + tuple(a[1:]), k


def cast_hidden(a, k):          # sparse_decoder(model_cfg, deblocks, conv_out, hidden, ...)
    hid = [replace_feature(h, h.features.float()) for h in a[3]]
    return tuple(a[:3]) + (hid,) + tuple(a[4:]), k


net.vfe.forward = no_autocast(net.vfe.forward, "vfe")
for i, blk in enumerate(bb.sst_blocks):
    blk.forward = no_autocast(blk.forward, f"stage{i + 1}", cast_sp)
def cast_hidden16(a, k):        # (fp32 stage outputs would send the deconvolutions down the bf16-weight fallback path)
    hid = [replace_feature(h, h.features.bfloat16()) if h.features.dtype != torch.bfloat16 else h for h in a[3]]
    return tuple(a[:3]) + (hid,) + tuple(a[4:]), k


gdec.sparse_decoder = no_autocast(gdec.sparse_decoder, "decoder", cast_hidden, cast_hidden16)
# finer split inside the stages: the sparse-conv blocks (conv_down / conv_out: spconv + BatchNorm + ReLU) and the transformer layers
import pcdet.models.backbones_3d.spt_backbone as sb_mod
for i, blk in enumerate(bb.sst_blocks):
    for nm in ("conv_down", "conv_out"):
        m = getattr(blk, nm, None)
        if m is not None:
            m.forward = no_autocast(m.forward, "convs", cast_sp)


def cast_feat(a, k):           # encoder_stage(blocks, feat, table, wplans, residual=...)
    return (a[0], a[1].float()) + tuple(a[2:]), k


sb_mod.genc.encoder_stage = no_autocast(sb_mod.genc.encoder_stage, "layers", cast_feat)
import pcdet.models.backbones_3d.spt_backbone_mae as mae_mod
mae_mod.gdec = gdec

refs, noises = [], []
for s in range(K):
    noise = torch.rand(M, generator=torch.Generator().manual_seed(1000 + s))
    with torch.no_grad():
        o = orc.forward(pts, B, cfg, {k: v.clone() for k, v in sd.items()}, ds.point_cloud_range, ds.voxel_size, ds.grid_size, noise=noise)
    refs.append(float(o["loss"])), noises.append(noise)
variants = [("all 16-bit (the bench mode)", set()), ("DynVFE fp32", {"vfe"}), ("stage 1 fp32", {"stage1"}), ("stage 2 fp32", {"stage2"}),
            ("stage 3 fp32", {"stage3"}), ("all stages fp32", {"stage1", "stage2", "stage3"}), ("decoder fp32", {"decoder"}),
            ("DynVFE + stages fp32 (decoder 16-bit)", {"vfe", "stage1", "stage2", "stage3"}),
            ("stages + decoder fp32 (DynVFE 16-bit)", {"stage1", "stage2", "stage3", "decoder"}),
            ("sparse-conv blocks of all stages fp32", {"convs"}), ("transformer layers of all stages fp32", {"layers"}),
            ("everything fp32 under the bench-mode module tree", {"vfe", "stage1", "stage2", "stage3", "decoder"})]
for label, on in variants:
    FP32.clear(); FP32.update(on)
    dv = []
    try:
        for s in range(K):
            bd = {"points": pts.to(dev), "batch_size": B, "mae_noise": noises[s].to(dev)}
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                ret, _, _ = net(bd)
            dv.append((float(ret["loss"]) - refs[s]) / refs[s])
        a = np.array(dv)
        print(f"{name} M={M} {label:52s}: mean {a.mean():+.2e}  std {a.std():.2e}  max |.| {np.abs(a).max():.2e}", flush=True)
    except Exception as e:       # noqa: BLE001
        print(f"{name} {label}: FAILED {type(e).__name__}: {str(e)[:200]}", flush=True)
