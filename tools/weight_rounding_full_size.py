"""GPU experiment behind DESIGN section 5 (why the bf16 mode sits where it sits against the fp32 mode at full size): config B, 8 frames,
HIP fp32 parity mode with the weight matrices of one group rounded to bf16 / fp16 in place - the SYSTEMATIC part of a 16-bit mode's
deviation (the same weight error at every site: it does not average over the pillars the way activation rounding does) - next to the bench
mode's own loss.  python tools/weight_rounding_full_size.py [--frames 8] [--seed 7]"""
import argparse, logging, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "gd-mae_amd")]
import torch
from gdmae_hip import configs, optim, synth, plan as gplan
from pcdet.models import build_network
ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=8)
ap.add_argument("--seed", type=int, default=7)
ap.add_argument("--config", default="B")
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg, ds, skw = configs.named_config(a.config, mask_ratio=0.75)
B = a.frames
pts = torch.from_numpy(synth.synth_batch(4242, B, ds.point_cloud_range, **skw)).to(dev)
vox = gplan.voxelize(pts, ds.point_cloud_range, ds.voxel_size, ds.grid_size, B)
noise = torch.rand(vox.M, generator=torch.Generator(device="cpu").manual_seed(11)).to(dev)

GROUPS = {
    "vfe": lambda n: n.startswith("vfe."),
    "spconv": lambda n: "conv_down" in n or "conv_" in n and "sst_block" in n and "encoder" not in n,
    "encoder": lambda n: ".encoder_list." in n or "attn" in n or "linear" in n,
    "deconv": lambda n: "decoder_deblocks" in n,
    "conv_out": lambda n: "decoder_conv_out" in n,
    "pred": lambda n: "decoder_pred" in n,
}

def run(mode, rnd=None, groups=None):
    torch.manual_seed(a.seed)
    net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev).train()
    net.sync_loss_scalar = False
    hit = {}
    if rnd is not None:
        with torch.no_grad():
            for n, p in net.named_parameters():
                if p.dim() < 2:
                    continue
                g = [k for k, f in GROUPS.items() if f(n)]
                if groups is None or any(k in groups for k in g):
                    p.copy_(p.to(rnd).float())
                    hit[g[0] if g else "other"] = hit.get(g[0] if g else "other", 0) + 1
    if mode == "bench":
        net.backbone_3d.dense_spatial_features = False
        opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(B), total_steps=10)
        opt.zero_grad()
    bd = {"points": pts, "batch_size": B, "mae_noise": noise}
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "bench"):
        ret, _, _ = net(bd)
    l = float(ret["loss"].detach())
    del net, bd, ret
    torch.cuda.empty_cache()
    return l, hit

names = None
l0, _ = run("fp32")
print(f"fp32 mode loss {l0:.7f}")
lb, _ = run("bench")
print(f"bench mode: {(lb - l0) / l0:+.3e}")
for label, rnd, groups in [("ALL weights bf16", torch.bfloat16, None), ("ALL weights fp16", torch.float16, None),
                           ("vfe bf16", torch.bfloat16, ["vfe"]), ("spconv bf16", torch.bfloat16, ["spconv"]),
                           ("encoder bf16", torch.bfloat16, ["encoder"]), ("deconv bf16", torch.bfloat16, ["deconv"]),
                           ("conv_out bf16", torch.bfloat16, ["conv_out"]), ("pred bf16", torch.bfloat16, ["pred"]),
                           ("deconv+conv_out fp16, rest bf16", None, None)]:
    if rnd is None:
        continue
    l, hit = run("fp32", rnd, groups)
    print(f"fp32 mode, {label}: {(l - l0) / l0:+.3e}   {hit}")
lbb, _ = run("bench", torch.bfloat16, None)
print(f"bench mode with pre-rounded (bf16) weights vs fp32 mode with the same weights: bench {(lbb - l0) / l0:+.3e} (vs exact-weight fp32)")
