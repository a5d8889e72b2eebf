"""Stand-alone GPU time of the geometry plan of one config-B batch (nothing else running): HIP events around the launches."""
import logging, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from gdmae_hip import configs, synth
from pcdet.models import build_network
dev = torch.device("cuda:0")
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
net = build_network(cfg, 3, ds, logging.getLogger("p")).to(dev).train()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batches = [torch.from_numpy(synth.synth_batch(5 + i, B, ds.point_cloud_range, **skw)).to(dev) for i in range(3)]
from gdmae_hip import plan as P
P.PlanPrefetch._side[dev.index] = torch.cuda.current_stream()
for i in range(5):
    net.backbone_3d.prefetch_plan(batches[i % 3], B).finish()
torch.cuda.synchronize()
ts = []
for i in range(20):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    pf = net.backbone_3d.prefetch_plan(batches[i % 3], B)
    b.record()
    pf.finish()
    torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
ts.sort()
print(f"plan stand-alone, {B} frames: median {ts[len(ts)//2]:.3f} ms, min {ts[0]:.3f} ms (host-issued, includes launch gaps)")
