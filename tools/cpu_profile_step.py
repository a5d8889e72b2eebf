"""Where does the HOST time of one training step go?  (cProfile of step() on the GPU box; GPU work is async.)"""
import cProfile, pstats, io, logging, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from gdmae_hip import configs, optim, synth
from pcdet.models import build_network
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
net = build_network(cfg, 3, ds, logging.getLogger("p")).to(dev).train()
net.sync_loss_scalar = False
net.backbone_3d.dense_spatial_features = False
opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=100)
pts = torch.from_numpy(synth.synth_batch(5, 8, ds.point_cloud_range, **skw)).to(dev)
def step(i):
    opt.zero_grad()
    bd = {"points": pts, "batch_size": 8}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ret, _, _ = net(bd)
    ret["loss"].backward()
    opt.step(i)
for i in range(6): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5): step(i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue time/step {(t1-t0)/5*1e3:.2f} ms; wall/step {(t2-t0)/5*1e3:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for i in range(3): step(i)
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
