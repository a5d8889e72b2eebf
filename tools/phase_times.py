"""GPU time per phase of the bench step from HIP events on the training stream (untraced run): forward, backward, optimizer,
and the gaps between consecutive steps (stream idle while the host prepares the next step)."""
import logging, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from gdmae_hip import configs, optim, synth
from pcdet.models import build_network
dev = torch.device("cuda:0")
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
torch.manual_seed(1234)
net = build_network(cfg, 3, ds, logging.getLogger("p")).to(dev).train()
net.sync_loss_scalar = False
net.backbone_3d.dense_spatial_features = False
opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=200)
batches = [torch.from_numpy(synth.synth_batch(5 + i, 8, ds.point_cloud_range, **skw)).to(dev) for i in range(4)]
resident = torch.cuda.Event(); resident.record()
pend = {}
marks = []
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
def step(i, rec):
    pts, nxt = batches[i % 4], batches[(i + 1) % 4]
    e0 = ev()
    opt.zero_grad()
    pf = pend.pop(i, None) or net.backbone_3d.prefetch_plan(pts, 8)
    bd = {"points": pts, "batch_size": 8, "_gdmae_grad_sync": opt.sync}
    bd["_gdmae_vox"], bd["_gdmae_plan"] = pf.finish() if hasattr(pf, "finish") else pf
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ret, _, _ = net(bd)
    e1 = ev()
    pfn = net.backbone_3d.prefetch_plan(nxt, 8, ready=resident)
    ret["loss"].backward()
    e2 = ev()
    opt.all_reduce_grads()
    opt.step(i)
    e3 = ev()
    pend[i + 1] = pfn.finish()
    if rec: marks.append((e0, e1, e2, e3))
for i in range(10): step(i, False)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 40
for i in range(10, 10 + N): step(i, True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
f = sum(a.elapsed_time(b) for a, b, _, _ in marks) / N
b = sum(b_.elapsed_time(c) for _, b_, c, _ in marks) / N
o = sum(c.elapsed_time(d) for _, _, c, d in marks) / N
gap = sum(marks[k][3].elapsed_time(marks[k + 1][0]) for k in range(N - 1)) / (N - 1)
print(f"wall/step {wall:.3f} ms | forward span {f:.3f} backward span {b:.3f} optimizer span {o:.3f} step-to-step gap {gap:.3f} | sum {f + b + o + gap:.3f}")
