"""Host-side cost of one training step by operator (torch.profiler, CPU activity, backward thread included)."""
import logging, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
from gdmae_hip import configs, optim, synth
from pcdet.models import build_network
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
net = build_network(cfg, 3, ds, logging.getLogger("p")).to(dev).train()
net.sync_loss_scalar = False
net.backbone_3d.dense_spatial_features = False
opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=100)
batches = [torch.from_numpy(synth.synth_batch(5 + i, 8, ds.point_cloud_range, **skw)).to(dev) for i in range(2)]
pend = {}
resident = torch.cuda.Event(); resident.record()
def step(i):
    pts, nxt = batches[i % 2], batches[(i + 1) % 2]
    opt.zero_grad()
    pf = pend.pop(i, None) or net.backbone_3d.prefetch_plan(pts, 8)
    bd = {"points": pts, "batch_size": 8}
    bd["_gdmae_vox"], bd["_gdmae_plan"] = pf.finish()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ret, _, _ = net(bd)
    ret["loss"].backward()
    pend[i + 1] = net.backbone_3d.prefetch_plan(nxt, 8, ready=resident)
    opt.step(i)
for i in range(6): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(6, 16): step(i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue time/step {(t1-t0)/10*1e3:.2f} ms; wall/step {(t2-t0)/10*1e3:.2f} ms")
if "--aten" in sys.argv:
    # which ATen operators still launch device kernels inside a step, and from which python line
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for i in range(16, 19): step(i)
        torch.cuda.synchronize()
    agg = {}
    for e in prof.events():
        dt = getattr(e, "self_device_time_total", 0)
        if not dt or e.name.startswith(("hip", "Cijk", "Custom_Cijk")) or "k_" in e.name[:24] or "gd_scan" in e.name:
            continue
        src = next((f for f in (e.stack or []) if "/repo/" in f), "")
        key = (e.name[:48], src.replace(REPO, "")[:100])
        t = agg.setdefault(key, [0, 0.0]); t[0] += 1; t[1] += dt
    for (name, src), (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{us / 3:9.1f} us/step {cnt / 3:6.1f} calls/step  {name:48s} {src}")
else:
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        for i in range(16, 19): step(i)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
