"""Host-side cost of the bench step by python function (cProfile, cumulative), at a chosen frames-per-step: what a rank of
config C (4 frames per GPU) spends on the host per step next to its GPU time."""
import cProfile, io, logging, os, pstats, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from gdmae_hip import configs, optim, synth
from pcdet.models import build_network
dev = torch.device("cuda:0")
NB = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 4
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
torch.manual_seed(1234)
net = build_network(cfg, 3, ds, logging.getLogger("p")).to(dev).train()
net.sync_loss_scalar = False
net.backbone_3d.dense_spatial_features = False
opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(NB), total_steps=400)
batches = [torch.from_numpy(synth.synth_batch(5 + i, NB, ds.point_cloud_range, **skw)).to(dev) for i in range(4)]
resident = torch.cuda.Event(); resident.record()
pend = {}
def step(i):
    pts, nxt = batches[i % 4], batches[(i + 1) % 4]
    opt.zero_grad()
    pf = pend.pop(i, None) or net.backbone_3d.prefetch_plan(pts, NB)
    bd = {"points": pts, "batch_size": NB, "_gdmae_grad_sync": opt.sync}
    bd["_gdmae_vox"], bd["_gdmae_plan"] = pf.finish()
    pend[i + 1] = net.backbone_3d.prefetch_plan(nxt, NB, ready=resident)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ret, _, _ = net(bd)
    ret["loss"].backward()
    opt.step(i)
for i in range(10): step(i)
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for i in range(10, 10 + K): step(i)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"frames/step {NB}: host issue {(t1 - t0) / K * 1e3:.2f} ms/step, wall {(t2 - t0) / K * 1e3:.2f} ms/step")
# host-only cost: the same loop with a device sync before every step, so the host never waits on the GPU inside the step
hs = 0.0
for i in range(10 + K, 10 + 2 * K):
    torch.cuda.synchronize(); a = time.perf_counter(); step(i); hs += time.perf_counter() - a
print(f"host time of a step issued into an idle device: {hs / K * 1e3:.2f} ms")
if "--single-thread-backward" in sys.argv:      # backward nodes on the calling thread: cProfile sees the Function.backward bodies
    torch.autograd.set_multithreading_enabled(False)
    hs = 0.0
    for i in range(10 + 2 * K, 10 + 3 * K):
        torch.cuda.synchronize(); a = time.perf_counter(); step(i); hs += time.perf_counter() - a
    print(f"host time of a step issued into an idle device, single-threaded autograd: {hs / K * 1e3:.2f} ms")
pr = cProfile.Profile()
base = 10 + 2 * K
torch.cuda.synchronize()
pr.enable()
for i in range(base, base + K):
    step(i)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("cumulative")
ps.print_stats(70)
print(s.getvalue().replace(REPO, "."))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue().replace(REPO, "."))
