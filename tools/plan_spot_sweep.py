"""Where in step t should the geometry plan of batch t + 1 be issued?  The plan (side stream) is issued right before the N-th C-ABI
call of the step (forward and backward calls counted together), for a list of N; prints wall time per step for each.
    python tools/plan_spot_sweep.py [--frames 4] [--spots 0,5,10,...]"""
import logging, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from gdmae_hip import configs, optim, synth
from gdmae_hip import lib as L
from pcdet.models import build_network
dev = torch.device("cuda:0")
NB = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 8
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
torch.manual_seed(1234)
net = build_network(cfg, 3, ds, logging.getLogger("p")).to(dev).train()
net.sync_loss_scalar = False
net.backbone_3d.dense_spatial_features = False
opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(NB), total_steps=4000)
torch.autograd.set_multithreading_enabled(False)
batches = [torch.from_numpy(synth.synth_batch(5 + i, NB, ds.point_cloud_range, **skw)).to(dev) for i in range(4)]
resident = torch.cuda.Event(); resident.record()

orig_call = L.call
state = {"n": 0, "spot": -1, "fire": None, "names": []}
def counted(name, *a):
    if state["fire"] is not None and state["n"] == state["spot"]:
        f, state["fire"] = state["fire"], None
        f()
    if state["record"]:
        state["names"].append(name)
    state["n"] += 1
    return orig_call(name, *a)
state["record"] = False
L.call = counted

pend = {}
def step(i, spot):
    pts, nxt = batches[i % 4], batches[(i + 1) % 4]
    opt.zero_grad()
    plan = pend.pop(i, None) or net.backbone_3d.prefetch_plan(pts, NB).finish()
    bd = {"points": pts, "batch_size": NB, "_gdmae_grad_sync": opt.sync}
    bd["_gdmae_vox"], bd["_gdmae_plan"] = plan
    box = []
    state["n"], state["spot"] = 0, spot
    state["fire"] = lambda: box.append(net.backbone_3d.prefetch_plan(nxt, NB, ready=resident))
    if spot < 0:
        state["fire"](); state["fire"] = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ret, _, _ = net(bd)
    ret["loss"].backward()
    if state["fire"] is not None:
        state["fire"](); state["fire"] = None
    opt.all_reduce_grads()
    opt.step(i)
    pend[i + 1] = box[0].finish()

it = 0
for _ in range(6):
    step(it, -1); it += 1
state["record"] = True; state["names"] = []
step(it, -1); it += 1
state["record"] = False
names = list(state["names"])
print("C-ABI calls of a step (index: name):")
print("  " + "  ".join(f"{i}:{n.replace('gdmae_', '')}" for i, n in enumerate(names)))
spots = [int(x) for x in sys.argv[sys.argv.index("--spots") + 1].split(",")] if "--spots" in sys.argv else [-1] + list(range(0, len(names), 3))
K = 24
for rep in range(2):
    for sp in spots:
        for _ in range(3):
            step(it, sp); it += 1
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(K):
            step(it, sp); it += 1
        torch.cuda.synchronize()
        print(f"rep {rep} spot {sp:3d} ({'start' if sp < 0 else names[sp].replace('gdmae_', '') if sp < len(names) else 'end'}): {(time.perf_counter() - t0) / K * 1e3:.3f} ms/step", flush=True)
