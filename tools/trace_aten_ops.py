"""Which ATen operators still launch device work inside one training step, and from which python line (dispatch-mode trace)."""
import collections, logging, os, sys, traceback
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from gdmae_hip import configs, optim, synth
from pcdet.models import build_network

SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.slice", "aten.select", "aten.t.", "aten.transpose", "aten.permute",
        "aten.expand", "aten.as_strided", "aten.empty", "aten.alias", "aten.unsqueeze", "aten.squeeze", "aten.reshape", "aten.split",
        "aten.unbind", "aten.lift_fresh", "aten._local_scalar_dense", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.numel",
        "aten.storage_offset", "aten.dim", "aten.record_stream", "aten.set_", "aten.resize_", "aten.new_empty", "aten.is_pinned")


class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.hits = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            dev = any(isinstance(a, torch.Tensor) and a.is_cuda for a in args)
            if dev:
                fr = [f for f in traceback.extract_stack() if "/repo/" in f.filename and "trace_aten_ops" not in f.filename]
                where = f"{fr[-1].filename.replace(REPO, '')}:{fr[-1].lineno}" if fr else "(autograd engine)"
                shape = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
                self.hits[(name, where, shape if name.startswith(("aten.mm", "aten.bmm", "aten.addmm")) else None)] += 1
        return func(*args, **(kwargs or {}))


dev = torch.device("cuda:0")
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
net = build_network(cfg, 3, ds, logging.getLogger("p")).to(dev).train()
net.sync_loss_scalar = False
net.backbone_3d.dense_spatial_features = False
opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=100)
pts = torch.from_numpy(synth.synth_batch(5, 8, ds.point_cloud_range, **skw)).to(dev)


def step(i):
    opt.zero_grad()
    pf = net.backbone_3d.prefetch_plan(pts, 8)
    bd = {"points": pts, "batch_size": 8}
    bd["_gdmae_vox"], bd["_gdmae_plan"] = pf.finish()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ret, _, _ = net(bd)
    ret["loss"].backward()
    opt.step(i)


for i in range(3):
    step(i)
torch.cuda.synchronize()
with Trace() as tr:
    step(3)
torch.cuda.synchronize()
for (name, where, shape), c in sorted(tr.hits.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"{c:4d}  {name:44s} {where} {shape or ''}")
print("total device ATen calls:", sum(tr.hits.values()))
