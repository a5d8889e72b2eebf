"""ATen operators of one training step of any bench configuration (dispatch-mode trace: op, calling python line, shape), with
the element count of the first tensor argument - where the framework-issued elementwise traffic of a step comes from.
Usage (GPU box): python tools/trace_aten_ops_cfg.py [--config D] [--batch 8]"""
import argparse, collections, os, sys, traceback
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="D")
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
sys.argv = [sys.argv[0], "--config", a.config]
args = bench.parse()
dev = torch.device("cuda:0")
wl = bench.Workload(args, a.config, a.batch, dev, 0, 1, args.mask_ratio, 20, 2)
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.slice", "aten.select", "aten.t.", "aten.transpose", "aten.permute",
        "aten.expand", "aten.as_strided", "aten.empty", "aten.alias", "aten.unsqueeze", "aten.squeeze", "aten.reshape", "aten.split",
        "aten.unbind", "aten.lift_fresh", "aten._local_scalar_dense", "aten.is_", "aten.sym_", "aten.stride", "aten.size", "aten.numel",
        "aten.storage_offset", "aten.dim", "aten.record_stream", "aten.set_", "aten.resize_", "aten.new_empty", "aten.is_pinned")


class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.hits = collections.defaultdict(lambda: [0, 0])

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            ts = [x for x in args if isinstance(x, torch.Tensor) and x.is_cuda]
            if ts:
                fr = [f for f in traceback.extract_stack() if "/repo/" in f.filename and "trace_aten_ops" not in f.filename and "bench.py" not in f.filename]
                where = f"{fr[-1].filename.replace(REPO, '')}:{fr[-1].lineno}" if fr else "(autograd engine)"
                if where == "(autograd engine)" and name.startswith(("aten.add.Tensor", "aten.mul.Tensor", "aten._to_copy")):
                    print("  autograd-engine", name, [(tuple(t.shape), str(t.dtype).replace("torch.", ""), t.stride()) for t in ts][:2])
                h = self.hits[(name, where)]
                h[0] += 1
                h[1] += max(t.numel() * t.element_size() for t in ts)
        return func(*args, **(kwargs or {}))


wl.feed_resident(3, 0)
torch.cuda.synchronize()
with Trace() as tr:
    wl.feed_resident(1, 3)
torch.cuda.synchronize()
tot = 0
for (name, where), (c, by) in sorted(tr.hits.items(), key=lambda kv: -kv[1][1]):
    tot += c
    print(f"{c:4d}  {by / 1e6:9.1f} MB  {name:40s} {where}")
print("total device ATen calls:", tot)
