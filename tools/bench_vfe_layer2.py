"""Kernel timing of the DynVFE layers (PointLayer1 pillar-major + PointLayer2Max) on a config-B batch."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch  # noqa: E402

from gdmae_hip import configs, plan as gplan, synth, vfe as gvfe  # noqa: E402

dev = torch.device("cuda:0")
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
B = 8
pts = torch.from_numpy(synth.synth_batch(0, B, ds.point_cloud_range, **skw)).to(dev)
vox = gplan.voxelize(pts, ds.point_cloud_range, ds.voxel_size, ds.grid_size, B)
N, D = int(vox.N), vox.n_cols + 5
torch.manual_seed(0)
W1 = (torch.randn(64, D, device=dev) * 0.1).requires_grad_()
W2 = (torch.randn(128, 64, device=dev) * 0.1).requires_grad_()
g1, b1 = torch.ones(64, device=dev, requires_grad=True), torch.zeros(64, device=dev, requires_grad=True)
g2, b2 = torch.ones(128, device=dev, requires_grad=True), torch.zeros(128, device=dev, requires_grad=True)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for p in (W1, W2, g1, b1, g2, b2):
        p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y, _, _ = gvfe.PointLayer1.apply(vox, W1, g1, b1, 1e-3, None, True)
        o, _, _ = gvfe.PointLayer2Max.apply(y, vox.row_pillar, W2, g2, b2, 1e-3, vox.pt_off, None)
    o.backward(torch.ones_like(o))
torch.cuda.synchronize()
print("ok", N, int(vox.M))
