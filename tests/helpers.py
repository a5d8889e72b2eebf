"""Shared helpers for the parity tests (golden loading, sampled-tensor comparison)."""
import os

import numpy as np
import torch

from gdmae_hip import configs
from oracle import gdmae_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DATASETS = {"kitti_b2": configs.KITTI, "kitti_b2_m75": configs.KITTI, "waymo_b1": configs.WAYMO, "once_e_b1": configs.ONCE}


def load_case(name):
    z = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    ds = configs.SyntheticDatasetInfo(**DATASETS[name])
    if name.startswith("once_e"):                       # BASELINE config E: d = 256 in every stage, 6 layers, VFE 64 -> 256
        cfg = configs.named_config("E", mask_ratio=float(z["mask_ratio"]))[0]
    else:
        cfg = configs.gdmae_ssl_model_cfg(mask_ratio=float(z["mask_ratio"]),
                                          eval_metric="kitti" if name.startswith("kitti") else "waymo_custom")
    shapes = orc.param_shapes(cfg, int(z["num_point_features"]))
    return z, ds, cfg, shapes


def sample(t: torch.Tensor, n=4096):
    f = t.detach().reshape(-1).double().cpu()
    step = max(1, f.numel() // n)
    return f[::step][:n].float().numpy(), np.array([float(f.sum()), float(f.abs().sum()), float((f * f).sum())])


def assert_sampled_close(t, s_ref, c_ref, rtol, what):
    s, c = sample(t)
    scale = np.abs(s_ref).max() + 1e-12
    err = np.abs(s - s_ref).max() / scale
    assert err <= rtol, f"{what}: sampled rel-max-err {err:.3e} > {rtol}"
    # abs-sum and square-sum checksums are well conditioned
    for i in (1, 2):
        e = abs(c[i] - c_ref[i]) / (abs(c_ref[i]) + 1e-12)
        assert e <= rtol, f"{what}: checksum[{i}] rel err {e:.3e} > {rtol}"


from head_seed import seeded_head_state  # noqa: E402,F401  (stand-alone module: the golden generator imports it too)
