"""a21: one-cycle schedule + fused clip/decay/Adam against the reference's known answers."""
import os

import numpy as np
import pytest
import torch

from gdmae_hip import configs, optim
from oracle import optim_oracle as oo

Z = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "optimizer.npz")))


def test_one_cycle_matches_reference_schedule():
    for t in range(100):
        for fn in (oo.one_cycle, optim.one_cycle):
            lr, mom = fn(t, 100, 0.003, [0.95, 0.85], 10, 0.4)
            assert abs(lr - Z["lr"][t]) <= 1e-12 + 1e-9 * Z["lr"][t], (fn.__module__, t)
            assert abs(mom - Z["mom"][t]) <= 1e-12
    # SURVEY §9.8 spot values
    assert abs(oo.one_cycle(20, 100, 0.003, [0.95, 0.85], 10, 0.4)[0] - 1.65e-3) < 1e-9


def _toy(device):
    shapes = [tuple(int(v) for v in s if v > 0) for s in Z["shapes"]]
    ps, off = [], 0
    for s in shapes:
        k = int(np.prod(s))
        ps.append(torch.from_numpy(Z["init"][off:off + k].reshape(s).copy()).to(device).requires_grad_(True))
        off += k
    return ps, shapes


def _set_grads(ps, shapes, flat):
    off = 0
    for p, s in zip(ps, shapes):
        k = int(np.prod(s))
        g = torch.from_numpy(flat[off:off + k].reshape(s).copy()).to(p.device)
        if p.grad is None:
            p.grad = g
        else:
            p.grad.copy_(g)
        off += k


def test_oracle_optimizer_matches_reference_trajectory():
    ps, shapes = _toy("cpu")
    opt = oo.AdamOneCycle(ps, wd=0.01)
    for t in range(3):
        _set_grads(ps, shapes, Z["grads"][t])
        opt.step(*oo.one_cycle(t, 100, 0.003, [0.95, 0.85], 10, 0.4))
        cur = np.concatenate([p.detach().numpy().ravel() for p in ps])
        assert np.abs(cur - Z["traj"][t]).max() <= 2e-6 * np.abs(Z["traj"][t]).max()


@pytest.mark.gpu
def test_hip_flat_adam_matches_reference_trajectory():
    dev = torch.device("cuda:0")
    ps, shapes = _toy(dev)
    model = torch.nn.Module()
    for i, p in enumerate(ps):
        model.register_parameter(f"p{i}", torch.nn.Parameter(p.detach().clone()))
    cfg = configs.optimization_cfg()
    opt = optim.FlatAdamOneCycle(model, cfg, total_steps=100)
    mp = list(model.parameters())
    for t in range(3):
        opt.zero_grad()
        _set_grads(mp, shapes, Z["grads"][t])
        opt.step(t)
        cur = np.concatenate([p.detach().cpu().numpy().ravel() for p in mp])
        assert np.abs(cur - Z["traj"][t]).max() <= 5e-6 * np.abs(Z["traj"][t]).max()
    # global-norm clipping active: scale the gradient 1000x -> same direction, clipped magnitude
    opt2_ps, _ = _toy(dev)
    big = [torch.nn.Parameter(p.detach().clone()) for p in opt2_ps]
    m2 = torch.nn.Module()
    for i, p in enumerate(big):
        m2.register_parameter(f"p{i}", p)
    o2 = optim.FlatAdamOneCycle(m2, cfg, total_steps=100)
    ref_ps, _ = _toy("cpu")
    ro = oo.AdamOneCycle(ref_ps, wd=0.01)
    _set_grads(list(m2.parameters()), shapes, Z["grads"][0] * 1000)
    _set_grads(ref_ps, shapes, Z["grads"][0] * 1000)
    o2.step(0)
    ro.step(*oo.one_cycle(0, 100, 0.003, [0.95, 0.85], 10, 0.4))
    a = np.concatenate([p.detach().cpu().numpy().ravel() for p in m2.parameters()])
    b = np.concatenate([p.detach().numpy().ravel() for p in ref_ps])
    assert np.abs(a - b).max() <= 5e-6 * np.abs(b).max()
