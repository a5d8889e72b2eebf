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


def test_optimised_parameter_set_equals_the_reference():
    """The reference's 'adam_onecycle' optimiser only holds the parameters of LEAF modules (flatten_model): the golden
    lists what the imported reference optimises for this model; FlatAdamOneCycle must freeze exactly the rest."""
    import json
    import logging
    from pcdet.models import build_network
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "optimizer_params.json")))
    cfg, ds, _ = configs.named_config("B", mask_ratio=0.75)
    net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t"))
    names = {id(p): n for n, p in net.named_parameters()}
    opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)
    assert sorted(names[id(p)] for p in opt.frozen) == g["skipped"]
    frozen_ids = {id(q) for q in opt.frozen}
    assert sorted(names[id(p)] for p in opt.params if id(p) not in frozen_ids) == sorted(g["optimised_non_bn"] + g["optimised_bn"])
    assert opt.n == g["numel_total"] and opt.n - opt.n_opt == g["numel_skipped"]
    assert all(n.endswith(("in_proj_weight", "in_proj_bias", "tau")) for n in g["skipped"])
    # the frozen tensors are the tail of the flat buffers
    tail = opt.flat_param[opt.n_opt:]
    assert opt.frozen[0].data_ptr() == tail.data_ptr()


@pytest.mark.gpu
def test_frozen_parameters_keep_their_values_but_count_in_the_clip_norm():
    dev = torch.device("cuda:0")
    net = torch.nn.Module()
    net.attn = torch.nn.MultiheadAttention(8, 2)                 # in_proj_* are direct parameters, out_proj is a child
    net.lin = torch.nn.Linear(8, 8)
    net = net.to(dev)
    cfgo = configs.optimization_cfg(8)
    res = {}
    for ref_groups in (True, False):
        torch.manual_seed(0)
        for p in net.parameters():
            p.data = torch.randn_like(p)                          # fresh storage (the previous optimizer owns the old views)
        opt = optim.FlatAdamOneCycle(net, cfgo, total_steps=10, reference_layer_groups=ref_groups)
        before = {n: p.detach().clone() for n, p in net.named_parameters()}
        opt.zero_grad()
        torch.manual_seed(1)
        for p in net.parameters():
            p.grad.copy_(torch.randn_like(p) * 100)               # large: the global clip is active
        opt.step(0)
        res[ref_groups] = {n: (p.detach() - before[n]).abs().max().item() for n, p in net.named_parameters()}
    assert res[True]["attn.in_proj_weight"] == 0 and res[True]["attn.in_proj_bias"] == 0
    assert res[True]["attn.out_proj.weight"] > 0 and res[True]["lin.weight"] > 0
    assert res[False]["attn.in_proj_weight"] > 0
    # same clip factor in both modes (the frozen gradients are part of the norm): identical updates of the others
    assert res[True]["lin.weight"] == res[False]["lin.weight"]


def test_checkpoint_wire_format_round_trips_through_torch_adam(tmp_path):
    """f4: optimizer_state written by FlatAdamOneCycle is a torch.optim.Adam state_dict with the reference's two param
    groups in the reference's order (golden from the imported reference) - a real torch Adam built the reference way
    loads it, and what that Adam saves loads back."""
    import json
    import logging
    from gdmae_hip import checkpoint as ck
    from pcdet.models import build_network
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "optimizer_params.json")))
    cfg, ds, _ = configs.named_config("B", mask_ratio=0.75)
    log = logging.getLogger("t")
    net = build_network(cfg, len(ds.class_names), ds, log)
    names = {id(p): n for n, p in net.named_parameters()}
    opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)
    groups = opt._reference_groups()
    assert [[names[id(p)] for p in gr] for gr in groups] == g["group_order"]
    # fake a few steps worth of state
    opt.t = 3
    opt.exp_avg.copy_(torch.randn(opt.n))
    opt.exp_avg_sq.copy_(torch.rand(opt.n))
    ck.save_checkpoint(ck.checkpoint_state(net, opt, epoch=2, it=3), str(tmp_path / "checkpoint_epoch_2"))
    disk = torch.load(str(tmp_path / "checkpoint_epoch_2.pth"), weights_only=False)
    assert set(disk) == {"epoch", "it", "model_state", "optimizer_state", "version"}
    assert set(disk["optimizer_state"]["param_groups"][0]) >= set(g["group_keys"]) - {"maximize", "foreach", "capturable",
                                                                                     "differentiable", "fused", "decoupled_weight_decay"}
    adam = torch.optim.Adam([{"params": gr, "lr": 0} for gr in groups], betas=(0.9, 0.99))
    adam.load_state_dict(disk["optimizer_state"])                      # the reference's resume path
    p0 = groups[0][5]
    o, k = opt._offsets()[id(p0)]
    assert torch.equal(adam.state[p0]["exp_avg"].reshape(-1), opt.exp_avg[o:o + k])
    # and back: a fresh model / optimizer resumes from what torch Adam writes
    net2 = build_network(cfg, len(ds.class_names), ds, log)
    opt2 = optim.FlatAdamOneCycle(net2, configs.optimization_cfg(8), total_steps=10)
    torch.save({"epoch": 2, "it": 3, "model_state": net.state_dict(), "optimizer_state": adam.state_dict(), "version": "x"},
               str(tmp_path / "ref.pth"))
    it, ep = net2.load_params_with_optimizer(str(tmp_path / "ref.pth"), to_cpu=True, optimizer=opt2, logger=log)
    assert (it, ep) == (3, 2) and opt2.t == 3
    assert torch.equal(opt2.exp_avg[:opt2.n_opt], opt.exp_avg[:opt.n_opt]) and torch.equal(opt2.flat_param, opt.flat_param)
