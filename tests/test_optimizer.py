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
    # layout: one contiguous range per gradient bucket (VFE, the three SST stages, the decoder - forward order), inside a
    # bucket the optimised parameters first; the Adam segments are exactly the optimised ranges
    assert [b for b, _, _ in opt.buckets] == ["vfe", "backbone_3d.sst_blocks.0", "backbone_3d.sst_blocks.1",
                                              "backbone_3d.sst_blocks.2", "backbone_3d.decoder"]
    assert opt.buckets[0][1] == 0 and opt.buckets[-1][2] == opt.n
    assert all(a[2] == b[1] for a, b in zip(opt.buckets, opt.buckets[1:]))
    seg = list(zip(opt.segments[0::2], opt.segments[1::2]))
    assert sum(e - b for b, e in seg) == opt.n_opt
    table = opt._offsets()
    for q in opt.params:
        o, k = table[id(q)]
        inside = any(b <= o and o + k <= e for b, e in seg)
        assert inside == (id(q) not in frozen_ids), names[id(q)]
        bname = optim.default_bucket_of(names[id(q)])
        lo, hi = next((lo, hi) for b, lo, hi in opt.buckets if b == bname)
        assert lo <= o and o + k <= hi


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
    for b, e in zip(opt.segments[0::2], opt.segments[1::2]):
        assert torch.equal(opt2.exp_avg[b:e], opt.exp_avg[b:e])
    assert opt2.segments == opt.segments and torch.equal(opt2.flat_param, opt.flat_param)


def test_checkpoint_loader_adapts_sparse_conv_layouts_and_filters_like_the_reference(tmp_path):
    """f4: ``_load_state_dict`` (reference detector3d_template.py:360-388): sparse-conv weights stored with the last two axes
    swapped or kernel-first (another spconv generation) are re-laid, foreign / mis-shaped keys are ignored, and with
    ``strict`` the FILTERED dict is what must cover the model."""
    import logging
    from pcdet.models import build_network
    from pcdet.utils.spconv_utils import find_all_spconv_keys
    cfg, ds, _ = configs.named_config("A")
    log = logging.getLogger("t")
    torch.manual_seed(0)
    net = build_network(cfg, len(ds.class_names), ds, log)
    keys = sorted(find_all_spconv_keys(net))
    assert keys and all(k.endswith("weight") for k in keys)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    disk = dict(sd)
    k0 = keys[0]
    disk[k0] = sd[k0].permute(1, 2, 3, 0).contiguous()                  # (kH, kW, Cin, Cout): spconv-1.x kernel-first layout
    disk["some.foreign.key"] = torch.zeros(3)
    disk["vfe.dvfe_mlps.0.0.weight"] = torch.zeros(5, 5)               # mis-shaped: must be ignored, not loaded
    torch.manual_seed(1)
    net2 = build_network(cfg, len(ds.class_names), ds, log)
    w_before = net2.state_dict()["vfe.dvfe_mlps.0.0.weight"].clone()
    state, update = net2._load_state_dict(disk, strict=False)
    assert "some.foreign.key" not in update and "vfe.dvfe_mlps.0.0.weight" not in update and k0 in update
    assert torch.equal(net2.state_dict()[k0], sd[k0])
    assert torch.equal(net2.state_dict()["vfe.dvfe_mlps.0.0.weight"], w_before)
    if sd[k0].shape[0] == sd[k0].shape[3]:                               # square: the swapped-axes form has the same shape
        pass
    else:
        disk2 = dict(sd)
        disk2[k0] = sd[k0].transpose(-1, -2).contiguous() if sd[k0].transpose(-1, -2).shape != sd[k0].shape else sd[k0]
        net2._load_state_dict(disk2, strict=False)
        assert torch.equal(net2.state_dict()[k0], sd[k0])
    # strict: the filtered dict must cover every key of the model -> the mis-shaped entry makes it incomplete
    with pytest.raises(RuntimeError):
        net2._load_state_dict(disk, strict=True)
    good = dict(sd)
    good[k0] = sd[k0].permute(1, 2, 3, 0).contiguous()
    good["some.foreign.key"] = torch.zeros(3)
    net2._load_state_dict(good, strict=True)                            # foreign keys do not break strict loading
    assert all(torch.equal(net2.state_dict()[k], sd[k]) for k in sd)
