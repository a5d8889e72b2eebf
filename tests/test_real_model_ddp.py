"""N > 1 on the REAL model (reference tools/train.py:146 = torch DDP's bucketed reducer).

1. Two ranks sharing cuda:0 over gloo (RCCL refuses two ranks per device; the hooks, buckets, streams and collective calls
   are the same code), each running the bench configuration of GDMAE (flat optimizer, bf16 autocast, hand-written backwards
   that write weight gradients straight into the flat buffer):
     * overlapped bucketed exchange (hooks inside backward) == tail-only exchange, bit for bit;
     * both == the sum of the two ranks' local gradients (the collective of two addends is order-free in fp32);
     * 'check' mode: no bucket changes after the hook that declares it complete.
2. The network under the reference's actual wrappers: nn.parallel.DistributedDataParallel (world 1) + a stock
   torch.optim.Adam, no flat optimizer - gradients equal the unwrapped model's, the step runs, model_func handles .module.
"""
import logging
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(config, B, rank):
    from gdmae_hip import configs, optim, synth
    from pcdet.models import build_network
    cfg, ds, skw = configs.named_config(config, mask_ratio=0.5 if config == "A" else 0.75)
    torch.manual_seed(1234)                                   # identical replicas, as bench.py does
    net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to("cuda:0").train()
    net.sync_loss_scalar = False
    net.backbone_3d.dense_spatial_features = False
    pts = torch.from_numpy(synth.synth_batch(100000 * rank + 5, B, ds.point_cloud_range, **skw)).to("cuda:0")
    return cfg, ds, net, pts


def _guard(fn):
    """Run a spawned worker; a failure is reported through the queue instead of a 15-minute timeout of the parent."""
    def run(*args):
        q = args[-1]
        try:
            fn(*args)
        except BaseException:   # noqa: BLE001
            import traceback
            q.put(("error", traceback.format_exc()))
            raise
    return run


def _grad_worker(rank, world, port, config, B, q):
    _guard(_grad_worker_body)(rank, world, port, config, B, q)


def _grad_worker_body(rank, world, port, config, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gdmae_hip import configs, optim
    cfg, ds, net, pts = _build(config, B, rank)
    opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(B), total_steps=10)
    opt.sync.autocheck = False          # the modes are driven explicitly here (the automatic first-step check is covered on the CPU)
    grads, logs = {}, {}
    for mode in ("off", "overlap", "tail", "check"):
        opt.sync.mode = mode
        opt.zero_grad()
        torch.manual_seed(99 + rank)                          # the same masking noise in every pass (drawn from the device RNG)
        bd = {"points": pts, "batch_size": B, "_gdmae_grad_sync": opt.sync}
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ret, _, _ = net(bd)
        ret["loss"].backward()
        opt.all_reduce_grads()
        torch.cuda.synchronize()
        grads[mode] = opt.flat_grad.detach().clone()
        logs[mode] = list(opt.sync.log)
    local = grads["off"]
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    total = gathered[0] + gathered[1]
    ok = {
        "overlap_eq_tail": bool(torch.equal(grads["overlap"], grads["tail"])),
        "check_eq_tail": bool(torch.equal(grads["check"], grads["tail"])),
        "tail_eq_sum": bool(torch.equal(grads["tail"], total)),
        "max_abs_diff_sum": float((grads["tail"] - total).abs().max()),
        "local_differs": bool(not torch.equal(gathered[0], gathered[1])),
        "nonzero_frac": float((local != 0).float().mean()),
        "log_overlap": logs["overlap"], "log_tail": logs["tail"], "log_off": logs["off"],
        "buckets": [b for b, _, _ in opt.buckets],
        "scale": opt._grad_scale,
    }
    # one optimizer step on the reduced gradient: replicas stay identical
    opt.sync.mode = "overlap"
    opt.step(0)
    torch.cuda.synchronize()
    ps = [torch.zeros_like(opt.flat_param) for _ in range(world)]
    dist.all_gather(ps, opt.flat_param.detach())
    ok["replicas_identical_after_step"] = bool(torch.equal(ps[0], ps[1]))
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(config, B):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_grad_worker, args=(r, world, port, config, B, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = [q.get(timeout=900) for _ in range(world)]
    err = [g for g in got if g[0] == "error"]
    if err:
        for p in ps:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
        raise AssertionError(err[0][1])
    res = sorted(got, key=lambda t: t[0])
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.gpu
@pytest.mark.parametrize("config,B", [("A", 2), ("B", 1)])
def test_overlapped_bucketed_allreduce_on_the_real_model_two_ranks(config, B):
    res = _run_two_ranks(config, B)
    for rank, ok in res:
        assert ok["local_differs"] and ok["nonzero_frac"] > 0.9, ok          # different frames per rank, gradients everywhere
        assert ok["overlap_eq_tail"], ok                                        # hooks inside backward == after backward
        assert ok["check_eq_tail"], ok                                          # (and no bucket was handed over unfinished)
        assert ok["tail_eq_sum"], ok                                            # == sum of the single-rank gradients
        assert ok["replicas_identical_after_step"], ok
        names = ok["buckets"]
        assert names[0] == "vfe" and names[-1] == "backbone_3d.decoder"
        # decoder first, then the stages last-to-first from inside backward; only the VFE bucket at the tail
        want = [[names[-1], "overlapped"]] + [[b, "overlapped"] for b in reversed(names[1:-1])] + [["vfe", "tail"]]
        assert [list(x) for x in ok["log_overlap"]] == want, ok["log_overlap"]
        assert all(h == "tail" for _, h in ok["log_tail"]) and len(ok["log_tail"]) == len(names)
        assert ok["log_off"] == []


def _rccl_worker(port, q):
    _guard(_rccl_worker_body)(port, q)


def _rccl_worker_body(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)          # nccl == RCCL on ROCm
    warm = torch.ones(1, device="cuda:0")
    dist.all_reduce(warm)                                             # communicator created on the main thread, as bench.py does
    from gdmae_hip import configs, optim
    cfg, ds, net, pts = _build("A", 2, 0)
    opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(2), total_steps=10)
    opt.sync.force = True
    opt.sync.autocheck = False
    grads, logs = {}, {}
    for mode in ("off", "overlap", "tail", "check", "overlap"):
        opt.sync.mode = mode
        opt.zero_grad()
        torch.manual_seed(99)
        bd = {"points": pts, "batch_size": 2, "_gdmae_grad_sync": opt.sync}
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ret, _, _ = net(bd)
        ret["loss"].backward()
        opt.all_reduce_grads()
        torch.cuda.synchronize()
        grads[mode] = opt.flat_grad.detach().clone()
        logs[mode] = list(opt.sync.log)
    opt.step(0)
    # the bench's timing protocol over RCCL: barrier + MAX all-reduce of the elapsed time
    dist.barrier()
    t = torch.tensor([1.25], device="cuda:0")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    q.put({"eq_overlap": bool(torch.equal(grads["overlap"], grads["off"])), "eq_tail": bool(torch.equal(grads["tail"], grads["off"])),
           "eq_check": bool(torch.equal(grads["check"], grads["off"])), "log_overlap": logs["overlap"], "log_tail": logs["tail"],
           "buckets": [b for b, _, _ in opt.buckets], "tmax": float(t.item()), "scale": opt._grad_scale,
           "nonzero": float((grads["off"] != 0).float().mean())})
    dist.destroy_process_group()


@pytest.mark.gpu
def test_bucketed_exchange_over_rccl_one_rank():
    """The RCCL call pattern of the overlapped exchange on real hardware: a one-rank nccl group (RCCL refuses two ranks on one
    device), GradSync forced on - collectives issued from the tensor hooks on the autograd thread and the communication stream,
    work.wait() ordering, barrier + MAX reduce of bench.py.  The sum over one rank is the identity: every mode must return the
    local gradient bit for bit, with the overlapped launch order of the two-rank gloo test."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=900)
    p.join(timeout=120)
    assert not (isinstance(out, tuple) and out[0] == "error"), out[1]
    assert p.exitcode == 0
    assert out["eq_overlap"] and out["eq_tail"] and out["eq_check"] and out["nonzero"] > 0.9, out
    names = out["buckets"]
    want = [[names[-1], "overlapped"]] + [[b, "overlapped"] for b in reversed(names[1:-1])] + [["vfe", "tail"]]
    assert [list(x) for x in out["log_overlap"]] == want, out["log_overlap"]
    assert all(h == "tail" for _, h in out["log_tail"]) and len(out["log_tail"]) == len(names)
    assert out["tmax"] == 1.25


def _ddp_worker(port, q):
    _guard(_ddp_worker_body)(port, q)


def _ddp_worker_body(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=0, world_size=1)
    from pcdet.models import model_fn_decorator
    cfg, ds, net, pts = _build("A", 2, 0)
    out = {}
    # reference gradients: the bare model, plain autograd into p.grad (no flat optimizer)
    torch.manual_seed(0)                                      # masking noise comes from the device RNG
    bd = {"points": pts, "batch_size": 2}
    ret, _, _ = net(bd)
    ret["loss"].backward()
    ref = {n: p.grad.detach().clone() for n, p in net.named_parameters()}
    mask_ref = net.backbone_3d.forward_ret_dict["mask"].clone()
    net.zero_grad(set_to_none=True)
    # the reference's wrappers (tools/train.py:146; build_optimizer -> here a stock Adam)
    ddp = torch.nn.parallel.DistributedDataParallel(net, device_ids=[0], find_unused_parameters=False)
    adam = torch.optim.Adam(ddp.parameters(), lr=1e-3)
    torch.manual_seed(0)
    bd = {"points": pts, "batch_size": 2}
    ret, _, _ = ddp(bd)
    ret["loss"].backward()
    # (DDP hands the module a re-built copy of the input dict, so the outputs are read from the module, not from `bd`)
    same_mask = bool(torch.equal(net.backbone_3d.forward_ret_dict["mask"], mask_ref))
    worst = 0.0
    missing = []
    for n, p in net.named_parameters():
        if p.grad is None:
            missing.append(n)
            continue
        if same_mask:
            den = float(ref[n].abs().max()) + 1e-12
            worst = max(worst, float((p.grad - ref[n]).abs().max()) / den)
    before = [p.detach().clone() for p in net.parameters()]
    adam.step()
    moved = sum(int(not torch.equal(a, p.detach())) for a, p in zip(before, net.parameters()))
    # the reference's model_func on the DDP-wrapped model (numpy batch in, ModelReturn out, global_step advanced)
    fn = model_fn_decorator()
    adam.zero_grad()
    r = fn(ddp, {"points": pts.cpu().numpy(), "batch_size": 2})
    r.loss.backward()
    out.update(missing=missing, worst=worst, same_mask=same_mask, moved=moved, n_params=len(before),
               global_step=int(net.global_step.item()), loss=float(r.loss.detach()))
    q.put(out)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_network_under_torch_ddp_and_stock_adam():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_ddp_worker, args=(_free_port(), q))
    p.start()
    out = q.get(timeout=900)
    p.join(timeout=120)
    assert not (isinstance(out, tuple) and out[0] == "error"), out[1]
    assert p.exitcode == 0
    assert out["missing"] == [], out["missing"]               # find_unused_parameters=False is safe: every parameter gets a gradient
    assert out["same_mask"], "masking noise must be reproducible for the comparison"
    assert out["worst"] <= 1e-5, out                          # DDP-wrapped gradients == bare-model gradients (world 1: mean of one)
    assert out["moved"] >= out["n_params"] - 2                # the stock optimizer updates every tensor (bar exact-zero gradients)
    assert out["global_step"] == 1 and out["loss"] == out["loss"]
