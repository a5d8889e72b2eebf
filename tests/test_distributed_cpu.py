"""N > 1 path on CPU: two gloo ranks, flat gradient buffer all-reduce (the only collective of the step)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gdmae_hip import configs, optim, synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1234)                                   # same init on every rank, as bench.py does
    model = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 2))
    opt = optim.FlatAdamOneCycle(model, configs.optimization_cfg(), total_steps=10)
    # parameters and gradients are views into the flat buffers
    assert all(p.data_ptr() >= opt.flat_param.data_ptr() for p in model.parameters())
    opt.zero_grad()
    x = torch.full((4, 6), float(rank + 1))
    model(x).sum().backward()                                 # autograd accumulates straight into flat_grad
    local = opt.flat_grad.clone()
    opt.all_reduce_grads()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    q.put((rank, opt.flat_grad.numpy().copy(), torch.stack(gathered).mean(0).numpy(), opt.flat_param.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, g0, m0, p0), (r1, g1, m1, p1) = res
    assert np.allclose(g0, m0, rtol=1e-6) and np.array_equal(g0, g1)       # averaged, identical on both ranks
    assert np.array_equal(p0, p1)                                            # replicas start identical
    assert np.abs(g0).sum() > 0


def test_ranks_get_disjoint_frames():
    _, ds, kw = configs.named_config("A")
    a = synth.synth_batch(100000 * 0 + 0, 2, ds.point_cloud_range, **kw)
    b = synth.synth_batch(100000 * 1 + 0, 2, ds.point_cloud_range, **kw)
    assert a.shape[1] == b.shape[1] == 5 and not np.array_equal(a[:100], b[:100])
    assert set(np.unique(a[:, 0])) == {0.0, 1.0}                             # frame index in column 0
