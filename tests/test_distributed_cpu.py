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
    opt.all_reduce_grads()                                    # SUM; the 1/world average is folded into the Adam launch
    assert opt._grad_scale == 1.0 / world
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    q.put((rank, opt.flat_grad.numpy().copy(), torch.stack(gathered).sum(0).numpy(), opt.flat_param.numpy().copy()))
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
    assert np.allclose(g0, m0, rtol=1e-6) and np.array_equal(g0, g1)       # summed, identical on both ranks
    assert np.array_equal(p0, p1)                                            # replicas start identical
    assert np.abs(g0).sum() > 0


def test_ranks_get_disjoint_frames():
    _, ds, kw = configs.named_config("A")
    a = synth.synth_batch(100000 * 0 + 0, 2, ds.point_cloud_range, **kw)
    b = synth.synth_batch(100000 * 1 + 0, 2, ds.point_cloud_range, **kw)
    assert a.shape[1] == b.shape[1] == 5 and not np.array_equal(a[:100], b[:100])
    assert set(np.unique(a[:, 0])) == {0.0, 1.0}                             # frame index in column 0


class _Staged(torch.nn.Module):
    """Toy model with the bucket structure of GDMAE (vfe -> backbone_3d.sst_blocks.{0,1} -> backbone_3d.decoder_*) that
    marks its stage inputs the way SPTBackboneMAE.forward does."""

    def __init__(self):
        super().__init__()
        self.vfe = torch.nn.Linear(6, 8)
        self.backbone_3d = torch.nn.Module()
        self.backbone_3d.sst_blocks = torch.nn.ModuleList([torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)])
        self.backbone_3d.decoder_pred = torch.nn.Linear(8, 3)

    def forward(self, x, sync):
        x = self.vfe(x)
        for i, blk in enumerate(self.backbone_3d.sst_blocks):
            sync.mark(x, [f"backbone_3d.sst_blocks.{i}"])
            x = torch.relu(blk(x))
        sync.mark(x, ["backbone_3d.decoder"])
        return self.backbone_3d.decoder_pred(x)


def _overlap_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1234)
    model = _Staged()
    opt = optim.FlatAdamOneCycle(model, configs.optimization_cfg(), total_steps=10)
    out = []
    # iteration 0 of a job that exchanges gradients is an automatic 'check' step (GradSync.autocheck): snapshots at the hooks are
    # compared with the final local gradients and everything is reduced at the tail; the overlapped exchange starts with step 1
    assert opt.sync.autocheck and opt.sync.checked_steps == 0
    opt.zero_grad()
    model(torch.randn(5, 6, generator=torch.Generator().manual_seed(77 + rank)), opt.sync).square().sum().backward()
    opt.all_reduce_grads()
    assert opt.sync.checked_steps == 1 and [how for _, how in opt.sync.log] == ["tail"] * 4, opt.sync.log
    for it in range(2):                                       # two steps: the per-step state of GradSync resets
        opt.zero_grad()
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 * it + rank))
        seen = []
        orig = opt.sync.reduce

        def spy(names, how, _orig=orig, _seen=seen):
            # at the time a bucket is handed to the collective its gradient must be final: snapshot it
            for b, lo, hi in opt.buckets:
                if b in names and b not in opt.sync.launched:
                    _seen.append((b, how, opt.flat_grad[lo:hi].clone()))
            return _orig(names, how)
        opt.sync.reduce = spy
        model(x, opt.sync).square().sum().backward()
        opt.all_reduce_grads()
        opt.sync.reduce = orig
        out.append((list(opt.sync.log), [(b, how, g.numpy()) for b, how, g in seen], opt.flat_grad.numpy().copy()))
    # single-process reference of the local gradients
    ref = []
    torch.manual_seed(1234)
    m2 = _Staged()
    for it in range(2):
        m2.zero_grad()
        x = torch.randn(5, 6, generator=torch.Generator().manual_seed(10 * it + rank))

        class _NoSync:
            def mark(self, *a):
                pass
        m2(x, _NoSync()).square().sum().backward()
        table = opt._offsets()
        flat = np.zeros(opt.n, dtype=np.float32)
        for (n1, p1), (n2, p2) in zip(model.named_parameters(), m2.named_parameters()):
            o, k = table[id(p1)]
            flat[o:o + k] = p2.grad.reshape(-1).numpy()
        ref.append(flat)
    q.put((rank, out, ref, [(b, lo, hi) for b, lo, hi in opt.buckets]))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_is_launched_from_inside_the_backward_two_ranks():
    """N > 1 step ordering on CPU (gloo, 2 ranks): the stage / decoder buckets are handed to the collective from tensor
    hooks INSIDE backward() in completion order (decoder, stage 1, stage 0), each with its final local gradient, only the
    VFE bucket is left for the tail; the result is the sum over ranks of the single-process gradients."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_overlap_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, out0, ref0, buckets), (_, out1, ref1, _) = res
    assert [b for b, _, _ in buckets] == ["vfe", "backbone_3d.sst_blocks.0", "backbone_3d.sst_blocks.1", "backbone_3d.decoder"]
    for it in range(2):
        log, seen, summed = out0[it]
        assert log == [("backbone_3d.decoder", "overlapped"), ("backbone_3d.sst_blocks.1", "overlapped"),
                       ("backbone_3d.sst_blocks.0", "overlapped"), ("vfe", "tail")]
        assert out1[it][0] == log
        for r, (out_r, ref_r) in enumerate(((out0, ref0), (out1, ref1))):
            for b, how, g in out_r[it][1]:                      # what each bucket held when it was handed over
                lo, hi = next((lo, hi) for bb, lo, hi in buckets if bb == b)
                assert np.allclose(g, ref_r[it][lo:hi], rtol=1e-5, atol=1e-7), (r, b, how)
        assert np.allclose(summed, ref0[it] + ref1[it], rtol=1e-5, atol=1e-7)
        assert np.array_equal(summed, out1[it][2])
