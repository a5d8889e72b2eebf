#!/usr/bin/env python
"""GD-MAE pre-training throughput on MI355X: frames/s of the full training step (forward, backward,
gradient all-reduce, global-norm clip, fused Adam) on synthetic Waymo-shape clouds.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; frames are sharded across ranks (weak scaling: --batch-per-gpu frames each), the only
collective is one all-reduce of the flat 32 MB gradient buffer per step (RCCL over xGMI).  Rank 0 prints
ONE JSON line.  Inputs are resident in HBM when the timed region starts (the PCIe-inclusive rate is
reported separately under "h2d_inclusive").
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "gd-mae_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16
MFMA_F32_PEAK_TF = 157.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=8, help="frames per GPU per step (gd_mae_ssl.yaml:184)")
    ap.add_argument("--config", default="B", choices=["A", "B", "E"])
    ap.add_argument("--mask-ratio", type=float, default=0.75)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--pool", type=int, default=3, help="distinct pre-generated batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--prefetch", type=int, default=1, help="1: build the geometry plan of batch t+1 on a side stream")
    ap.add_argument("--miopen-find", type=int, default=1, help="1: torch.backends.cudnn.benchmark (MIOpen find mode)")
    return ap.parse_args()


def algorithmic_bytes_per_frame(N, M, Ms, ds, G, a):
    """BYTES_FWD of SURVEY.md §8(d) (S1..S5); BYTES_TRAIN = 3 x BYTES_FWD."""
    F = 5
    s1 = N * (1 + F) * 4 + N * 4 + M * (16 + 4 * F)
    s2 = 3 * (N * (1 + F) * 4 + N * 4) + M * 128 * a
    s3 = sum(4 * 10 * m * d * a for m, d in zip(Ms, ds))
    cin = [128] + list(ds[:-1])
    s3 += sum((mi * ci + mo * co) * a for mi, ci, mo, co in zip([Ms[0]] + list(Ms[:-1]), cin, Ms, ds))   # conv_down
    s3 += sum(2 * m * d * a for m, d in zip(Ms, ds))                                                       # conv_out
    s4 = 2208 * G * a
    s5 = M * (128 + 48) * a + N * 12 + 2 * M * 64 * 12
    return s1 + s2 + s3 + s4 + s5


def measure_cpu_baseline(config: str, mask_ratio: float):
    """The CPU oracle (oracle/gdmae_oracle.py + oracle/optim_oracle.py: the reference algorithm restated in
    PyTorch fp32, validated against the imported reference modules) timed on this box's host cores on ONE
    frame of the same workload: forward + backward + optimizer step."""
    from gdmae_hip import configs, synth
    from oracle import gdmae_oracle as orc
    from oracle import optim_oracle as oo
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # PyTorch-CPU slows down badly when oversubscribed (256 threads: 643 s for this sample on the MI355X host),
    # so pick the thread count with a 1-2 s probe (a 128->128 3x3 convolution on a 234x234 map, fwd + bwd)
    best, cores = None, 1
    x = torch.randn(1, 128, 234, 234)
    w = torch.randn(128, 128, 3, 3, requires_grad=True)
    for c in [c for c in (8, 16, 32, 64, 128) if c <= avail] or [avail]:
        torch.set_num_threads(c)
        torch.nn.functional.conv2d(x, w, padding=1).sum().backward()
        t0 = time.perf_counter()
        torch.nn.functional.conv2d(x, w, padding=1).relu().sum().backward()
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, cores = dt, c
    torch.set_num_threads(cores)
    cfg, ds, skw = configs.named_config(config, mask_ratio=mask_ratio)
    F = ds.point_feature_encoder.num_point_features
    pts = torch.from_numpy(synth.synth_batch(424242, 1, ds.point_cloud_range, **skw))
    sd = orc.seeded_state_dict(orc.param_shapes(cfg, F), seed=1, requires_grad=True)
    names = sorted(sd)
    opt = oo.AdamOneCycle([sd[k] for k in names])
    t0 = time.perf_counter()
    o = orc.forward(pts, 1, cfg, sd, ds.point_cloud_range, ds.voxel_size, ds.grid_size, noise_seed=0)
    o["loss"].backward()
    opt.step(*oo.one_cycle(0, 100, 0.003, [0.95, 0.85], 10, 0.4))
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 frame of config {config} ({pts.shape[0]} pts), full train step fwd+bwd+Adam in fp32, {dt:.1f} s, "
                      f"{cores} threads (best of a probe over 8..128; {avail} logical CPUs available)"}


def measure_roofline(step, dev_batches, args, n_steps=4):
    """HIP-event timing of the dominant hand-written kernel over `n_steps` extra steps (after the timed region).

    Dominant HIP kernel of the step (profiles/r01_kernel_stats.csv): the windowed cosine attention backward
    (entry point gdmae_window_attention_bwd = k_win_attn_bwd for the T=16 level + k_attn_mfma16_bwd for T=32/64 with
    bf16 rows; k_attn_mfma_bwd with fp32 rows).
    Algorithmic bytes per launch = tokens_of_level * (7 * d * elem_size + 4) + 8 * windows (read the q, k, v, dOut
    rows, write the dq, dk, dv rows, once each, + CSR), DESIGN.md section 4.  achieved = sum(bytes) / sum(duration)
    over all its launches, timed with HIP events on the launch stream.  `traffic` = measured HBM bytes per launch
    from the committed rocprofv3 PMC passes (2 x FETCH_SIZE + WRITE_SIZE in KiB, profiles/r01_pmc_*_by_kernel.csv):
    it equals the algorithmic bytes, i.e. no re-reads - the kernel is latency / issue bound, not bandwidth bound."""
    from gdmae_hip import timing
    with timing.collect() as T:
        for i in range(n_steps):
            step(args.warmup + args.steps, dev_batches[i % args.pool])
        summ = T.summary()
    k = summ["k_win_attn_bwd"]
    gbs = k["total_bytes"] / (k["total_ms"] * 1e-3) / 1e9
    traffic = None
    try:
        import csv
        tot = {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}
        for tag in tot:
            for r in csv.DictReader(open(os.path.join(REPO, "profiles", f"r01_pmc_{tag}_by_kernel.csv"))):
                if r["kernel"] in ("k_win_attn_bwd", "k_attn_mfma_bwd", "k_attn_mfma16_bwd"):
                    tot[tag][0] += float(r[f"sum_{tag}"])
                    tot[tag][1] += int(r["dispatches"])
        if tot["FETCH_SIZE"][1] and tot["FETCH_SIZE"][1] == tot["WRITE_SIZE"][1]:
            traffic = int((2 * tot["FETCH_SIZE"][0] + tot["WRITE_SIZE"][0]) * 1024 / tot["FETCH_SIZE"][1])
    except Exception:
        traffic = None
    return {"kernel": "gdmae_window_attention_bwd (k_win_attn_bwd + k_attn_mfma16_bwd)", "bound": "hbm", "achieved": round(gbs, 1),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
            "launches_per_step": k["launches"] / n_steps, "avg_launch_us": round(k["avg_us"], 2),
            "algorithmic_bytes_per_launch": int(k["bytes_per_launch"]),
            "also": {n: {"avg_us": round(v["avg_us"], 2), "GBs": round(v["total_bytes"] / (v["total_ms"] * 1e-3) / 1e9, 1)}
                     for n, v in summ.items() if n != "k_win_attn_bwd"}}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    local = local % torch.cuda.device_count()      # several ranks may share a GPU in the 2-rank smoke run (gloo backend)
    torch.cuda.set_device(local)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl == RCCL on ROCm; GDMAE_DIST_BACKEND=gloo only for the control-flow smoke run of two ranks on one GPU
        dist.init_process_group(os.environ.get("GDMAE_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    from gdmae_hip import configs, optim, synth
    from pcdet.models import build_network

    cfg, ds, skw = configs.named_config(args.config, mask_ratio=args.mask_ratio)
    torch.manual_seed(1234)                       # identical initial weights on every rank
    net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("bench")).to(dev).train()
    net.sync_loss_scalar = False                  # keep the loss on the device: no per-step host sync
    net.backbone_3d.dense_spatial_features = False   # the step only consumes decoder rows at the pillar sites
    total_steps = args.warmup + args.steps + 1
    opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(args.batch_per_gpu), total_steps=total_steps)
    n_params = opt.n

    # synthetic frames: frame f of rank r uses seed 1000 * r + f; pool of distinct batches, resident in HBM
    B = args.batch_per_gpu
    # every pooled batch must pass through the untimed warmup once (first-use GEMM algorithm timing, MIOpen find)
    args.pool = max(1, min(args.pool, args.warmup))
    host_batches = [synth.synth_batch(100000 * rank + 97 * k, B, ds.point_cloud_range, **skw) for k in range(args.pool)]
    pinned = [torch.from_numpy(b).pin_memory() for b in host_batches]
    dev_batches = [p.to(dev, non_blocking=True) for p in pinned]
    resident = torch.cuda.Event()
    resident.record()                              # every pooled batch is in HBM once this event has completed
    torch.cuda.synchronize()
    use_bf16 = args.dtype == "bf16"

    pending = {}

    def step(i, pts, nxt=None, nxt_ready=None):
        opt.zero_grad()
        bd = {"points": pts, "batch_size": B}
        if args.prefetch:
            plan = pending.pop(id(pts), None) or net.backbone_3d.prefetch_plan(pts, B).finish()
            bd["_gdmae_vox"], bd["_gdmae_plan"] = plan
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=use_bf16):
            ret, tb, _ = net(bd)
        ret["loss"].backward()
        pf = None
        if args.prefetch and nxt is not None:
            # geometry plan of the NEXT batch: issued while the GPU is still busy with this backward (the host is ahead
            # here), on a side stream that only waits for that batch's points to be resident
            pf = net.backbone_3d.prefetch_plan(nxt, B, ready=nxt_ready)
        opt.all_reduce_grads()
        opt.step(i)
        if pf is not None:
            # ... and collected here, still behind the queued backward / optimizer kernels: the host-side part of the plan
            # (counts to Python ints, ~300 tensor views) is off the next step's critical path, like a data-loader batch
            # that is ready before it is asked for
            pending[id(nxt)] = pf.finish()
        return ret["loss"], bd

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        loss, bd = step(i, dev_batches[i % args.pool], dev_batches[(i + 1) % args.pool] if i + 1 < args.warmup else None, resident)
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, bd = step(args.warmup + i, dev_batches[i % args.pool], dev_batches[(i + 1) % args.pool] if i + 1 < args.steps else None,
                        resident)
    sync_all()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    frames = B * world * args.steps
    fps = frames / dt
    final_loss = float(loss.detach())
    assert np.isfinite(final_loss), "training diverged"

    out = {"metric": "MAE pre-train frames/sec (Waymo-shape, 180k pts, 75% mask)", "value": round(fps, 2),
           "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"config {args.config}: synthetic Waymo-shape clouds ~180k pts x5 feat, 0.32 m pillars (468x468), "
                                  f"GD-MAE SRA encoder 128/256/256 x12 layers + generative decoder, mask {args.mask_ratio}, "
                                  f"full train step (fwd+bwd+grad all-reduce+clip+Adam)" if args.config == "B" else f"config {args.config}",
                      "frames_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                      "params": n_params, "mask_ratio": args.mask_ratio, "loss_last": round(final_loss, 5)}}

    # the roofline steps run on EVERY rank: they contain the gradient all-reduce, a collective the other ranks must join
    roofline = measure_roofline(step, dev_batches, args) if not args.no_roofline else None
    if rank == 0:
        # ---- sizes of the last batch (for the algorithmic byte model)
        vox, ep = bd["_gdmae_vox"], bd["_gdmae_plan"]
        N, M = vox.N / B, vox.M / B
        Ms = [s.n_tok / B for s in ep.stages]
        dsz = [int(b.ENCODER.D_MODEL) for b in cfg.BACKBONE_3D.SST_BLOCK_LIST]
        G = int(ds.grid_size[0]) * int(ds.grid_size[1])
        a = 2 if use_bf16 else 4
        bytes_train = 3 * algorithmic_bytes_per_frame(N, M, Ms, dsz, G, a)
        out["config"].update({"points_per_frame": int(N), "pillars_per_frame": int(M), "tokens_per_frame": [int(m) for m in Ms]})
        out["step_bytes_model"] = {"bytes_train_per_frame": int(bytes_train), "achieved_GBs": round(bytes_train * fps / world / 1e9, 1),
                                   "frac_of_8TBs": round(bytes_train * fps / world / 1e9 / HBM_PEAK_GBS, 4),
                                   "note": "SURVEY §8d whole-step algorithmic bytes x frames/s per GPU"}
        if roofline is not None:
            out["roofline"] = roofline
        # ---- PCIe-inclusive variant (host buffers handed over every step); never the headline value
        sync_all_local = torch.cuda.synchronize
        sync_all_local()
        if world == 1:
            t1 = time.perf_counter()
            k = max(3, args.steps // 4)
            nxt = pinned[0].to(dev, non_blocking=True)
            for i in range(k):
                cur, nxt, ev = nxt, None, None
                if i + 1 < k:                          # next batch's H2D copy is queued before this step's kernels
                    nxt = pinned[(i + 1) % args.pool].to(dev, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                step(args.warmup + args.steps, cur, nxt, ev)
            sync_all_local()
            out["h2d_inclusive"] = {"value": round(B * k / (time.perf_counter() - t1), 2), "unit": "frames/s"}
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = measure_cpu_baseline(args.config, args.mask_ratio)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
