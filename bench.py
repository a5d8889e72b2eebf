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

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver (set before HIP starts)
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
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch-per-gpu", type=int, default=8, help="frames per GPU per step (gd_mae_ssl.yaml:184)")
    ap.add_argument("--config", default="B", choices=["A", "B", "E"])
    ap.add_argument("--mask-ratio", type=float, default=0.75)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--pool", type=int, default=3, help="distinct pre-generated batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra fp32 / mask-0.85 / H2D-inclusive legs")
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
    # PyTorch-CPU slows down badly when oversubscribed on this workload (sparse gathers, many small ops): measured on the MI355X
    # host (256 logical CPUs) one step takes 5.1 s with 32 threads, 7.4 s with 64, 21 s with 128 and 643 s with 256 - a
    # convolution micro-probe picked 128 on some boxes, so the thread count is fixed at the best measured one
    cores = min(32, avail)
    torch.set_num_threads(cores)
    cfg, ds, skw = configs.named_config(config, mask_ratio=mask_ratio)
    F = ds.point_feature_encoder.num_point_features
    pts = torch.from_numpy(synth.synth_batch(424242, 1, ds.point_cloud_range, **skw))
    sd = orc.seeded_state_dict(orc.param_shapes(cfg, F), seed=1, requires_grad=True)
    names = sorted(sd)
    opt = oo.AdamOneCycle([sd[k] for k in names])
    times = []
    for it in range(4):                               # one un-timed warm-up (allocator, thread pools), then 3 samples
        for k in names:
            sd[k].grad = None
        t0 = time.perf_counter()
        o = orc.forward(pts, 1, cfg, sd, ds.point_cloud_range, ds.voxel_size, ds.grid_size, noise_seed=0)
        o["loss"].backward()
        opt.step(*oo.one_cycle(it, 100, 0.003, [0.95, 0.85], 10, 0.4))
        if it:
            times.append(time.perf_counter() - t0)
    dt = sorted(times)[1]
    return {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"1 frame of config {config} ({pts.shape[0]} pts), full train step fwd+bwd+Adam in fp32; median of 3 after one "
                      f"warm-up step ({', '.join('%.1f' % t for t in times)} s), {cores} threads (best of 16 / 32 / 64 / 128 measured "
                      f"on the MI355X host; {avail} logical CPUs available)"}


def _pmc_traffic(kernels):
    """HBM bytes per launch of `kernels` from the committed rocprofv3 PMC passes of the current round (separate --pmc runs of
    this same command, tools/collect_profiles.sh): 2 x FETCH_SIZE + WRITE_SIZE (KiB; the x2 is the guide's gfx950
    correction for wide coalesced reads).  None when the profile files are not there."""
    import csv
    import glob
    try:
        tags = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_FETCH_SIZE_by_kernel.csv")))
        if not tags:
            return None, None
        fetch = tags[-1]
        write = fetch.replace("FETCH_SIZE", "WRITE_SIZE")
        tot = {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}
        for tag, path in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            for r in csv.DictReader(open(path)):
                if r["kernel"] in kernels:
                    tot[tag][0] += float(r[f"sum_{tag}"])
                    tot[tag][1] += int(r["dispatches"])
        if tot["FETCH_SIZE"][1] and tot["FETCH_SIZE"][1] == tot["WRITE_SIZE"][1]:
            return int((2 * tot["FETCH_SIZE"][0] + tot["WRITE_SIZE"][0]) * 1024 / tot["FETCH_SIZE"][1]), os.path.relpath(fetch, REPO)
    except Exception:
        pass
    return None, None


# instrumented entry points (gdmae_hip.timing brackets): what bounds them and which device kernels they launch
ROOFLINE_KERNELS = {
    "k_conv3x3_tiles": ("mfma", ("k_conv3x3_tiles",)),
    "k_conv_grad_taps": ("hbm", ("k_conv_grad_taps",)),
    "k_win_attn_bwd": ("hbm", ("k_win_attn_bwd", "k_attn_mfma_bwd", "k_attn_t16_bwd", "k_attn_t32_bwd", "k_attn_t64_bwd")),
    "k_win_attn_fwd": ("hbm", ("k_win_attn_fwd", "k_attn_mfma_fwd", "k_attn_t16_fwd", "k_attn_t32_fwd", "k_attn_t64_fwd")),
}


def _attention_product_timing(step, dev_batches, args, n_steps):
    """{entry: summary} of the forward / backward all-levels attention entries over `n_steps` product-path steps: HIP events
    recorded by the library around the calls (include/gdmae_hip.h gdmae_attention_timing), algorithmic bytes from the plans of
    those steps (tokens x (4 | 7) x d x 2 B + CSR, per layer: every stage runs 2 layers on each of its 2 window shifts)."""
    import ctypes as C
    from gdmae_hip import lib as L
    if not hasattr(L.load(), "gdmae_attention_timing"):
        return None
    L.call("gdmae_attention_timing", 1)
    by = {"k_win_attn_fwd": 0.0, "k_win_attn_bwd": 0.0}
    try:
        for i in range(n_steps):
            _, bd = step(args.warmup + args.steps, dev_batches[i % args.pool])
            plan = bd["_gdmae_plan"]
            for st, dm in zip(plan.stages, bd["_gdmae_dims"]):
                for w in st.windows:
                    for lvl, nw in enumerate(w.n_win):
                        if nw > 0:
                            by["k_win_attn_fwd"] += 2 * (w.n_tok[lvl] * (4 * dm * 2 + 4) + 8 * nw)
                            by["k_win_attn_bwd"] += 2 * (w.n_tok[lvl] * (7 * dm * 2 + 4) + 8 * nw)
        torch.cuda.synchronize()
        out = {}
        for which, name in ((0, "k_win_attn_fwd"), (1, "k_win_attn_bwd")):
            ms, calls = C.c_double(0.0), C.c_longlong(0)
            L.call("gdmae_attention_timing_read", which, C.byref(ms), C.byref(calls))
            if calls.value:
                out[name] = {"launches": calls.value, "total_ms": ms.value, "avg_us": 1e3 * ms.value / calls.value,
                             "bytes_per_launch": by[name] / calls.value, "total_bytes": by[name], "flops_per_launch": 0.0,
                             "total_flops": 0.0, "extra": {"timed": "product path: one call per layer = all occupancy levels"}}
        return out
    finally:
        L.call("gdmae_attention_timing", 0)


def measure_roofline(step, dev_batches, args, n_steps=4):
    """HIP-event timing (events on the launch stream) of the instrumented hand-written kernels over `n_steps` extra steps
    after the timed region; `roofline` describes the one with the LARGEST TOTAL TIME in this run, the others are listed
    under "also".  Algorithmic work per launch:
      k_conv3x3_tiles   executed MFMA flops = active tiles x 64 sites x 128 channels x (9 x Cin) x 2   (bound: bf16 MFMA);
                        the dense convolution of SURVEY section 8d would be B x H x W sites - reported as dense_equivalent
      k_conv_grad_taps  active sites x 9 taps x 128 channels x 2 B read + written
      k_win_attn_*      tokens x (7 | 4) x d x elem + CSR bytes (q, k, v, dOut rows read, dq, dk, dv rows written, once each);
                        one "launch" = one call of the all-levels entry (a layer's T = 16 launch + its T = 32 / 64 launch)
    `traffic` = HBM bytes per launch from this round's committed rocprofv3 PMC passes (profiles/, see _pmc_traffic)."""
    from gdmae_hip import timing
    with timing.collect() as T:
        for i in range(n_steps):
            step(args.warmup + args.steps, dev_batches[i % args.pool])
        summ = T.summary()
    summ = {k: v for k, v in summ.items() if k in ROOFLINE_KERNELS and v["total_ms"] > 0}
    # the attention entry points are timed on the PRODUCT path (the per-operator path above launches every occupancy level on its
    # own; the layer executor issues a layer's levels through gdmae_window_attention_levels_*, bracketed with HIP events there)
    prod = _attention_product_timing(step, dev_batches, args, n_steps)
    if prod:
        summ.update(prod)
    if not summ:
        return None

    def describe(name, v):
        bound, devk = ROOFLINE_KERNELS[name]
        sec = v["total_ms"] * 1e-3
        if bound == "mfma":
            ach, peak, unit = v["total_flops"] / sec / 1e12, MFMA_BF16_PEAK_TF, "TFLOP/s"
        else:
            ach, peak, unit = v["total_bytes"] / sec / 1e9, HBM_PEAK_GBS, "GB/s"
        d = {"kernel": name, "bound": bound, "achieved": round(ach, 1), "peak": peak, "unit": unit, "frac": round(ach / peak, 4),
             "launches_per_step": v["launches"] / n_steps, "avg_launch_us": round(v["avg_us"], 2),
             "total_us_per_step": round(1e3 * v["total_ms"] / n_steps, 1)}
        if bound == "mfma":
            d["algorithmic_flops_per_launch"] = int(v["flops_per_launch"])
        else:
            d["algorithmic_bytes_per_launch"] = int(v["bytes_per_launch"])
        for k2, val in v.get("extra", {}).items():
            d[k2] = val
        return d, devk

    top = max(summ, key=lambda k: summ[k]["total_ms"])
    out, devk = describe(top, summ[top])
    out["traffic"], src = _pmc_traffic(devk)
    if src:
        out["traffic_source"] = src + " (2 x FETCH_SIZE + WRITE_SIZE per launch, committed rocprofv3 --pmc passes of this command)"
    out["also"] = {n: describe(n, v)[0] for n, v in summ.items() if n != top}
    return out


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
    if world > 1:
        # create the communicator HERE, on the main thread and the current stream: the bucketed gradient exchange issues its
        # first collectives from tensor hooks inside backward() (autograd thread, communication stream)
        warm = torch.ones(1, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize()
        assert int(warm.item()) == world

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
    mode = {"bf16": args.dtype == "bf16"}
    use_bf16 = mode["bf16"]

    pending = {}
    stage_dims = [int(b.ENCODER.D_MODEL) for b in cfg.BACKBONE_3D.SST_BLOCK_LIST]

    def step(i, pts, nxt=None, nxt_ready=None):
        opt.zero_grad()
        bd = {"points": pts, "batch_size": B, "_gdmae_grad_sync": opt.sync, "_gdmae_dims": stage_dims}
        if args.prefetch:
            plan = pending.pop(id(pts), None) or net.backbone_3d.prefetch_plan(pts, B).finish()
            bd["_gdmae_vox"], bd["_gdmae_plan"] = plan
        pf = None
        if args.prefetch and nxt is not None:
            # geometry plan of the NEXT batch: issued at the start of the step on a side stream that is ordered after the work
            # queued so far (the previous step) - its ~100 small kernels run under this step's forward and are complete long
            # before the host asks for them at the end of the step
            pf = net.backbone_3d.prefetch_plan(nxt, B, ready=nxt_ready)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode["bf16"]):
            ret, tb, _ = net(bd)
        ret["loss"].backward()
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
    vox, ep = bd["_gdmae_vox"], bd["_gdmae_plan"]

    def timed_leg(n_warm, n_timed, feed=None):
        """frames/s of `n_timed` more steps after `n_warm` untimed ones (single process; used for the extra legs)."""
        pending.clear()
        for i in range(n_warm):
            step(args.warmup + args.steps, dev_batches[i % args.pool])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        if feed is None:
            for i in range(n_timed):
                step(args.warmup + args.steps, dev_batches[i % args.pool],
                     dev_batches[(i + 1) % args.pool] if i + 1 < n_timed else None, resident)
        else:
            feed(n_timed)
        torch.cuda.synchronize()
        dtl = time.perf_counter() - t1
        return {"value": round(B * n_timed / dtl, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dtl / n_timed, 3), "steps": n_timed}

    also = {}
    if world == 1 and not args.no_also:
        # ---- SURVEY section 8d step incl. the H2D copy of the point batch: pinned host buffers handed over every step, the
        #      copy of batch t+1 queued before the kernels of batch t (never the headline value: inputs there are resident)
        copy_stream = torch.cuda.Stream(device=dev)

        def fetch(j):
            # the copy of batch j runs on its own stream (xGMI/PCIe DMA next to the compute of batch j - 1); the plan stream
            # and the compute stream wait for its event only
            with torch.cuda.stream(copy_stream):
                t = pinned[j % args.pool].to(dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            return t, ev

        def h2d_feed(k):
            nxt, nev = fetch(0)
            for i in range(k):
                cur, cev = nxt, nev
                torch.cuda.current_stream().wait_event(cev)
                cur.record_stream(torch.cuda.current_stream())
                nxt, nev = fetch(i + 1) if i + 1 < k else (None, None)
                step(args.warmup + args.steps, cur, nxt, nev)
        also["h2d_inclusive"] = timed_leg(2, args.steps, h2d_feed)
        also["h2d_inclusive"]["note"] = "same step with the 4.3 MB/frame H2D copy of every batch inside the timed region (SURVEY 8d), on a copy stream"
        # ---- the reference yaml's own mask ratio (tools/cfgs/waymo_models/gd_mae_ssl.yaml:158)
        other = 0.85 if abs(args.mask_ratio - 0.85) > 1e-6 else 0.75
        net.backbone_3d.mask_ratio = other
        also[f"mask_{other}"] = timed_leg(3, max(5, args.steps // 2))
        net.backbone_3d.mask_ratio = args.mask_ratio
        # ---- fp32 parity mode (the mode whose loss is held to 1e-4 of the reference by tests/)
        if mode["bf16"]:
            mode["bf16"] = False
            also["fp32_parity_mode"] = timed_leg(3, max(4, args.steps // 4))
            also["fp32_parity_mode"]["note"] = "no autocast: fp32 rows / GEMMs, dense F.conv2d decoder convolution"
            mode["bf16"] = True
        pending.clear()
    if rank == 0:
        # ---- sizes of the last timed batch (for the algorithmic byte model)
        N, M = vox.N / B, vox.M / B
        Ms = [s.n_tok / B for s in ep.stages]
        dsz = [int(b.ENCODER.D_MODEL) for b in cfg.BACKBONE_3D.SST_BLOCK_LIST]
        G = int(ds.grid_size[0]) * int(ds.grid_size[1])
        a = 2 if use_bf16 else 4
        bytes_train = 3 * algorithmic_bytes_per_frame(N, M, Ms, dsz, G, a)
        out["config"].update({"points_per_frame": int(N), "pillars_per_frame": int(M), "tokens_per_frame": [int(m) for m in Ms]})
        if ep.dec_tiles is not None:
            nt = B * ((int(ds.grid_size[1]) + 7) // 8) * ((int(ds.grid_size[0]) + 7) // 8)
            out["config"]["decoder_active_tiles"] = f"{ep.dec_tiles.n_act} of {nt}"
        out["step_bytes_model"] = {"bytes_train_per_frame": int(bytes_train), "achieved_GBs": round(bytes_train * fps / world / 1e9, 1),
                                   "frac_of_8TBs": round(bytes_train * fps / world / 1e9 / HBM_PEAK_GBS, 4),
                                   "note": "SURVEY 8d whole-step algorithmic bytes (dense-decoder formula) x frames/s per GPU"}
        if world > 1:
            out["grad_sync"] = {"buckets": [[b, hi - lo] for b, lo, hi in opt.buckets], "last_step": opt.sync.log,
                                "note": "one all-reduce per bucket; 'overlapped' = launched on the communication stream from inside "
                                        "backward(), 'tail' = after it"}
        if roofline is not None:
            out["roofline"] = roofline
        shares = os.path.join(REPO, "profiles", "class_shares.json")
        if os.path.exists(shares):
            out["kernel_time_shares"] = json.load(open(shares))
        if also:
            out["also"] = also
            out["h2d_inclusive"] = also["h2d_inclusive"]
        if not args.no_cpu_baseline and world == 1:      # reported at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = measure_cpu_baseline(args.config, args.mask_ratio)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
