#!/usr/bin/env python
"""GD-MAE pre-training throughput on MI355X: frames/s of the full training step (forward, backward,
gradient all-reduce, global-norm clip, fused Adam) on synthetic Waymo-shape clouds.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; frames are sharded across ranks (weak scaling: --batch-per-gpu frames each), the only
collective is the bucketed all-reduce of the flat 32 MB gradient buffer, launched from inside backward (RCCL over
xGMI).  Rank 0 prints ONE JSON line.

Timed region: the batch of step t is resident in HBM when step t starts - its host-to-device copy
was issued on a copy stream during step t - 1, the way a pin_memory data loader feeds the reference - so every
batch consumed after the first one is copied INSIDE the timed region, overlapped with compute.  The same loop fed
from batches that never leave HBM is reported under also.resident_inputs.

Defaults: N = 1 -> BASELINE config B (8 frames per GPU, the reference yaml's BATCH_SIZE_PER_GPU); N > 1 -> BASELINE
config C (global batch 32 on 8 GPUs = 4 frames per GPU, also used for its 2 / 4 GPU scaling points).
"""
from __future__ import annotations

import argparse
import json
import logging
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this driver (set before HIP starts)
os.environ.setdefault("GDMAE_NO_LIBRARY", "1")            # csrc/gemm.hip: a product that would reach hipBLASLt fails instead (every leg of this file runs on own kernels)
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
    ap.add_argument("--batch-per-gpu", type=int, default=None,
                    help="frames per GPU per step; default 8 at every N (the reference's BATCH_SIZE_PER_GPU, gd_mae_ssl.yaml:184: weak scaling, "
                         "global batch 8 N); 4 at N = 8 is BASELINE config C's global batch 32")
    ap.add_argument("--config", default="B", choices=["A", "B", "D", "E"])
    ap.add_argument("--mask-ratio", type=float, default=0.75)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--pool", type=int, default=3, help="distinct pre-generated batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the extra legs (resident inputs, fp32, mask 0.85, config E, config-C batch)")
    ap.add_argument("--legs", choices=("default", "full"), default="default",
                    help="kept for command-line compatibility: every leg, config D (fine-tune) included, is part of the default run since its "
                         "dense convolutions are the library's own (round 5)")
    ap.add_argument("--feed", default="h2d", choices=["h2d", "resident"], help="h2d: every batch copied from pinned host memory inside the timed region")
    ap.add_argument("--prefetch", type=int, default=1, help="1: build the geometry plan of batch t+1 on a side stream")
    ap.add_argument("--plan-at", default="conv", choices=["start", "conv", "bwd"],
                    help="where in step t the plan of batch t+1 is issued: at the start of the step, or right before the decoder's tile convolution")
    ap.add_argument("--miopen-find", type=int, default=1, help="1: torch.backends.cudnn.benchmark (MIOpen find mode)")
    ap.add_argument("--autograd-thread", type=int, default=0,
                    help="0 (default): torch.autograd.set_multithreading_enabled(False) - the ~60 hand-written backward nodes of a step run on "
                         "the calling thread (0.8 ms less host time per step when issued into an idle device: 3.0 vs 3.8 ms, "
                         "tools/host_profile.py; no effect on the GPU-bound step); 1: torch's autograd worker thread (torch's default)")
    return ap.parse_args()


def algorithmic_bytes_per_frame(N, M, Ms, ds, G, a):
    """BYTES_FWD of SURVEY.md §8(d) (S1..S5); BYTES_TRAIN = 3 x BYTES_FWD.  G = decoder sites per frame: every BEV cell for the
    reference's dense decoder, the sites of the active 8 x 8 tiles for the sparse-aware decoder this build runs."""
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
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(1.0 / dt, 4), "unit": "frames/s", "cores": cores, "cpu_model": model, "kind": "port",
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


# instrumented kernel families: roofline that bounds them and the device kernels behind them.  Every family is timed with HIP
# events on the launch stream ON THE PRODUCT PATH: the encoder families by brackets inside the native layer / stage executors
# (include/gdmae_hip.h gdmae_kernel_timing), the decoder kernels by brackets around their single C-ABI call (gdmae_hip.timing)
ROOFLINE_KERNELS = {
    "k_tok_gemm": ("hbm", ("k_tok_gemm", "k_tok_gemm_multi", "k_tok_ffn", "k_layer_fwd", "k_layer_bwd_ffn", "k_layer_bwd_in", "k_ln2_bwd_top")),
    "k_dw_grouped": ("hbm", ("k_dw_grouped",)),
    "k_rows_gemm": ("hbm", ("k_rows_gemm", "k_rows_gemm_kc", "k_pred_fwd", "k_pred_bwd_input")),
    "k_layer_tail": ("hbm", ("k_layer_tail",)),
    "k_conv3x3_tiles": ("mfma", ("k_conv3x3_tiles",)),
    "k_conv_grad_taps": ("hbm", ("k_conv_grad_taps",)),
    "k_dec_conv_bwd": ("hbm", ("k_dec_conv_bwd",)),
    "k_spconv_fwd": ("hbm", ("k_spconv_fwd",)),
    "k_spconv_bwd": ("hbm", ("k_spconv_bwd",)),
    "k_win_attn_bwd": ("hbm", ("k_win_attn_bwd", "k_attn_mfma_bwd", "k_attn_t16_bwd", "k_attn_t32_bwd", "k_attn_t64_bwd")),
    "k_win_attn_fwd": ("hbm", ("k_win_attn_fwd", "k_attn_mfma_fwd", "k_attn_t16_fwd", "k_attn_t32_fwd", "k_attn_t64_fwd")),
}


def _library_slots():
    """{family: summary} of the measurement slots of libgdmae_hip.so (brackets inside the native executors)."""
    import ctypes as C
    from gdmae_hip import lib as L
    lib = L.load()
    out = {}
    for slot in range(lib.gdmae_kernel_timing_slots()):
        name = lib.gdmae_kernel_timing_name(slot).decode()
        if not name:
            continue
        ms, calls, by, fl = C.c_double(0.0), C.c_longlong(0), C.c_double(0.0), C.c_double(0.0)
        L.call("gdmae_kernel_timing_read", slot, C.byref(ms), C.byref(calls), C.byref(by), C.byref(fl))
        if calls.value:
            side = C.c_double(0.0)
            L.call("gdmae_kernel_timing_read_side", slot, C.byref(side))
            out[name] = {"launches": calls.value, "total_ms": ms.value, "avg_us": 1e3 * ms.value / calls.value,
                         "bytes_per_launch": by.value / calls.value, "total_bytes": by.value, "side_bytes": side.value,
                         "flops_per_launch": fl.value / calls.value, "total_flops": fl.value, "extra": {}}
    return out


def measure_roofline(step, feed, n_steps=4):
    """HIP-event timing (events on the launch stream) of the instrumented kernel families over `n_steps` extra steps of the
    PRODUCT path after the timed region; `roofline` describes the family with the LARGEST TOTAL TIME in this run (the
    dominant kernel), the others are listed under "also".  Algorithmic work per launch:
      k_tok_gemm        every fused token-GEMM launch of the encoder (k_tok_gemm, k_tok_gemm_multi): operand rows read once,
                        result rows written once, the packed weight image once - summed per launch inside the library
      k_dw_grouped      rows x (M + N) x 2 B per weight gradient + the fp32 partial tiles
      k_conv3x3_tiles   executed MFMA flops = active tiles x 64 sites x 128 channels x (9 x Cin) x 2   (bound: bf16 MFMA);
                        the dense convolution of SURVEY section 8d would be B x H x W sites - reported as dense_equivalent
      k_conv_grad_taps  active sites x 9 taps x 128 channels x 2 B read + written
      k_win_attn_*      tokens x (7 | 4) x d x elem + CSR bytes (q, k, v, dOut rows read, dq, dk, dv rows written, once each);
                        one "launch" = one call of the all-levels entry (a layer's T = 16 launch + its T = 32 / 64 launch)
    `traffic` = HBM bytes per launch from this round's committed rocprofv3 PMC passes (profiles/, see _pmc_traffic)."""
    from gdmae_hip import lib as L
    from gdmae_hip import timing
    L.call("gdmae_kernel_timing", 1)
    try:
        with timing.collect() as T:
            feed(n_steps)
            summ = T.summary()
        summ.update(_library_slots())
    finally:
        L.call("gdmae_kernel_timing", 0)
    summ = {k: v for k, v in summ.items() if k in ROOFLINE_KERNELS and v["total_ms"] > 0}
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
            if name == "k_dw_grouped":
                # what the workgroups request from L2: every job's G and X rows (the nine tap jobs of a gathered launch read the same G
                # nine times) - divided by the HBM peak this is NOT a roofline fraction, it is listed for what it is
                sb = v.get("side_bytes", 0.0)
                d["l2_stream_bytes_per_launch"] = int(sb / max(v["launches"], 1))
                d["l2_stream_GBs"] = round(sb / sec / 1e9, 1)
                d["algorithmic_bytes_note"] = "every distinct operand matrix of a launch once (G and X of the nine tap jobs counted once) + index columns + fp32 partial tiles"
            if name in ("k_win_attn_fwd", "k_win_attn_bwd"):
                # beside the minimum rows of an attention pass (4 d forward, 7 d backward per token): the log-sum-exp rows the forward
                # leaves, and the forward's output rows + those lse rows + the dtau partial slots the backward moves instead of re-deriving
                # the softmax statistics (the PMC traffic of the backward is 1.40 x the 7 d model, 1.2 x with these)
                sb = v.get("side_bytes", 0.0)
                d["side_stream_bytes_per_launch"] = int(sb / max(v["launches"], 1))
                d["achieved_incl_side_streams"] = round((v["total_bytes"] + sb) / sec / 1e9, 1)
            if name == "k_tok_gemm":
                # what the family's launches move BESIDES the bf16 operand / result rows and weight images the fraction is computed
                # from: fp32 statistics rows, per-workgroup partial rows, fp32 rows at the stage boundary, the y + pos copies
                sb = v.get("side_bytes", 0.0)
                d["side_stream_bytes_per_launch"] = int(sb / max(v["launches"], 1))
                d["algorithmic_GB_per_step"] = round(v["total_bytes"] / n_steps / 1e9, 3)
                d["side_stream_GB_per_step"] = round(sb / n_steps / 1e9, 3)
                d["achieved_incl_side_streams"] = round((v["total_bytes"] + sb) / sec / 1e9, 1)
                d["kernels"] = ("k_layer_fwd / k_layer_bwd_ffn / k_layer_bwd_in / k_ln2_bwd_top (csrc/layer_fused.hip: three fused launches per "
                                "encoder layer and direction around the attention, bf16 residual stream; the first workgroups of k_layer_bwd_in "
                                "carry the layer's closing reductions - split-K partial tiles, LayerNorm partial rows, dtau: their bytes "
                                "are side-stream bytes here) + k_tok_gemm_multi (q / k / v projections)")
        for k2, val in v.get("extra", {}).items():
            d[k2] = val
        return d, devk

    top = max(summ, key=lambda k: summ[k]["total_ms"])
    out, devk = describe(top, summ[top])
    out["timed"] = ("HIP events on the launch stream, product path (native stage executor), %d steps after the timed region; the next "
                    "batch's geometry plan (side stream) is issued at the start of these steps, not under the tile convolution as in the "
                    "timed region" % n_steps)
    out["traffic"], src = _pmc_traffic(devk)
    if src:
        out["traffic_source"] = ("PROFILING BOX, not this run: " + src + " (2 x FETCH_SIZE + WRITE_SIZE per launch, committed rocprofv3 --pmc "
                                 "passes of this command - PMC counters cannot be collected inside a timed bench run)")
    out["also"] = {n: describe(n, v)[0] for n, v in summ.items() if n != top}
    return out


WORKLOADS = {
    "A": "config A: synthetic KITTI-shape 20k-pt clouds, 0.32 m pillars, 2-layer SRA encoder d=128 + generative decoder",
    "B": "config B: synthetic Waymo-shape clouds ~180k pts x5 feat, 0.32 m pillars (468x468), GD-MAE SRA encoder 128/256/256 x12 "
         "layers + generative decoder",
    "C": "config C: config B at global batch 32 on 8 GPUs = 4 frames per GPU (--batch-per-gpu 4), DDP-style bucketed RCCL gradient "
         "all-reduce overlapped with backward",
    "D": "config D: synthetic KITTI-shape 20k-pt clouds, 0.16 m pillars (432x496), fine-tune step SPTBackbone + SSTBEVBackbone + "
         "CenterHead (CenterPoint), synthetic boxes",
    "E": "config E: synthetic ONCE-shape 60k-pt clouds, 0.32 m pillars, 6-layer SRA d=256 + generative decoder",
}


class Workload:
    """One model + optimizer + pool of pinned host batches, and the training step over them."""

    def __init__(self, args, config, B, dev, rank, world, mask_ratio, total_steps, pool):
        from gdmae_hip import configs, optim, synth
        from pcdet.models import build_network
        self.args, self.config, self.B, self.dev, self.world = args, config, B, dev, world
        cfg, ds, skw = configs.named_config(config, mask_ratio=mask_ratio)
        self.cfg, self.ds = cfg, ds
        torch.manual_seed(1234)                       # identical initial weights on every rank
        net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("bench")).to(dev).train()
        net.sync_loss_scalar = False                  # keep the loss on the device: no per-step host sync
        self.mae = hasattr(net.backbone_3d, "prefetch_plan")
        if self.mae:
            net.backbone_3d.dense_spatial_features = False   # the step only consumes decoder rows at the pillar sites
        self.net = net
        self.opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(B), total_steps=total_steps)
        # synthetic frames: frame f of rank r uses seed 100000 * r + 97 * k; pool of distinct batches in pinned host memory
        self.pool = pool
        host = [synth.synth_batch(100000 * rank + 97 * k, B, ds.point_cloud_range, **skw) for k in range(pool)]
        self.pinned = [torch.from_numpy(b).pin_memory() for b in host]
        self.gt = None
        if not self.mae:                              # fine-tune step: synthetic ground-truth boxes (same recipe as the tests)
            rng = np.random.default_rng(5 + rank)
            self.gt = [torch.from_numpy(_synth_boxes(rng, B, 40, np.asarray(ds.point_cloud_range), len(ds.class_names))).to(dev)
                       for _ in range(pool)]
        self.resident = [p.to(dev, non_blocking=True) for p in self.pinned]
        torch.cuda.synchronize()
        # one large cached segment for the caching allocator to carve from: token counts change with every random mask, and a
        # request in a size class the warm-up has not seen would otherwise be a hipMalloc (~1 ms, device-synchronising) inside the
        # timed region (`device_allocs_in_timed_region` in the line counts them); 288 GB of HBM make the reservation free
        if os.environ.get("GDMAE_BENCH_RESERVE_GB", "16") != "0":
            del_me = torch.empty(int(float(os.environ.get("GDMAE_BENCH_RESERVE_GB", "16")) * (1 << 30)), dtype=torch.uint8, device=dev)
            del del_me
        self.mode = {"bf16": args.dtype == "bf16"}
        self.pending = {}
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.stage_dims = [int(b.ENCODER.D_MODEL) for b in cfg.BACKBONE_3D.SST_BLOCK_LIST]
        self.last = None
        self.advance = True           # False: every further step uses the same schedule position (extra legs after the timed region)
        self.host_s = 0.0

    def step(self, i, pts, nxt=None, k=0, nxt_ready=None):
        if getattr(self, "pre_sync", False):          # host-time leg: every step is issued into an idle device
            torch.cuda.synchronize()
        t_host = time.perf_counter()
        try:
            return self._step(i, pts, nxt, k, nxt_ready)
        finally:
            self.host_s += time.perf_counter() - t_host       # host time inside the step call (issue + any wait on the plan event)

    def _step(self, i, pts, nxt=None, k=0, nxt_ready=None):
        net, opt, B = self.net, self.opt, self.B
        opt.zero_grad()
        bd = {"points": pts, "batch_size": B, "_gdmae_grad_sync": opt.sync, "_gdmae_dims": self.stage_dims}
        if self.gt is not None:
            bd["gt_boxes"] = self.gt[k % self.pool]
        pf = None
        if self.mae and self.args.prefetch:
            plan = self.pending.pop(id(pts), None) or net.backbone_3d.prefetch_plan(pts, B).finish()
            bd["_gdmae_vox"], bd["_gdmae_plan"] = plan
            if nxt is not None and self.args.plan_at == "conv":
                # geometry plan of the NEXT batch: issued from inside this step's forward right before the decoder's tile convolution
                # (the matrix-core-bound launch of the step) on a side stream ordered after the work queued so far (incl. the wait
                # for that batch's copy): its atomics-and-scatter kernels run under that launch (7.54 -> 7.44 ms per step same-box
                # against issuing it at the start of the step) and are complete long before the host asks for them
                pf = net.backbone_3d.prefetch_plan_under_decoder(nxt, B, ready=nxt_ready)
            elif nxt is not None and self.args.plan_at == "start":
                pf = net.backbone_3d.prefetch_plan(nxt, B, ready=nxt_ready)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=self.mode["bf16"]):
            ret, tb, _ = net(bd)
        if hasattr(pf, "ensure_issued"):
            pf.ensure_issued()                   # (a forward without the tile convolution - fp32 mode - issues the plan here)
        if self.mae and self.args.prefetch and nxt is not None and self.args.plan_at == "bwd":
            pf = net.backbone_3d.prefetch_plan(nxt, B, ready=nxt_ready)      # experiment: under the first launches of the backward (slower)
        ret["loss"].backward()
        opt.all_reduce_grads()
        opt.step(i)
        if pf is not None:
            # ... and collected here, still behind the queued backward / optimizer kernels: the host-side part of the plan
            # (counts to Python ints, ~300 tensor views) is off the next step's critical path, like a data-loader batch
            # that is ready before it is asked for
            self.pending[id(nxt)] = pf.finish()
        self.last = (ret["loss"], bd)
        return ret["loss"], bd

    # ---- feeds: run n steps
    def _fetch(self, j):
        """H2D copy of pooled batch j on the copy stream (DMA next to the compute of the previous batch)."""
        with torch.cuda.stream(self.copy_stream):
            t = self.pinned[j % self.pool].to(self.dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        return t, ev

    def prime(self):
        """Batch 0 of an h2d feed: copied BEFORE the timed region (inputs of the first step are resident when it starts)."""
        self._primed = self._fetch(0)
        self._primed[1].synchronize()

    def feed_h2d(self, n, i0=0):
        nxt, nev = getattr(self, "_primed", None) or self._fetch(0)
        self._primed = None
        main = torch.cuda.current_stream()
        for i in range(n):
            cur, cev = nxt, nev
            main.wait_event(cev)
            cur.record_stream(main)
            # the next batch's copy: inside the timed region, overlapped with this step (the plan stream waits for its event)
            nxt, nev = self._fetch(i + 1) if i + 1 < n else (None, None)
            self.step(i0 + (i if self.advance else 0), cur, nxt, k=i, nxt_ready=nev)

    def feed_resident(self, n, i0=0):
        r = self.resident
        for i in range(n):
            self.step(i0 + (i if self.advance else 0), r[i % self.pool], r[(i + 1) % self.pool] if i + 1 < n else None, k=i)

    def feed(self, n, i0=0):
        (self.feed_h2d if self.args.feed == "h2d" else self.feed_resident)(n, i0)


def _synth_boxes(rng, B, n_max, pcr, n_class):
    """(B, n_max, 8) ground-truth boxes [x, y, z, dx, dy, dz, heading, class] with trailing zero rows (fine-tune legs)."""
    out = np.zeros((B, n_max, 8), dtype=np.float32)
    size = {1: (4.2, 1.8, 1.6), 2: (0.8, 0.7, 1.7), 3: (1.8, 0.7, 1.6)}
    for b in range(B):
        n = int(rng.integers(n_max // 2, n_max - 1))
        out[b, :n, 0] = rng.uniform(pcr[0] + 1, pcr[3] - 1, n)
        out[b, :n, 1] = rng.uniform(pcr[1] + 1, pcr[4] - 1, n)
        out[b, :n, 2] = rng.uniform(-1.5, 0.5, n)
        cls = rng.integers(1, n_class + 1, n)
        for i in range(n):
            out[b, i, 3:6] = np.array(size[min(int(cls[i]), 3)]) * rng.uniform(0.8, 1.3, 3)
        out[b, :n, 6] = rng.uniform(-np.pi, np.pi, n)
        out[b, :n, 7] = cls
    return out


def pin_rank_to_numa(local, world):
    """One rank per GPU on one host: bind the process to cores of the NUMA node its GPU hangs off (sysfs numa_node of the PCI
    function; the node's cpu list is split evenly between the ranks that land on it), so that eight ranks' interpreters,
    autograd threads and pinned staging buffers do not migrate across sockets.  Best effort: returns a description or None."""
    try:
        if not hasattr(os, "sched_setaffinity"):
            return None
        prop = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (getattr(prop, "pci_domain_id", 0), prop.pci_bus_id, prop.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = []
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus += list(range(int(lo), int(hi or lo) + 1))
        allowed = sorted(set(cpus) & set(os.sched_getaffinity(0)))
        # ranks sharing the node: GPUs whose numa_node is the same, in device order
        same = []
        for dv in range(torch.cuda.device_count()):
            pr = torch.cuda.get_device_properties(dv)
            b2 = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            try:
                if int(open(f"/sys/bus/pci/devices/{b2}/numa_node").read()) == node:
                    same.append(dv)
            except OSError:
                pass
        if world > 1 and local in same and len(allowed) >= 4 * len(same):
            per = len(allowed) // len(same)
            k = same.index(local)
            allowed = allowed[k * per:(k + 1) * per]
        # rehearsal of N ranks on fewer GPUs (tools/eight_ranks_one_gpu.sh: gloo, every rank on cuda:0): the ranks that share a device
        # split its share of the cores as if each had a device of its own there
        share = max(1, world // max(1, torch.cuda.device_count()))
        k = int(os.environ.get("LOCAL_RANK", "0")) // torch.cuda.device_count()
        if os.environ.get("GDMAE_BENCH_PIN_SLICE"):            # "k/n": independent processes sharing a device (tools/eight_procs_one_gpu.sh)
            k, share = (int(v) for v in os.environ["GDMAE_BENCH_PIN_SLICE"].split("/"))
        if share > 1 and len(allowed) >= 2 * share:
            per = len(allowed) // share
            allowed = allowed[k * per:(k + 1) * per]
        if not allowed:
            return None
        os.sched_setaffinity(0, allowed)
        return {"numa_node": node, "cpus": f"{allowed[0]}-{allowed[-1]} ({len(allowed)})", "pci": bdf}
    except Exception:      # noqa: BLE001  (placement is an optimisation, never a reason to fail the run)
        return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    local = local % torch.cuda.device_count()      # several ranks may share a GPU in the 2-rank smoke run (gloo backend)
    torch.cuda.set_device(local)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    if not args.autograd_thread:
        torch.autograd.set_multithreading_enabled(False)
    dev = torch.device("cuda", local)
    affinity = pin_rank_to_numa(local, world) if (world > 1 or os.environ.get("GDMAE_BENCH_PIN", "0") == "1") else None
    # GDMAE_BENCH_FORCE_DIST=1: rehearsal of the N > 1 code path on ONE GPU over RCCL (a one-rank nccl group; the gradient
    # exchange is forced on - sum over one rank = identity): every collective, stream hand-over and barrier of the multi-GPU
    # run executes against the real library.  The printed line is the N = 1 line plus "grad_sync".
    force_dist = world == 1 and os.environ.get("GDMAE_BENCH_FORCE_DIST", "0") == "1"
    distd = world > 1 or force_dist
    if distd:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # nccl == RCCL on ROCm; GDMAE_DIST_BACKEND=gloo only for the control-flow smoke run of two ranks on one GPU
        dist.init_process_group(os.environ.get("GDMAE_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if distd:
        # create the communicator HERE, on the main thread and the current stream: the bucketed gradient exchange issues its
        # first collectives from tensor hooks inside backward() (autograd thread, communication stream)
        warm = torch.ones(1, device=dev)
        dist.all_reduce(warm)
        torch.cuda.synchronize()
        assert int(warm.item()) == world
        # RCCL prints a version banner through C stdio when the communicator is created; on a pipe that buffer would be flushed at process
        # exit, i.e. BEHIND the JSON line (seen with the one-rank RCCL rehearsal: the line was not the last one on stdout).  Flush it now.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:      # noqa: BLE001
            pass

    # BASELINE config B's step on every GPU (8 frames per GPU = the reference's BATCH_SIZE_PER_GPU) at every N: the per-GPU work is FIXED as
    # N grows, which is what "scaling": "weak" in the line says and what the driver's efficiency (value(N) / (N value(1))) assumes.  Until
    # round 6 the default at N > 1 was config C's 4 frames per GPU (global batch 32 at N = 8): against the 8-frame N = 1 point that curve
    # could not exceed 0.78 even with a free gradient exchange (the 4-frame step is 4.6 - 4.9 ms for half the frames of the 7.4 - 7.7 ms
    # 8-frame step).  Config C's share per GPU: --batch-per-gpu 4 (timed on one GPU in every default run: also.config_C_batch_4_per_gpu).
    explicit_batch = args.batch_per_gpu is not None
    if args.batch_per_gpu is None:
        args.batch_per_gpu = 8
    B = args.batch_per_gpu
    named = args.config
    if args.config == "B" and world > 1 and B == 4:
        named = "C"
    # every pooled batch must pass through the untimed warmup once (first-use GEMM algorithm timing, MIOpen find)
    args.pool = max(1, min(args.pool, args.warmup))
    total_steps = args.warmup + args.steps + 1
    wl = Workload(args, args.config, B, dev, rank, world, args.mask_ratio, total_steps, args.pool)
    if force_dist:
        wl.opt.sync.force = True
    n_params = wl.opt.n
    use_bf16 = wl.mode["bf16"]

    def sync_all():
        torch.cuda.synchronize()
        if distd:
            dist.barrier()
        torch.cuda.synchronize()

    wl.feed(args.warmup, 0)
    if args.feed == "h2d":
        wl.prime()                                  # the first timed step's batch is resident when the clock starts
    sync_all()
    from gdmae_hip import plan as gplan
    n_dev_alloc0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
    trace_allocs = os.environ.get("GDMAE_BENCH_TRACE_ALLOCS", "0") == "1"        # diagnosis: who calls hipMalloc inside the timed region
    if trace_allocs:
        torch.cuda.memory._record_memory_history(max_entries=200000, stacks="python")
    wl.host_s = 0.0
    wait0 = gplan.EVENT_WAIT_S
    wl.opt.sync.measure = distd
    t0 = time.perf_counter()
    wl.feed(args.steps, args.warmup)
    sync_all()
    dt = time.perf_counter() - t0
    n_dev_alloc = torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - n_dev_alloc0      # hipMalloc calls inside the timed region
    if trace_allocs:
        snap = torch.cuda.memory._snapshot()
        torch.cuda.memory._record_memory_history(enabled=None)
        for tr in snap.get("device_traces", []):
            for ev in tr:
                if ev.get("action") in ("segment_alloc", "segment_free"):
                    fr = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in ev.get("frames", []) if "torch/" not in f["filename"]][:6]
                    print(f"[alloc trace] {ev['action']} {ev['size'] / 2**20:.1f} MiB stream {ev.get('stream')} <- {' <- '.join(fr)}", file=sys.stderr)
    host_wait_ms = 1e3 * (gplan.EVENT_WAIT_S - wait0) / args.steps
    host_ms = 1e3 * wl.host_s / args.steps - host_wait_ms      # issue time: what the host needs per step when it never has to wait
    wl.opt.sync.measure = False
    exposed_ms = wl.opt.sync.exposed_ms() if distd else None
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if distd:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if distd:                                       # slowest rank's host time / exposed exchange time
        hx = torch.tensor([host_ms, exposed_ms or 0.0, -host_wait_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(hx, op=dist.ReduceOp.MAX)
        host_ms, exposed_ms, host_wait_ms = float(hx[0].item()), float(hx[1].item()), -float(hx[2].item())
    frames = B * world * args.steps
    fps = frames / dt
    wl.advance = False
    # what the host needs to ISSUE a step when nothing makes it wait (device idle at the start of every step, the wait for the plan's
    # event excluded): the split of host_ms_per_step / host_wait_ms_per_step above moves with where a full launch queue happens to block
    # the issuing thread, this number does not
    host_idle_ms = None
    if wl.mae and args.prefetch:
        wl.pre_sync = True
        wl.feed(2, args.warmup)                       # (first steps after the switch: pending plan handed over)
        h0, w0 = wl.host_s, gplan.EVENT_WAIT_S
        wl.feed(8, args.warmup)
        host_idle_ms = 1e3 * ((wl.host_s - h0) - (gplan.EVENT_WAIT_S - w0)) / 8
        wl.pre_sync = False
        sync_all()
        if distd:
            hi = torch.tensor([host_idle_ms], dtype=torch.float64, device=dev)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            host_idle_ms = float(hi.item())
    loss, bd = wl.last
    final_loss = float(loss.detach())
    assert np.isfinite(final_loss), "training diverged"

    pre = args.config in ("A", "B", "E")
    out = {"metric": "MAE pre-train frames/sec (Waymo-shape, 180k pts, 75% mask)" if pre else "fine-tune frames/sec (KITTI-shape, CenterPoint head)",
           "value": round(fps, 2),
           "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(1e3 * dt / args.steps, 3), "host_ms_per_step": round(host_ms, 3), "host_wait_ms_per_step": round(host_wait_ms, 3), "host_issue_ms_idle_device": None if host_idle_ms is None else round(host_idle_ms, 3), "device_allocs_in_timed_region": int(n_dev_alloc), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": WORKLOADS[named] + (f", mask {args.mask_ratio}" if pre else "") +
                                  ", full train step (H2D of the next batch + fwd + bwd + grad all-reduce + clip + Adam)",
                      "frames_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                      "precision": ("16-bit mode: bf16 rows, gradients and products, fp32 accumulation; fp16 operands (11 significand bits, same "
                                    "matrix-core rate) in the forward products of DynVFE's second layer, the decoder's deconvolution rows and its "
                                    "tile convolution; index / loss kernels fp32, prediction head fp32-grade") if use_bf16 else "fp32",
                      "library_gemm": "forbidden: GDMAE_NO_LIBRARY=%s (a product outside the own kernels' shapes raises)" % os.environ.get("GDMAE_NO_LIBRARY"),
                      "device_allocs_in_timed_region": int(n_dev_alloc),      # hipMalloc calls between the two clocks (each synchronises the device)
                      "geometry_plan": ("of batch t+1: one library call on a side stream, issued from inside step t's forward right before the "
                                        "decoder's tile convolution (SPTBackboneMAE.prefetch_plan_under_decoder)" if args.plan_at == "conv" else
                                        "of batch t+1: one library call on a side stream, issued at the %s of step t" % ("start" if args.plan_at == "start" else "start of the backward"))
                                       if (pre and args.prefetch) else None,
                      "autograd": "backward nodes on the calling thread (torch.autograd.set_multithreading_enabled(False))" if not args.autograd_thread
                                  else "torch default (autograd worker thread)",
                      "params": n_params, "mask_ratio": args.mask_ratio if pre else None, "loss_last": round(final_loss, 5),
                      "inputs": ("pinned host batches; batch t+1 copied on a copy stream during step t (inside the timed region), "
                                 "batch 0 resident when the clock starts" if args.feed == "h2d" else "resident in HBM"),
                      "outputs_skipped": ["batch_dict['spatial_features'] dense (B,128,Y,X) map - the pre-training step consumes the "
                                          "decoder only at the pillar sites (SPTBackboneMAE.dense_spatial_features = False)"] if wl.mae else [],
                      "parity_bound": ("16-bit throughput mode (bf16 rows and gradients; the decoder's forward products and DynVFE's second layer multiply fp16 "
                                       "operands): voxel indices / token masks / window partition bit-exact vs the oracle; at 8 full-size "
                                       "frames loss within 1e-4 (measured 5.6e-5 / 3.8e-5 for two weight seeds), per-parameter gradient norm "
                                       "within 6.5 % (2.3 %) and cosine >= 0.988 (0.994) of the fp32 parity mode "
                                       "(tests/test_full_size_properties.py), which itself is held to loss 1e-4 rel of the reference and of "
                                       "the oracle at full size (also.fp32_parity_mode is that mode's throughput, on this library's own "
                                       "fp32 GEMM)") if use_bf16 else
                                      "fp32 parity mode: bit-exact indices / masks, Chamfer loss within 1e-4 rel of the reference"}}

    # the roofline steps run on EVERY rank: they contain the gradient all-reduce, a collective the other ranks must join
    # kernel brackets with the training stream alone on the device: during these extra steps the next batch's plan is issued at the START
    # of the step (as before round 5's last change), not under the decoder's tile convolution - a bracket around that launch would
    # otherwise time the plan's kernels with it (0.47 -> 0.37 of the matrix-core peak in the same run)
    plan_at, args.plan_at = args.plan_at, "start"
    roofline = measure_roofline(wl.step, lambda n: wl.feed(n, total_steps - 1), 4) if (not args.no_roofline and wl.mae) else None
    args.plan_at = plan_at
    loss, bd = wl.last
    vox, ep = bd.get("_gdmae_vox"), bd.get("_gdmae_plan")

    def timed_leg(w, n_warm, n_timed, feed=None):
        """frames/s of `n_timed` more steps after `n_warm` untimed ones (single process; used for the extra legs)."""
        t_leg0 = time.perf_counter()
        w.pending.clear()
        f = feed or w.feed
        f(n_warm, total_steps - 1)
        if f == w.feed_h2d or (feed is None and w.args.feed == "h2d"):
            w.prime()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        f(n_timed, total_steps - 1)
        torch.cuda.synchronize()
        dtl = time.perf_counter() - t1
        return {"value": round(w.B * n_timed / dtl, 2), "unit": "frames/s", "ms_per_step": round(1e3 * dtl / n_timed, 3), "steps": n_timed,
                "frames_per_gpu": w.B, "leg_wall_s": round(time.perf_counter() - t_leg0, 1)}

    also = {}
    if world == 1 and not args.no_also and pre:
        # ---- the same loop with batches that never leave HBM (what round 1 / 2 reported as the headline)
        other_feed = wl.feed_resident if args.feed == "h2d" else wl.feed_h2d
        also["resident_inputs" if args.feed == "h2d" else "h2d_inclusive"] = timed_leg(wl, 2, args.steps, other_feed)
        # ---- the drop-in default: batch_dict['spatial_features'] materialised dense as the reference module does
        #      (spt_backbone_mae.py:130-138; SPTBackboneMAE.dense_spatial_features = True is the class default, the headline turns it
        #      off because the pre-training step never reads the map - INTEGRATION.md section A)
        if wl.mae:
            wl.net.backbone_3d.dense_spatial_features = True
            also["drop_in_default"] = timed_leg(wl, 3, max(5, args.steps // 2))
            also["drop_in_default"]["note"] = "dense_spatial_features = True: the (B,128,Y,X) map of the reference is written every step"
            wl.net.backbone_3d.dense_spatial_features = False
        # ---- the reference yaml's own mask ratio (tools/cfgs/waymo_models/gd_mae_ssl.yaml:158)
        other = 0.85 if abs(args.mask_ratio - 0.85) > 1e-6 else 0.75
        wl.net.backbone_3d.mask_ratio = other
        also[f"mask_{other}"] = timed_leg(wl, 6, max(5, args.steps // 2))      # (new token counts: every pooled batch twice through the warm-up)
        wl.net.backbone_3d.mask_ratio = args.mask_ratio
        # ---- fp32 parity mode (the mode whose loss is held to 1e-4 of the reference by tests/)
        if wl.mode["bf16"]:
            wl.mode["bf16"] = False
            also["fp32_parity_mode"] = timed_leg(wl, 3, max(4, args.steps // 4))
            also["fp32_parity_mode"]["note"] = ("no autocast: fp32 rows, every product on the library's own exact-fp32 MFMA GEMM (csrc/gemm_f32.hip), "
                                                "the decoder's dense 3x3 convolution as six bf16 matrix-core launches on three-piece operand splits with fp32 accumulation "
                                                "(csrc/conv_dense.hip OF32, 2e-6 vs fp64) - no library call in this mode either")
            wl.mode["bf16"] = True
        wl.pending.clear()
        if args.config == "B" and not explicit_batch:
            # ---- config C's per-GPU batch on this one GPU (the N = 1 point of the driver's scaling curve runs 8 frames per GPU;
            #      this is the same step at the 4 frames per GPU that --gpus 2 / 4 / 8 use) and the other single-GPU configs
            extra = [("config_C_batch_4_per_gpu", "B", 4), ("config_E", "E", 8)]
            # ... and the headline's step at twice the reference's per-GPU batch: how much of the step is per-launch cost that more rows per
            # launch amortise (the fused layer launches' weight stream per 32-row tile, ~7 us of fixed cost per launch)
            extra.append(("config_B_batch_16_per_gpu", "B", 16))
            # config D (fine-tune step) is part of the default run since round 5: its dense convolutions are the library's own
            # (csrc/conv_dense.hip), there is no MIOpen search at first use any more
            extra.append(("config_D_finetune", "D", 8))
            for key, cfg_name, b in extra:
                try:
                    t_build = time.perf_counter()
                    w2 = Workload(args, cfg_name, b, dev, rank, world, args.mask_ratio, total_steps, 2)
                    w2.advance = False
                    also[key] = timed_leg(w2, 4, max(8, args.steps // 2))
                    also[key]["leg_wall_s"] = round(time.perf_counter() - t_build, 1)
                    also[key]["workload"] = WORKLOADS["C" if key.startswith("config_C") else cfg_name]
                    if key == "config_B_batch_16_per_gpu":
                        also[key]["note"] = "NOT the reference's batch (BATCH_SIZE_PER_GPU = 8 is the headline): listed for the batch dependence only"
                    del w2
                    torch.cuda.empty_cache()
                except Exception as e:     # noqa: BLE001  (an extra leg must not take the headline line down)
                    also[key] = {"error": repr(e)[:300]}
    if rank == 0:
        if vox is not None and ep is not None and pre:
            # ---- sizes of the last timed batch (for the algorithmic byte model)
            N, M = vox.N / B, vox.M / B
            Ms = [s.n_tok / B for s in ep.stages]
            dsz = [int(b.ENCODER.D_MODEL) for b in wl.cfg.BACKBONE_3D.SST_BLOCK_LIST]
            ds = wl.ds
            G = int(ds.grid_size[0]) * int(ds.grid_size[1])
            a = 2 if use_bf16 else 4
            bytes_train = 3 * algorithmic_bytes_per_frame(N, M, Ms, dsz, G, a)
            out["config"].update({"points_per_frame": int(N), "pillars_per_frame": int(M), "tokens_per_frame": [int(m) for m in Ms]})
            if ep.dec_tiles is not None:
                nt = B * ((int(ds.grid_size[1]) + 7) // 8) * ((int(ds.grid_size[0]) + 7) // 8)
                out["config"]["decoder_active_tiles"] = f"{ep.dec_tiles.n_act} of {nt}"
            dense_model = {"bytes_train_per_frame": int(bytes_train), "achieved_GBs": round(bytes_train * fps / world / 1e9, 1),
                           "frac_of_8TBs": round(bytes_train * fps / world / 1e9 / HBM_PEAK_GBS, 4),
                           "note": "SURVEY 8d whole-step algorithmic bytes with the DENSE-decoder term (2208 x all BEV cells x a): bytes this "
                                   "build does not move - it flatters the step; listed for comparison only"}
            if ep.dec_tiles is not None:
                # first: the byte model of what actually runs (decoder term over the sites of the active 8 x 8 tiles only)
                g_act = 64.0 * ep.dec_tiles.n_act / B
                bt_sp = 3 * algorithmic_bytes_per_frame(N, M, Ms, dsz, g_act, a)
                out["step_bytes_model"] = {
                    "bytes_train_per_frame": int(bt_sp), "achieved_GBs": round(bt_sp * fps / world / 1e9, 1),
                    "frac_of_8TBs": round(bt_sp * fps / world / 1e9 / HBM_PEAK_GBS, 4), "decoder_sites_per_frame": int(g_act),
                    "note": "SURVEY 8d whole-step algorithmic bytes x frames/s per GPU, decoder term over the sites of the active tiles (what "
                            "this build moves)",
                    "dense_decoder_formula": dense_model}
            else:
                out["step_bytes_model"] = dense_model
        if affinity is not None and not distd:
            out["pin"] = affinity
        if distd:
            opt = wl.opt
            out["grad_sync"] = {"buckets": [[b, hi - lo] for b, lo, hi in opt.buckets], "last_step": opt.sync.log,
                                "exposed_ms": None if exposed_ms is None else round(exposed_ms, 4),
                                "exposed_ms_note": "time per step the compute stream waited for the last collective after its own backward work "
                                                   "(HIP events on both streams, max over ranks); 0 = the exchange hid under the backward",
                                "checked_steps": opt.sync.checked_steps,
                                "cpu_affinity": affinity,
                                "note": "one all-reduce per bucket; 'overlapped' = launched on the communication stream from inside "
                                        "backward(), 'tail' = after it"}
        if roofline is not None:
            out["roofline"] = roofline
        shares = os.path.join(REPO, "profiles", "class_shares.json")
        if os.path.exists(shares):
            # NOT measured in this run: the committed one-step rocprofv3 trace of the builder's profiling box
            out["kernel_time_shares_profiling_box"] = json.load(open(shares))
        if also:
            out["also"] = also
            if isinstance(also.get("drop_in_default"), dict) and "value" in also["drop_in_default"]:
                # what a tools/train.py user gets without touching a switch (the reference module's dense spatial_features map written)
                out["config"]["drop_in_default_frames_per_s"] = also["drop_in_default"]["value"]
        if not args.no_cpu_baseline and world == 1 and pre:      # reported at N = 1 only (rank 0's host cores)
            out["cpu_baseline"] = measure_cpu_baseline(args.config, args.mask_ratio)
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)         # (anything a library left in the C buffer goes out before the line, not after it)
        except Exception:      # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    if distd:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
