"""GPU time per phase of the bench step from HIP events on the training stream (untraced run): forward, backward, optimizer,
and the gaps between consecutive steps (stream idle while the host prepares the next step)."""
import logging, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from gdmae_hip import configs, optim, synth
from pcdet.models import build_network
dev = torch.device("cuda:0")
NB = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 8     # frames per step (4 = config C's share per GPU)
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
torch.manual_seed(1234)
net = build_network(cfg, 3, ds, logging.getLogger("p")).to(dev).train()
net.sync_loss_scalar = False
net.backbone_3d.dense_spatial_features = False
opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(NB), total_steps=200)
batches = [torch.from_numpy(synth.synth_batch(5 + i, NB, ds.point_cloud_range, **skw)).to(dev) for i in range(4)]
resident = torch.cuda.Event(); resident.record()
pend = {}
marks = []
if "--plan-on-main" in sys.argv:           # geometry plan kernels on the training stream itself (serial) instead of the side stream
    from gdmae_hip import plan as _plan
    _plan.PlanPrefetch._side[dev.index] = torch.cuda.current_stream()
SPIN = float(sys.argv[sys.argv.index("--spin-ms") + 1]) if "--spin-ms" in sys.argv else 0.0
PFDROP = 1 if "--prefetch-drop" in sys.argv else (2 if "--prefetch-finish-drop" in sys.argv else 0)
KEEP = []
VOXONLY = "--vox-only" in sys.argv
ENCONLY = "--enc-only" in sys.argv
if ENCONLY:
    from gdmae_hip import plan as _p0
    VRAW = [_p0._voxelize_launch(b, net.backbone_3d.point_cloud_range, net.backbone_3d.voxel_size, net.backbone_3d.grid_size, NB) for b in batches]
    torch.cuda.synchronize()
DUMMY = int(sys.argv[sys.argv.index("--dummy") + 1]) if "--dummy" in sys.argv else 0
SIDE = torch.cuda.Stream()
TINY = torch.zeros(64, device=dev)
REUSE = "--reuse-plans" in sys.argv      # geometry plans of the 4 pooled batches built once: the step without its plan kernels
PLANS = [net.backbone_3d.prefetch_plan(b, NB).finish() for b in batches] if REUSE else None
PREBUILT = "--prebuilt" in sys.argv       # one never-consumed plan per step, all built before the loop (no plan kernels inside the steps)
if PREBUILT:
    REUSE = True
    PLANS = [net.backbone_3d.prefetch_plan(batches[i % 4], NB).finish() for i in range(64)]
    torch.cuda.synchronize()
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
HOST = {"finish": 0.0, "prefetch": 0.0, "forward": 0.0, "backward": 0.0, "opt": 0.0, "n": 0}
def step(i, rec):
    pts, nxt = batches[i % 4], batches[(i + 1) % 4]
    h0 = time.perf_counter()
    e0 = ev()
    opt.zero_grad()
    if REUSE:
        pf = PLANS[i % len(PLANS)]
    else:
        pf = pend.pop(i, None) or net.backbone_3d.prefetch_plan(pts, NB)
    bd = {"points": pts, "batch_size": NB, "_gdmae_grad_sync": opt.sync}
    bd["_gdmae_vox"], bd["_gdmae_plan"] = pf.finish() if hasattr(pf, "finish") else pf
    hp0 = time.perf_counter()
    if REUSE:
        pfn = None
    elif "--plan-at-start" in sys.argv:          # the plan of the next batch issued here, at the start of the step (before round 5's last change)
        pfn = net.backbone_3d.prefetch_plan(nxt, 8, ready=resident)
    else:                                        # ... or from inside the forward, right before the decoder's tile convolution (product)
        pfn = net.backbone_3d.prefetch_plan_under_decoder(nxt, 8, ready=resident)
    h1 = time.perf_counter()
    HOST["issue"] = HOST.get("issue", 0.0) + (h1 - hp0 if rec else 0.0)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ret, _, _ = net(bd)
    e1 = ev()
    h2 = time.perf_counter()
    if PFDROP:                                   # the complete prefetch (launches, count copy, event) without finish(); result dropped
        KEEP.append(net.backbone_3d.prefetch_plan(nxt, 8, ready=resident))
        if len(KEEP) > 2:
            old = KEEP.pop(0)
            if PFDROP == 2:
                old.finish()                     # ... or finished two steps later (host-side finalize + stream ordering), then dropped
    if VOXONLY:                                  # only the voxelization kernels of the next batch on the side stream (result dropped)
        from gdmae_hip import plan as _p
        with torch.cuda.stream(SIDE):
            _p._voxelize_launch(nxt, net.backbone_3d.point_cloud_range, net.backbone_3d.voxel_size, net.backbone_3d.grid_size, NB)
    if ENCONLY:                                  # only the encoder-plan kernels (masking, token sets, rulebooks, windows, tiles)
        from gdmae_hip import plan as _p
        from pcdet.models.backbones_3d.spt_backbone import stage_plan_args
        bb = net.backbone_3d
        with torch.cuda.stream(SIDE):
            vr = VRAW[(i + 1) % 4]
            gx, gy, gz = vr["grid"]
            _p._encoder_launch(vr, max(1, min(vr["n0"], 8 * gx * gy * gz)), *stage_plan_args(bb.model_cfg.SST_BLOCK_LIST),
                               1 - bb.mask_ratio, None, bb._dec_sources())
    if SPIN:                                     # host busy-wait: emulates host-side work without any GPU work
        t_end = time.perf_counter() + SPIN * 1e-3
        while time.perf_counter() < t_end:
            pass
    if DUMMY:                                    # N empty launches on a side stream: what do kernel boundaries on another queue cost?
        with torch.cuda.stream(SIDE):
            for _ in range(DUMMY):
                TINY.add_(1.0)
    h3 = time.perf_counter()
    ret["loss"].backward()
    e2 = ev()
    h4 = time.perf_counter()
    opt.all_reduce_grads()
    opt.step(i)
    e3 = ev()
    h5 = time.perf_counter()
    if not REUSE:
        hw0 = time.perf_counter()
        if hasattr(pfn, "event"):
            pfn.event.synchronize()
        hw1 = time.perf_counter()
        pend[i + 1] = pfn.finish()
        if rec:
            HOST["wait"] = HOST.get("wait", 0.0) + hw1 - hw0
    h6 = time.perf_counter()
    if rec:
        marks.append((e0, e1, e2, e3))
        HOST["finish"] += (h1 - h0) + (h6 - h5); HOST["forward"] += h2 - h1; HOST["prefetch"] += h3 - h2
        HOST["backward"] += h4 - h3; HOST["opt"] += h5 - h4; HOST["n"] += 1
print("stream priority range (least, greatest):", torch.cuda.Stream.priority_range())
if "--high-priority" in sys.argv:          # training on a high-priority stream: the side-stream plan only fills what it leaves idle
    hp = torch.cuda.Stream(priority=torch.cuda.Stream.priority_range()[1])
    hp.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(hp)
def gemm_stats():
    import ctypes as C
    from gdmae_hip import lib as L
    a = (C.c_longlong * 2)()
    L.call("gdmae_gemm_stats", a)
    return int(a[0]), int(a[1])
for i in range(10): step(i, False)
torch.cuda.synchronize()
print("library GEMM calls / plans created after the warm-up:", gemm_stats())
ALLOC0 = torch.cuda.memory_stats(dev).get("num_device_alloc", 0)
t0 = time.perf_counter()
N = 40
for i in range(10, 10 + N): step(i, True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
print("library GEMM calls / plans created after the timed steps:", gemm_stats(), "| hipMalloc calls inside the timed steps:",
      torch.cuda.memory_stats(dev).get("num_device_alloc", 0) - ALLOC0)
f = sum(a.elapsed_time(b) for a, b, _, _ in marks) / N
b = sum(b_.elapsed_time(c) for _, b_, c, _ in marks) / N
o = sum(c.elapsed_time(d) for _, _, c, d in marks) / N
gap = sum(marks[k][3].elapsed_time(marks[k + 1][0]) for k in range(N - 1)) / (N - 1)
print("host ms/step: " + ", ".join(f"{k} {1e3 * v / HOST['n']:.2f}" for k, v in HOST.items() if k != "n") + f" | total {1e3 * sum(v for k, v in HOST.items() if k not in ('n', 'issue', 'wait')) / HOST['n']:.2f} (issue / wait are parts of finish)")
print(f"wall/step {wall:.3f} ms | forward span {f:.3f} backward span {b:.3f} optimizer span {o:.3f} step-to-step gap {gap:.3f} | sum {f + b + o + gap:.3f}")
if os.environ.get("V3_TRACE_DUMP"):       # variant library built with -DV3_TRACE=<workgroup> (csrc/layer_v3.hip): stamps of the last launch
    import ctypes as _C
    from gdmae_hip import lib as _L
    _buf = (_C.c_ulonglong * 512)()
    assert _C.CDLL(_L.LIB_PATH).gdmae_debug_v3_trace(_buf) == 0
    for w in range(4):
        t = [_buf[w * 64 + i] for i in range(64)]
        print("v3 wave", w, " ".join("%d:%.2f" % (i, (t[i] - t[0]) / 100.0) for i in range(64) if t[i]))
if os.environ.get("TL_TRACE_DUMP"):       # variant library built with -DTL_TRACE=<workgroup> (csrc/layer_fused.hip k_layer_fwd): stamps of the last launch
    import ctypes as _C
    from gdmae_hip import lib as _L
    _buf = (_C.c_ulonglong * 256)()
    assert _C.CDLL(_L.LIB_PATH).gdmae_debug_tl_trace(_buf) == 0
    for w in range(8):
        t = [_buf[w * 32 + i] for i in range(32)]
        print("tl wave", w, " ".join("%d:%.2f" % (i, (t[i] - t[0]) / 100.0) for i in range(32) if t[i]))
