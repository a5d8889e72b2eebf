"""Would the occupancy levels of one attention call overlap if they were one launch?  Chains of the T = 16 launch and of the
T = 64 launch of a stage-2 layer (bench windows) on one stream vs on two streams, forward and backward."""
import logging, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from gdmae_hip import configs, synth
from gdmae_hip import lib as L
from pcdet.models import build_network
dev = torch.device("cuda:0")
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
net = build_network(cfg, 3, ds, logging.getLogger("p")).to(dev).train()
pts = torch.from_numpy(synth.synth_batch(5, 8, ds.point_cloud_range, **skw)).to(dev)
vox, plan = net.backbone_3d.prefetch_plan(pts, 8).finish()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
REP = 30
for si in (1, 2):
    st = plan.stages[si]
    d = cfg.BACKBONE_3D.SST_BLOCK_LIST[si].ENCODER.D_MODEL
    H = cfg.BACKBONE_3D.SST_BLOCK_LIST[si].ENCODER.NHEAD
    w = st.windows[0]
    qk = torch.randn(st.n_tok, 2 * d, device=dev).to(torch.bfloat16)
    v = torch.randn(st.n_tok, d, device=dev).to(torch.bfloat16)
    g = torch.randn(st.n_tok, d, device=dev).to(torch.bfloat16)
    out = torch.empty_like(v); dqk = torch.empty_like(qk); dv = torch.empty_like(v)
    tau = torch.full((1,), 0.1, device=dev)
    part = torch.zeros(sum(w.n_win) * H + 1, device=dev)
    lv = {w.max_tokens[l]: (sum(w.n_win[:l]), w.n_win[l]) for l in range(len(w.n_win))}
    print("stage", si, "levels", {T: nw for T, (_, nw) in lv.items()})

    def call(T, stream, bwd):
        base, nw = lv[T]
        if bwd:
            L.call("gdmae_window_attention_bwd", L.ptr(qk), L.ptr(v), L.ptr(g), L.ptr(dqk), L.ptr(dv), 1, L.ptr(part), L.ptr(w.csr_tok),
                   L.ptr(w.win_start[base:]), L.ptr(w.win_len[base:]), nw, T, d, H, L.ptr(tau), 0.01, stream.cuda_stream)
        else:
            L.call("gdmae_window_attention_fwd", L.ptr(qk), L.ptr(v), L.ptr(out), 1, L.ptr(w.csr_tok), L.ptr(w.win_start[base:]),
                   L.ptr(w.win_len[base:]), nw, T, d, H, L.ptr(tau), 0.01, stream.cuda_stream)

    def wall(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream()
        e0.record(cur)
        sA.wait_event(e0); sB.wait_event(e0)
        fn()
        cur.wait_stream(sA); cur.wait_stream(sB)
        e1.record(cur)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / REP

    Ts = sorted(lv)
    Tbig = Ts[-1]
    for bwd in (False, True):
        for rnd in range(2):
            ta = wall(lambda: [call(16, sA, bwd) for _ in range(REP)])
            tb = wall(lambda: [call(Tbig, sB, bwd) for _ in range(REP)])
            ts = wall(lambda: [(call(16, sA, bwd), call(Tbig, sA, bwd)) for _ in range(REP)])
            tc = wall(lambda: ([call(16, sA, bwd) for _ in range(REP)], [call(Tbig, sB, bwd) for _ in range(REP)]))
        print(f"  {'bwd' if bwd else 'fwd'}: T16 alone {ta:.1f} us, T{Tbig} alone {tb:.1f} us, alternating on one stream {ts:.1f} us, two streams {tc:.1f} us")
