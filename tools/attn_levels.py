"""Window statistics of the bench workload: per stage / shift / occupancy level the window and token counts, and a stand-alone
timing of the attention entry points on exactly those windows (bf16 rows)."""
import logging, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from gdmae_hip import configs, synth, ops
from gdmae_hip import lib as L
from pcdet.models import build_network
dev = torch.device("cuda:0")
cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
net = build_network(cfg, 3, ds, logging.getLogger("p")).to(dev).train()
pts = torch.from_numpy(synth.synth_batch(5, 8, ds.point_cloud_range, **skw)).to(dev)
vox, plan = net.backbone_3d.prefetch_plan(pts, 8).finish()
for si, st in enumerate(plan.stages):
    d = cfg.BACKBONE_3D.SST_BLOCK_LIST[si].ENCODER.D_MODEL
    H = cfg.BACKBONE_3D.SST_BLOCK_LIST[si].ENCODER.NHEAD
    for shift, w in enumerate(st.windows):
        qk = torch.randn(st.n_tok, 2 * d, device=dev).to(torch.bfloat16)
        v = torch.randn(st.n_tok, d, device=dev).to(torch.bfloat16)
        g = torch.randn(st.n_tok, d, device=dev).to(torch.bfloat16)
        out = torch.empty_like(v); dqk = torch.empty_like(qk); dv = torch.empty_like(v)
        tau = torch.full((1,), 0.1, device=dev)
        part = torch.zeros(sum(w.n_win) * H + 1, device=dev)
        base = 0
        for lvl, nw in enumerate(w.n_win):
            T = w.max_tokens[lvl]
            ntok = w.n_tok[lvl]
            def fwd():
                L.call("gdmae_window_attention_fwd", L.ptr(qk), L.ptr(v), L.ptr(out), 1, L.ptr(w.csr_tok), L.ptr(w.win_start[base:]),
                       L.ptr(w.win_len[base:]), nw, T, d, H, L.ptr(tau), 0.01, L.stream())
            def bwd():
                L.call("gdmae_window_attention_bwd", L.ptr(qk), L.ptr(v), L.ptr(g), L.ptr(dqk), L.ptr(dv), 1, L.ptr(part), L.ptr(w.csr_tok),
                       L.ptr(w.win_start[base:]), L.ptr(w.win_len[base:]), nw, T, d, H, L.ptr(tau), 0.01, L.stream())
            res = []
            for f in (fwd, bwd):
                for _ in range(3): f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(20): f()
                e1.record(); torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / 20 * 1e3)
            fb, bb = ntok * 4 * d * 2, ntok * 7 * d * 2
            print(f"stage {si} d={d} H={H} shift {shift} T={T:2d}: windows {nw:6d} tokens {ntok:7d} ({ntok / max(nw, 1):5.1f}/win)  "
                  f"fwd {res[0]:6.1f} us {fb / res[0] / 1e6:5.2f} TB/s   bwd {res[1]:6.1f} us {bb / res[1] / 1e6:5.2f} TB/s")
            base += nw
