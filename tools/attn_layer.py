"""All occupancy levels of one attention call (the entry the layer executor uses) on the bench workload's windows, per stage and
shift: product path (impl 0: merged workgroup-cooperative launches) against the round-4 per-(window, head) kernels (impl 3)."""
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
frames = int(os.environ.get("FRAMES", "8"))
pts = torch.from_numpy(synth.synth_batch(5, frames, ds.point_cloud_range, **skw)).to(dev)
vox, plan = net.backbone_3d.prefetch_plan(pts, frames).finish()
tot = {0: [0.0, 0.0], 3: [0.0, 0.0]}
IDENT = bool(os.environ.get("IDENT"))       # experiment: tokens already in window-major order (csr_tok = identity)
COLD = bool(os.environ.get("COLD"))         # every timed call behind a 1 GB fill (L2 / MALL hold none of its operands)
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev) if COLD else None
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
        nl = len(w.n_win)
        lse = torch.empty(st.n_tok, H, device=dev)
        nw_h, T_h = L.host_i32(w.n_win), L.host_i32(w.max_tokens)
        if os.environ.get("LEVELS"):         # experiment: only the occupancy levels whose bit is set (1 = T16, 2 = T32, 4 = T64) have windows
            nw_h = L.host_i32([n if (int(os.environ["LEVELS"]) >> i) & 1 else 0 for i, n in enumerate(w.n_win)])
        ws_t, wl_t = w.win_start, w.win_len
        MT = int(os.environ.get("MERGE16", "0"))          # 1 / 32: onto 32-row tiles (16-row buckets), 64: onto 64-row tiles (48-row buckets)
        MT = 32 if MT == 1 else MT
        if MT in (32, 64) and w.max_tokens[0] == 16 and MT in w.max_tokens and w.n_win[0] > 0:
            # experiment (timing only, the values attend across windows): the T = 16 level's windows merged into pseudo-windows of <= 32
            # rows (all windows whose first row lies in the same 16-row bucket of the level's CSR range) and handed to the T = 32
            # cooperative path - what the sparse level would cost on 32-row tiles with a block-diagonal mask
            n0 = w.n_win[0]
            i32 = w.max_tokens.index(MT)
            s0, l0 = w.win_start[:n0].long(), w.win_len[:n0].long()
            b = (s0 - s0[0]) // (MT - 16)
            first = torch.ones_like(b, dtype=torch.bool); first[1:] = b[1:] != b[:-1]
            last = torch.ones_like(b, dtype=torch.bool); last[:-1] = b[1:] != b[:-1]
            ms, me = s0[first], (s0 + l0)[last]
            assert int((me - ms).max()) <= MT and int((me - ms).sum()) == int(l0.sum())
            offs = [0]
            for n in w.n_win: offs.append(offs[-1] + n)
            parts_s, parts_l, nw_new = [], [], []
            for li in range(len(w.n_win)):
                if li == 0:
                    nw_new.append(0)
                    continue
                ps, pl = [w.win_start[offs[li]:offs[li + 1]]], [w.win_len[offs[li]:offs[li + 1]]]
                if li == i32:
                    ps.insert(0, ms.int()); pl.insert(0, (me - ms).int())
                parts_s += ps; parts_l += pl
                nw_new.append(sum(int(x.numel()) for x in ps))
            ws_t, wl_t = torch.cat(parts_s).contiguous(), torch.cat(parts_l).contiguous()
            nw_h = L.host_i32(nw_new)
            part = torch.zeros(sum(nw_new) * H + 1, device=dev)
            print(f"   merged {n0} T=16 windows ({int(l0.sum())} tokens) into {ms.numel()} pseudo-windows of <= {MT} rows (mean {float((me - ms).float().mean()):.1f})")
        csr = torch.arange(st.n_tok, dtype=torch.int32, device=dev) if IDENT else w.csr_tok
        if os.environ.get("NOCSR"):          # experiment: no index load at all (rows in window-major order, csr_tok = null)
            csr = None
        def fwd():
            L.call("gdmae_window_attention_levels_fwd", L.ptr(qk), L.ptr(v), L.ptr(out), 1, L.ptr(csr), L.ptr(ws_t), L.ptr(wl_t), nl,
                   nw_h, T_h, d, H, L.ptr(tau), 0.01, L.ptr(lse), L.stream())
        def bwd():
            L.call("gdmae_window_attention_levels_bwd", L.ptr(qk), L.ptr(v), L.ptr(g), L.ptr(dqk), L.ptr(dv), 1, L.ptr(part), L.ptr(csr),
                   L.ptr(ws_t), L.ptr(wl_t), nl, nw_h, T_h, d, H, L.ptr(tau), 0.01, L.ptr(out), L.ptr(lse), L.stream())
        line = f"stage {si} shift {shift} windows {w.n_win} tokens {w.n_tok}:"
        for impl in ((0,) if os.environ.get("NOCSR") else (3, 0)):
            L.call("gdmae_set_attention_impl", impl)
            res = []
            for f in (fwd, bwd):
                for _ in range(3): f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if COLD:
                    t = 0.0
                    for _ in range(8):
                        flush.fill_(1); e0.record(); f(); e1.record(); torch.cuda.synchronize()
                        t += e0.elapsed_time(e1)
                    res.append(t / 8 * 1e3)
                    continue
                torch.cuda.synchronize(); e0.record()
                for _ in range(20): f()
                e1.record(); torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / 20 * 1e3)
            tot[impl][0] += res[0]; tot[impl][1] += res[1]
            line += f"   impl {impl}: fwd {res[0]:6.1f} bwd {res[1]:6.1f} us"
        print(line)
L.call("gdmae_set_attention_impl", 0)
print("per step (2 layers per shift and stage):", {k: [round(2 * x, 1) for x in v] for k, v in tot.items()})
