"""Micro-benchmark of gdmae_tok_gemm (all epilogues) against torch's bf16 GEMM (hipBLASLt) on the encoder's shapes.
Usage (GPU box): python tools/bench_tok_gemm.py [rows]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "gd-mae_amd")):
    sys.path.insert(0, p)
import torch
from gdmae_hip import lib as L
if os.environ.get("GDMAE_LIB"):          # kernel experiments: a variant library built next to the product one
    L.LIB_PATH = os.path.abspath(os.environ["GDMAE_LIB"])

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 40960


def pack(w):
    M, K = w.shape
    dst = torch.empty(M * K, dtype=torch.bfloat16, device=dev)
    jobs = torch.tensor([w.data_ptr(), dst.data_ptr(), M, K, K, 0], dtype=torch.int64).to(dev)
    L.call("gdmae_tok_gemm_pack", L.ptr(jobs), 1, L.stream())
    return dst


def timeit(fn, it=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / it * 1e3


for n in ((rows,) if os.environ.get("TG_ONE") else (rows, rows * 4 // 5, rows // 2)):
    n_pad = (n + 2047) // 2048 * 2048
    for K, N in ((128, 128), (128, 256), (256, 128), (256, 256), (256, 512), (512, 256)):
        X = torch.randn(n_pad, K, device=dev).bfloat16()
        W = torch.randn(N, K, device=dev) / K ** 0.5
        Wb = W.bfloat16()
        Wp = pack(W)
        bias = torch.randn(N, device=dev).bfloat16()
        out0 = torch.empty(n_pad, N, dtype=torch.bfloat16, device=dev)
        out1 = torch.empty_like(out0)
        aux = torch.randn(n_pad, N, device=dev).bfloat16()
        res = torch.randn(n_pad, N, device=dev)
        y = torch.empty(n_pad, N, device=dev)
        st = torch.empty(n_pad, 2, device=dev)
        g, b = torch.ones(N, device=dev), torch.zeros(N, device=dev)
        ybf, ypos = torch.empty_like(out0), torch.empty_like(out0)
        pos = torch.randn(64, N, device=dev)
        tp = torch.randint(0, 64, (n_pad,), device=dev).int()

        def call(epi):
            L.call("gdmae_tok_gemm", L.ptr(X), L.ptr(Wp), L.ptr(bias), n, n_pad, K, N, epi, L.ptr(out0), L.ptr(out1), L.ptr(aux), L.ptr(res),
                   L.ptr(g), L.ptr(b), 1e-5, L.ptr(y), L.ptr(st), L.ptr(ybf), L.ptr(pos), L.ptr(tp), L.ptr(ypos), L.stream())
        la = torch.randn(n_pad, N, device=dev)
        part = torch.empty(n_pad // 32, 3, N, device=dev)

        def call_lnb():
            L.call("gdmae_tok_gemm_ln_bwd", L.ptr(X), L.ptr(Wp), n, n_pad, K, N, L.ptr(res), L.ptr(aux), L.ptr(la), L.ptr(out1), L.ptr(st),
                   L.ptr(g), L.ptr(y), L.ptr(ybf), L.ptr(part), L.stream())
        t_ref = timeit(lambda: torch.addmm(bias, X, Wb.t())) if os.environ.get("TG_REF", "1") == "1" else 0.0
        ts = [timeit(lambda e=e: call(e)) for e in ((0, 1, 2, 3) if N <= 256 else (0, 1, 2))]
        if N <= 256:
            ts.append(timeit(call_lnb))
        fl = 2.0 * n_pad * K * N
        by = [n_pad * (K + N) * 2, n_pad * (K + 2 * N) * 2, n_pad * (K + 2 * N) * 2, n_pad * (K * 2 + N * 2 + N * 4 * 2 + N * 4),
              n_pad * (K * 2 + N * (4 + 2 + 4 + 2 + 4 + 2) + 8)]
        print(f"rows {n_pad:6d} K {K:3d} N {N:3d}  hipBLASLt {t_ref:6.1f} us | " +
              "  ".join(f"epi{e} {t:6.1f} us {by[e] / t / 1e6:5.2f} TB/s" for e, t in enumerate(ts)))
    if True:
        for d in (128, 256):
            Xp = torch.randn(n_pad, d, device=dev).bfloat16()
            Xv = torch.randn(n_pad, d, device=dev).bfloat16()
            Wqk, Wv = pack(torch.randn(2 * d, d, device=dev) / d ** 0.5), pack(torch.randn(d, d, device=dev) / d ** 0.5)
            b3 = torch.randn(3 * d, device=dev).bfloat16()
            qk = torch.empty(n_pad, 2 * d, dtype=torch.bfloat16, device=dev)
            v = torch.empty(n_pad, d, dtype=torch.bfloat16, device=dev)
            tq = timeit(lambda: L.call("gdmae_tok_gemm_qkv", L.ptr(Xp), L.ptr(Xv), L.ptr(Wqk), L.ptr(Wv), L.ptr(b3), n_pad, d, L.ptr(qk), L.ptr(v),
                                       L.stream()))
            print(f"rows {n_pad:6d} qkv d {d:3d}  {tq:6.1f} us {n_pad * d * 2 * 5 / tq / 1e6:5.2f} TB/s")
            # feed-forward block in one launch (linear1 + GELU + linear2 + residual + LayerNorm)
            ff = 2 * d
            W1p, W2p = pack(torch.randn(ff, d, device=dev) / d ** 0.5), pack(torch.randn(d, ff, device=dev) / ff ** 0.5)
            b1, b2 = torch.randn(ff, device=dev).bfloat16(), torch.randn(d, device=dev).bfloat16()
            hb = torch.empty(n_pad, ff, dtype=torch.bfloat16, device=dev)
            resd, yd, std = torch.randn(n_pad, d, device=dev), torch.empty(n_pad, d, device=dev), torch.empty(n_pad, 2, device=dev)
            gd, bd = torch.ones(d, device=dev), torch.zeros(d, device=dev)
            ybd, ypd, fd = (torch.empty(n_pad, d, dtype=torch.bfloat16, device=dev) for _ in range(3))
            posd = torch.randn(64, d, device=dev)
            tpd = torch.randint(0, 64, (n_pad,), device=dev).int()
            tf = timeit(lambda: L.call("gdmae_tok_gemm_ffn", L.ptr(Xp), L.ptr(W1p), L.ptr(b1), L.ptr(W2p), L.ptr(b2), n, n_pad, d, L.ptr(hb),
                                       L.ptr(resd), L.ptr(gd), L.ptr(bd), 1e-5, L.ptr(yd), L.ptr(std), L.ptr(ybd), L.ptr(posd), L.ptr(tpd),
                                       L.ptr(ypd), L.ptr(fd), L.stream()))
            print(f"rows {n_pad:6d} ffn d {d:3d}  {tf:6.1f} us {n_pad * (d * 2 + ff * 2 + d * (4 + 4 + 6)) / tf / 1e6:5.2f} TB/s")
