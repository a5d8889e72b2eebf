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


for n in (rows, rows * 4 // 5, rows // 2):
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
        t_ref = timeit(lambda: torch.addmm(bias, X, Wb.t()))
        ts = [timeit(lambda e=e: call(e)) for e in ((0, 1, 2, 3) if N <= 256 else (0, 1, 2))]
        fl = 2.0 * n_pad * K * N
        by = [n_pad * (K + N) * 2, n_pad * (K + 2 * N) * 2, n_pad * (K + 2 * N) * 2, n_pad * (K * 2 + N * 2 + N * 4 * 2 + N * 4)]
        print(f"rows {n_pad:6d} K {K:3d} N {N:3d}  hipBLASLt {t_ref:6.1f} us | " +
              "  ".join(f"epi{e} {t:6.1f} us {by[e] / t / 1e6:5.2f} TB/s {fl / t / 1e6:5.0f} TF" for e, t in enumerate(ts)))
