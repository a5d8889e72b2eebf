"""Weight-gradient products of one encoder layer at the bench sizes: hand-written TN kernel (one matrix per call) vs the library split-K path."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from gdmae_hip import lib as L, ops
dev = torch.device("cuda:0")

def timeit(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for rows, d in ((32768, 128), (49152, 256), (22528, 256)):
    ff = 2 * d
    tot = [0.0, 0.0]
    for M, N in ((d, ff), (ff, d), (d, d), (2 * d, d), (d, d)):
        G = torch.randn(rows, M, device=dev).to(torch.bfloat16)
        X = torch.randn(rows, N, device=dev).to(torch.bfloat16)
        dW = torch.empty(M, N, dtype=torch.float32, device=dev)
        db = torch.empty(M, dtype=torch.float32, device=dev)
        ws = torch.empty(L.load().gdmae_dw_gemm_workspace_bytes(rows, M, N), dtype=torch.uint8, device=dev)
        t_own = timeit(lambda: L.call("gdmae_dw_gemm", L.ptr(G), L.ptr(X), rows, M, N, L.ptr(dW), L.ptr(db), L.ptr(ws), L.stream()))
        t_lib = timeit(lambda: ops.splitk_tn(G, X))
        fl = 2.0 * rows * M * N
        print(f"rows {rows} M {M} N {N}: own {t_own:7.1f} us ({fl / t_own / 1e6:6.1f} TF/s)   library split-K {t_lib:7.1f} us ({fl / t_lib / 1e6:6.1f} TF/s)")
        tot[0] += t_own; tot[1] += t_lib
    print(f"  layer total: own {tot[0]:.1f} us, library {tot[1]:.1f} us (separate launches; the executor groups the five)")
