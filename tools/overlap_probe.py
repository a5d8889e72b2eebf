"""Do two latency-bound kernels of the training step overlap when they run on two streams?  A chain of token GEMMs
(LayerNorm epilogue, 40 960 x 256 -> 256) on stream A, a chain of weight-gradient products (gdmae_dw_gemm) on stream B:
time of A alone, B alone, A then B on one stream, A and B concurrently.  Usage (GPU box): python tools/overlap_probe.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from gdmae_hip import lib as L

dev = torch.device("cuda:0")
n = n_pad = 40960
K = N = 256
X = torch.randn(n_pad, K, device=dev).bfloat16()
W = torch.randn(N, K, device=dev) / K ** 0.5
Wp = torch.empty(N * K, dtype=torch.bfloat16, device=dev)
jobs = torch.tensor([W.data_ptr(), Wp.data_ptr(), N, K, K, 0], dtype=torch.int64).to(dev)
L.call("gdmae_tok_gemm_pack", L.ptr(jobs), 1, L.stream())
bias = torch.randn(N, device=dev).bfloat16()
out0 = torch.empty(n_pad, N, dtype=torch.bfloat16, device=dev)
res = torch.randn(n_pad, N, device=dev)
y = torch.empty(n_pad, N, device=dev)
st = torch.empty(n_pad, 2, device=dev)
g, b = torch.ones(N, device=dev), torch.zeros(N, device=dev)
ybf, ypos = torch.empty_like(out0), torch.empty_like(out0)
pos = torch.randn(64, N, device=dev)
tp = torch.randint(0, 64, (n_pad,), device=dev).int()

G = torch.randn(n_pad, 512, device=dev).bfloat16()
X2 = torch.randn(n_pad, 256, device=dev).bfloat16()
dW = torch.empty(512, 256, dtype=torch.float32, device=dev)
db = torch.empty(512, dtype=torch.float32, device=dev)
ws = torch.empty(L.load().gdmae_dw_gemm_workspace_bytes(n_pad, 512, 256), dtype=torch.uint8, device=dev)

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
REP = 40


def chain_a(stream):
    for _ in range(REP):
        L.call("gdmae_tok_gemm", L.ptr(X), L.ptr(Wp), L.ptr(bias), n, n_pad, K, N, 3, L.ptr(out0), None, None, L.ptr(res), L.ptr(g), L.ptr(b),
               1e-5, L.ptr(y), L.ptr(st), L.ptr(ybf), L.ptr(pos), L.ptr(tp), L.ptr(ypos), stream.cuda_stream)


def chain_b(stream):
    for _ in range(REP):
        L.call("gdmae_dw_gemm", L.ptr(G), L.ptr(X2), n_pad, 512, 256, L.ptr(dW), L.ptr(db), L.ptr(ws), stream.cuda_stream)


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


for rnd in range(3):
    ta = wall(lambda: chain_a(sA))
    tb = wall(lambda: chain_b(sB))
    tab_serial = wall(lambda: (chain_a(sA), chain_b(sA)))
    tab_conc = wall(lambda: (chain_a(sA), chain_b(sB)))
    print(f"per iteration: A alone {ta:.1f} us, B alone {tb:.1f} us, A then B on one stream {tab_serial:.1f} us, A || B {tab_conc:.1f} us")
