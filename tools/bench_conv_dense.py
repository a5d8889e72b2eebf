"""Stand-alone timing of the dense 3 x 3 convolution kernels (csrc/conv_dense.hip) at config D's map size (8 x 496 x 432 sites):
forward / input-gradient launches and the weight gradient, per layer shape; TFLOP/s against the 2.5 PFLOP/s bf16 peak.
GDMAE_LIB=<variant .so> selects a variant library."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd")]
import torch
from gdmae_hip import lib as L
dev = torch.device("cuda:0")
B, H, W = 8, 496, 432
R = B * H * W
lib = L.load()
shapes = [(128, 128, 1), (128, 128, 2), (384, 128, 1), (128, 384, 1), (128, 64, 1), (64, 128, 1), (64, 64, 1), (64, 32, 1), (32, 64, 1)]
tot_f = tot_w = 0.0
for cin, cout, dil in shapes:
    x = torch.randn(R, cin, device=dev).to(torch.bfloat16)
    dy = torch.randn(R, cout, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.02
    y = torch.empty(R, cout, dtype=torch.bfloat16, device=dev)
    packed = torch.empty(lib.gdmae_conv3x3_dense_packed_bytes(cin, cout), dtype=torch.uint8, device=dev)
    L.call("gdmae_conv3x3_dense_pack", L.ptr(w), cin, cout, dil, 0, L.ptr(packed), L.stream())
    def fwd():
        L.call("gdmae_conv3x3_dense", L.ptr(x), B, H, W, cin, cout, dil, L.ptr(packed), None, L.ptr(y), L.stream())
    res = []
    fns = [fwd]
    if cin % 64 == 0:
        dW = torch.zeros(cout, cin, 3, 3, device=dev)
        ws = torch.empty(lib.gdmae_conv3x3_dense_dw_workspace_bytes(B, H, W, cin, cout), dtype=torch.uint8, device=dev)
        def dw():
            L.call("gdmae_conv3x3_dense_bwd_weight", L.ptr(x), L.ptr(dy), B, H, W, cin, cout, cin, cout, dil, L.ptr(dW), L.ptr(ws), L.stream())
        fns.append(dw)
    for f in fns:
        for _ in range(2): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5 * 1e3)
    fl = 2.0 * R * cin * cout * 9
    line = f"{cin:4d} -> {cout:4d} dil {dil}: fwd {res[0]:8.1f} us {fl / res[0] / 1e6:7.1f} TFLOP/s"
    if len(res) > 1:
        line += f"   dW {res[1]:8.1f} us {fl / res[1] / 1e6:7.1f} TFLOP/s"
    print(line, flush=True)
