import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/gd-mae_amd"]
import torch
from gdmae_hip import lib as L
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
bits = torch.randint(-2**31, 2**31 - 1, (1 << 20,), generator=g, dtype=torch.int64).to(torch.int32)
special = torch.tensor([0, -2**31, 0x7F800000, -8388608, 1, 0x00007FFF, 0x00008000, 0x00008001, 0x007FFFFF, 0x7F7FFFFF, 0x7F7F8000, 0x7FC00000, 0x7F800001], dtype=torch.int32)
a = torch.cat([bits, special]).view(torch.float32).to(dev)
out = torch.empty(a.numel(), dtype=torch.bfloat16, device=dev)
L.call("gdmae_add3_to", L.ptr(a), None, 0, None, 0, a.numel(), L.ptr(out), 1, L.stream())
torch.cuda.synchronize()
ref = a.to(torch.bfloat16)
nan = torch.isnan(a)
print("nan agree", torch.equal(torch.isnan(out), nan))
bad = (out.view(torch.int16) != ref.view(torch.int16)) & ~nan
print("bad", int(bad.sum()), "of", a.numel())
idx = bad.nonzero().flatten()[:12]
for i in idx.tolist():
    print(hex(a.view(torch.int32)[i].item() & 0xffffffff), hex(out.view(torch.int16)[i].item() & 0xffff), hex(ref.view(torch.int16)[i].item() & 0xffff))
