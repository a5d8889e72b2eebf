"""GPU parity of the dense 3x3 convolution (csrc/conv_dense.hip, row f1: SSTBEVBackbone / CenterHead / dense conv_out - reference
sst_bev_backbone.py:14-40, center_head.py:20-35, spt_backbone.py:289-291) against torch's CPU convolution in fp64 on the same
bf16-rounded operands: forward, input gradient and weight / bias gradients, through the C ABI and through the autograd wrapper."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def _ref(x, w, b, dil, gy):
    """fp64 CPU reference on the bf16-rounded x / w / gy: y, dx, dw, db."""
    xd = x.to(torch.bfloat16).double().cpu().requires_grad_(True)
    wd = w.to(torch.bfloat16).double().cpu().requires_grad_(True)
    bd = None if b is None else b.double().cpu().requires_grad_(True)
    y = F.conv2d(xd, wd, bd, padding=dil, dilation=dil)
    y.backward(gy.to(torch.bfloat16).double().cpu())
    return y.detach(), xd.grad, wd.grad, None if bd is None else bd.grad


CASES = [(2, 20, 27, 128, 128, 1, False), (1, 17, 16, 128, 128, 2, False), (2, 9, 30, 384, 128, 1, False), (2, 16, 24, 128, 64, 1, True),
         (1, 24, 17, 64, 64, 1, True), (2, 12, 21, 64, 3, 1, True), (1, 8, 8, 64, 2, 1, True), (1, 30, 9, 64, 1, 1, True)]


@pytest.mark.parametrize("B,H,W,cin,cout,dil,bias", CASES)
def test_dense_conv3x3_matches_cpu_fp64(B, H, W, cin, cout, dil, bias):
    from gdmae_hip import dense as gdense
    g = torch.Generator().manual_seed(cin * 7 + cout + dil)
    x = torch.randn(B, cin, H, W, generator=g).to(dev())
    conv = nn.Conv2d(cin, cout, 3, padding=dil, dilation=dil, bias=bias).to(dev())
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (9 * cin)) ** 0.5)
        if bias:
            conv.bias.copy_(torch.randn(cout, generator=g))
    gy = torch.randn(B, cout, H, W, generator=g).to(dev())
    xg = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert gdense.conv3x3_supported(conv, xg)
        y = gdense.conv3x3(conv, xg)
    assert y.shape == (B, cout, H, W) and y.dtype == torch.bfloat16
    y.backward(gy)
    ry, rdx, rdw, rdb = _ref(x, conv.weight.detach(), conv.bias.detach() if bias else None, dil, gy)
    sc = float(ry.abs().max())
    assert float((y.detach().double().cpu() - ry).abs().max()) <= 6e-3 * sc                 # bf16 output rounding: 2^-8 of the largest value
    assert float((xg.grad.double().cpu() - rdx).abs().max()) <= 6e-3 * float(rdx.abs().max())
    # weight gradient: fp32 accumulation of exact bf16 products in a fixed order
    assert float((conv.weight.grad.double().cpu() - rdw).abs().max()) <= 2e-5 * float(rdw.abs().max()) + 1e-6
    if bias:
        assert float((conv.bias.grad.double().cpu() - rdb).abs().max()) <= 2e-5 * float(rdb.abs().max()) + 1e-6
    # repeatable bit for bit
    conv.weight.grad = None
    xg2 = x.clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y2 = gdense.conv3x3(conv, xg2)
    y2.backward(gy)
    assert torch.equal(y2, y) and torch.equal(xg2.grad, xg.grad)


def test_dense_conv3x3_accumulates_into_given_gradient():
    """gdmae_conv3x3_dense_bwd_weight ADDS to dW (the flat optimizer hands its gradient buffer over)."""
    from gdmae_hip import lib as L
    B, H, W, cin, cout = 1, 16, 16, 64, 64
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, H, W, cin, generator=g).to(dev()).to(torch.bfloat16)
    dy = torch.randn(B, H, W, cout, generator=g).to(dev()).to(torch.bfloat16)
    lib = L.load()
    ws = torch.empty(lib.gdmae_conv3x3_dense_dw_workspace_bytes(B, H, W, cin, cout), dtype=torch.uint8, device=dev())
    d0 = torch.zeros(cout, cin, 3, 3, device=dev())
    L.call("gdmae_conv3x3_dense_bwd_weight", L.ptr(x), L.ptr(dy), B, H, W, cin, cout, cin, cout, 1, L.ptr(d0), L.ptr(ws), L.stream())
    d1 = torch.full((cout, cin, 3, 3), 0.5, device=dev())
    L.call("gdmae_conv3x3_dense_bwd_weight", L.ptr(x), L.ptr(dy), B, H, W, cin, cout, cin, cout, 1, L.ptr(d1), L.ptr(ws), L.stream())
    assert torch.allclose(d1 - 0.5, d0, rtol=0, atol=1e-4 * float(d0.abs().max()))
    assert float(d0.abs().max()) > 0


@pytest.mark.parametrize("B,H,W,cin,cout,dil,bias", [(2, 20, 27, 128, 128, 1, False), (1, 17, 16, 128, 128, 2, False), (1, 9, 30, 384, 128, 1, False),
                                                     (1, 24, 17, 64, 64, 1, True), (2, 12, 21, 64, 3, 1, True)])
def test_dense_conv3x3_fp32_grade_matches_cpu_fp64(B, H, W, cin, cout, dil, bias):
    """No autocast: the six-term split form (Conv3x3DenseF32: three bf16 pieces per operand) against the fp64 convolution of the
    UNROUNDED fp32 operands - forward, input gradient, weight / bias gradients to 2e-6 of the largest value, i.e. fp32 round-off (a
    bf16-operand product sits at 4e-3, a two-piece split at 1e-5)."""
    from gdmae_hip import dense as gdense
    g = torch.Generator().manual_seed(cin * 3 + cout + dil)
    x = torch.randn(B, cin, H, W, generator=g).to(dev())
    conv = nn.Conv2d(cin, cout, 3, padding=dil, dilation=dil, bias=bias).to(dev())
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (9 * cin)) ** 0.5)
        if bias:
            conv.bias.copy_(torch.randn(cout, generator=g))
    gy = torch.randn(B, cout, H, W, generator=g).to(dev())
    xg = x.clone().requires_grad_(True)
    assert gdense.conv3x3_supported(conv, xg)
    y = gdense.conv3x3(conv, xg)
    assert y.dtype == torch.float32 and y.shape == (B, cout, H, W)
    y.backward(gy)
    xd, wd = x.double().cpu().requires_grad_(True), conv.weight.detach().double().cpu().requires_grad_(True)
    bd = conv.bias.detach().double().cpu().requires_grad_(True) if bias else None
    ry = F.conv2d(xd, wd, bd, padding=dil, dilation=dil)
    ry.backward(gy.double().cpu())
    for got, ref, nm in ((y.detach(), ry.detach(), "y"), (xg.grad, xd.grad, "dx"), (conv.weight.grad, wd.grad, "dw")):
        err = float((got.double().cpu() - ref).abs().max()) / float(ref.abs().max())
        assert err <= 2e-6, (nm, err)
    if bias:
        assert float((conv.bias.grad.double().cpu() - bd.grad).abs().max()) <= 3e-5 * float(bd.grad.abs().max())


@pytest.mark.parametrize("B,H,W,cin,cout,dil", [(2, 20, 27, 128, 128, 1), (1, 17, 16, 128, 128, 2), (2, 16, 24, 128, 64, 1), (1, 33, 9, 64, 64, 1)])
def test_dense_conv3x3_statistics_epilogue(B, H, W, cin, cout, dil):
    """gdmae_conv3x3_dense_stats: the (256, 2, C) partial rows the epilogue leaves sum to the column sums / sums of squares of the bf16
    output map (partial tiles at the map border excluded from nothing, padded sites from everything), the output equals the plain
    launch's, and BatchNorm + ReLU folded from those rows equals the one folded from a pass over the map."""
    from gdmae_hip import dense as gdense
    g = torch.Generator().manual_seed(cin + cout + dil)
    x = torch.randn(B, cin, H, W, generator=g).to(dev())
    conv = nn.Conv2d(cin, cout, 3, padding=dil, dilation=dil, bias=True).to(dev())
    bn = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).to(dev()).train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y0 = gdense.conv3x3(conv, x)
        y1, part = gdense.conv3x3(conv, x, want_stats=True)
    assert torch.equal(y0, y1) and part.shape == (256, 2, cout)
    rows = y1.permute(0, 2, 3, 1).reshape(-1, cout).double()
    s1, s2 = part[:, 0].double().sum(0), part[:, 1].double().sum(0)
    assert float((s1 - rows.sum(0)).abs().max()) <= 1e-5 * float(rows.abs().sum(0).max())
    assert float((s2 - (rows * rows).sum(0)).abs().max()) <= 1e-5 * float((rows * rows).sum(0).max())
    bn2 = nn.BatchNorm2d(cout, eps=1e-3, momentum=0.01).to(dev()).train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a = gdense.bn_relu_2d(y1, bn, None, part)
        b = gdense.bn_relu_2d(y1, bn2, None, None)
    assert float((a.float() - b.float()).abs().max()) <= 2e-2 * float(b.float().abs().max())       # one bf16 ulp where the fold differs in the last bit
    assert torch.allclose(bn.running_mean, bn2.running_mean, rtol=1e-5, atol=1e-7) and torch.allclose(bn.running_var, bn2.running_var, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("B,H,W,c,dil", [(2, 20, 27, 128, 1), (1, 17, 16, 128, 2), (1, 9, 11, 64, 1)])
def test_shortcut_block_as_one_node_matches_the_two_node_form(B, H, W, c, dil):
    """dense.ConvBNReLUShortcut (x + relu(bn(conv(x))) as ONE autograd node whose input-gradient convolution adds the shortcut's gradient
    in its store pass: gdmae_conv3x3_dense_add) against the two-node form (Conv3x3Dense, then BNReLURows with the residual, the engine
    adding the two gradients of x): the same kernels on the same operands - output, running statistics and the parameter gradients
    identical bit for bit, dx identical up to the one bf16 rounding the engine's addition performs in a different place (the two-node
    form rounds dx_conv, then the sum; the fused store pass does exactly that too: bit-identical as well)."""
    import os
    from gdmae_hip import dense as gdense
    g = torch.Generator().manual_seed(c + dil + H)
    x0 = torch.randn(B, c, H, W, generator=g).to(dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, c, H, W, generator=g).to(dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w0 = torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5
    res = {}
    for fuse in (True, False):
        gdense.FUSE_SHORTCUT = fuse
        try:
            block = nn.Sequential(nn.Conv2d(c, c, 3, padding=dil, dilation=dil, bias=False), nn.BatchNorm2d(c, eps=1e-3, momentum=0.01),
                                  nn.ReLU()).to(dev()).train()
            with torch.no_grad():
                block[0].weight.copy_(w0)
                block[1].weight.copy_(torch.linspace(0.5, 1.5, c))
                block[1].bias.copy_(torch.linspace(-0.3, 0.3, c))
            x = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = gdense.conv_bn_relu(block, x, shortcut=x)
            y.backward(gy)
            res[fuse] = (y.detach().clone(), x.grad.clone(), block[0].weight.grad.clone(), block[1].weight.grad.clone(), block[1].bias.grad.clone(),
                         block[1].running_mean.clone(), block[1].running_var.clone())
        finally:
            gdense.FUSE_SHORTCUT = os.environ.get("GDMAE_DENSE_FUSE", "1") != "0"
    for a, b in zip(res[True], res[False]):
        assert torch.equal(a, b)
    # ... and the sum is what torch computes from the two-node form's pieces: y = relu(bn(conv x)) + x in bf16
    assert y.dtype == torch.bfloat16 and y.shape == x0.shape


def test_fan_out_sums_the_gradients_in_one_pass():
    """ops.FanOut: k handles on a bf16 channels-last map; the backward is gdmae_sum_bf16 (fp32 accumulation, one rounding) - against the
    fp64 sum of the bf16 gradients (half a bf16 ulp), a missing consumer (None gradient) and a consumer whose gradient has another
    layout (the engine's own sum takes over: same values within the bf16 rounding of the pairwise additions)."""
    from gdmae_hip import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 64, 13, 9, generator=g).to(dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ws = [torch.randn(2, 64, 13, 9, generator=g).to(dev()).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(5)]
    ys = ops.FanOut.apply(x, 5)
    loss = sum((y.float() * w.float()).sum() for y, w in zip(ys[:4], ws[:4]))        # the fifth handle has no consumer
    loss.backward()
    ref = sum(w.double() for w in ws[:4])
    assert x.grad.dtype == torch.bfloat16 and x.grad.stride() == x.stride()
    assert float((x.grad.double() - ref).abs().max()) <= 2 ** -8 * float(ref.abs().max())
    x.grad = None
    ys = ops.FanOut.apply(x, 2)
    (ys[0].float() * ws[0].float()).sum().backward(retain_graph=False)
    assert torch.equal(x.grad, ws[0])                                                   # a single gradient passes through untouched
    x.grad = None
    ys = ops.FanOut.apply(x, 2)
    ((ys[0].float() * ws[0].float()).sum() + (ys[1].contiguous().float() * ws[1].contiguous().float()).sum()).backward()
    assert float((x.grad.double() - (ws[0].double() + ws[1].double())).abs().max()) <= 2 ** -7 * float((ws[0].double() + ws[1].double()).abs().max())
