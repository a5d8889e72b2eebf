"""CPU: the restatements of the un-vendored third-party arithmetic (oracle/thirdparty.py: torch_scatter, spconv 2-D,
pytorch3d chamfer - "parity unpinned") against INDEPENDENT formulations of the same published semantics:
per-segment python loops for the scatters, a per-site neighbour loop over a coordinate dictionary for the two sparse
convolutions (the restatement itself goes through a dense F.conv2d), brute-force ``cdist`` for the Chamfer distance."""
import numpy as np
import torch

from oracle import thirdparty as tp


def _sites(rng, B, Y, X, n):
    keys = rng.choice(B * Y * X, size=n, replace=False)
    keys.sort()
    return torch.from_numpy(np.stack([keys // (Y * X), (keys // X) % Y, keys % X], axis=1).astype(np.int32))


def test_scatter_mean_and_max_match_segment_loops():
    g = torch.Generator().manual_seed(0)
    n, c, m = 500, 7, 40
    src = torch.randn(n, c, generator=g)
    src[::9] = src[1::9][: src[::9].shape[0]]                      # exact ties between rows of different segments
    index = torch.randint(0, m, (n,), generator=g)
    index[index == 5] = 6                                          # an empty segment
    src[index == 3] = 1.25                                         # a segment of ties: the lowest row must win
    mean = tp.scatter_mean(src, index, m)
    mx, arg = tp.scatter_max(src, index, m)
    for s in range(m):
        rows = (index == s).nonzero().view(-1)
        if rows.numel() == 0:
            assert float(mean[s].abs().sum()) == 0 and int(arg[s].min()) == n       # torch_scatter: empty segment -> 0 / out of range
            continue
        seq = torch.zeros(c)
        for r in rows.tolist():                                    # ascending row order = the canonical order
            seq = seq + src[r]
        assert torch.equal(mean[s], seq / rows.numel())
        for k in range(c):
            col = src[rows, k]
            assert float(mx[s, k]) == float(col.max())
            assert int(arg[s, k]) == int(rows[(col == col.max()).nonzero()[0, 0]])
    # gradient reaches the arg-max rows only
    x = src.clone().requires_grad_(True)
    out, a = tp.scatter_max(x, index, m)
    keep = (a < n)
    out[keep].sum().backward()
    want = torch.zeros(n, c)
    for s in range(m):
        for k in range(c):
            if int(a[s, k]) < n:
                want[int(a[s, k]), k] += 1
    assert torch.equal(x.grad, want)


def _neighbour_conv(feat, idx, Y, X, W, out_sites, stride, pad):
    table = {tuple(int(v) for v in r): i for i, r in enumerate(idx.tolist())}
    out = torch.zeros(len(out_sites), W.shape[0], dtype=torch.float64)
    for o, (b, y, x) in enumerate(out_sites):
        for ky in range(3):
            for kx in range(3):
                iy, ix = y * stride - pad + ky, x * stride - pad + kx
                j = table.get((b, iy, ix))
                if j is not None:
                    out[o] += W[:, ky, kx, :].double() @ feat[j].double()
    return out


def test_sparse_convolutions_match_neighbour_loops():
    rng = np.random.default_rng(1)
    B, Y, X, cin, cout = 2, 13, 11, 5, 6                           # odd extents: the strided output grid has a ragged edge
    idx = _sites(rng, B, Y, X, 90)
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(idx.shape[0], cin, generator=g)
    W = torch.randn(cout, 3, 3, cin, generator=g)                  # spconv-2.x layout (Cout, kH, kW, Cin)
    # submanifold: active set unchanged
    sub = tp.subm_conv2d(feat, idx, [Y, X], B, W)
    want = _neighbour_conv(feat, idx, Y, X, W, [tuple(r) for r in idx.tolist()], 1, 1)
    assert float((sub.double() - want).abs().max()) < 1e-5
    # strided k3 s2 p1: output site active iff any of its 9 input taps is active; ascending (b, y, x)
    of, oidx, oshape = tp.sparse_conv2d(feat, idx, [Y, X], B, W, stride=2, padding=1)
    assert oshape == [(Y + 2 - 3) // 2 + 1, (X + 2 - 3) // 2 + 1] == tp.strided_out_shape([Y, X])
    active = set()
    for b, y, x in idx.tolist():
        for ky in range(3):
            for kx in range(3):
                oy, ox = y + 1 - ky, x + 1 - kx
                if oy % 2 == 0 and ox % 2 == 0 and 0 <= oy // 2 < oshape[0] and 0 <= ox // 2 < oshape[1]:
                    active.add((b, oy // 2, ox // 2))
    assert [tuple(r) for r in oidx.tolist()] == sorted(active)
    want = _neighbour_conv(feat, idx, Y, X, W, sorted(active), 2, 1)
    assert float((of.double() - want).abs().max()) < 1e-5
    # densify: features at their sites, zeros elsewhere
    d = tp.densify(feat, idx, [Y, X], B)
    assert d.shape == (B, cin, Y, X) and abs(float(d.double().abs().sum()) - float(feat.double().abs().sum())) < 1e-9
    assert int((d != 0).sum()) == int((feat != 0).sum())
    b, y, x = idx[7].tolist()
    assert torch.equal(d[b, :, y, x], feat[7])


def test_chamfer_matches_cdist_brute_force():
    g = torch.Generator().manual_seed(2)
    N, P1, P2 = 9, 16, 64
    x = torch.randn(N, P1, 3, generator=g, requires_grad=True)
    y = torch.randn(N, P2, 3, generator=g)
    w = (torch.rand(N, generator=g) > 0.4).float()
    loss, _ = tp.chamfer_distance(x, y, weights=w)
    d = torch.cdist(x.detach().double(), y.double()) ** 2
    want = ((d.min(2).values.mean(1) + d.min(1).values.mean(1)) * w.double()).sum() / w.double().sum()
    assert abs(float(loss) - float(want)) < 1e-6 * float(want)
    loss.backward()
    assert float(x.grad[w == 0].abs().sum()) == 0 and float(x.grad[w == 1].abs().sum()) > 0
    # all weights zero: exactly 0, with a graph
    z, _ = tp.chamfer_distance(x, y, weights=torch.zeros(N))
    assert float(z) == 0.0 and z.requires_grad
    # unweighted: mean over the batch
    u, _ = tp.chamfer_distance(x.detach(), y)
    assert abs(float(u) - float((d.min(2).values.mean(1) + d.min(1).values.mean(1)).mean())) < 1e-6
