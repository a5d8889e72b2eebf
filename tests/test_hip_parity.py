"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the reference goldens.

Integer/index outputs must be bit-exact; floating point within the tolerance written next to each check
(north_star: Chamfer loss within 1e-4 relative in fp32)."""
import os

import numpy as np
import pytest
import torch

from helpers import assert_sampled_close, load_case
from oracle import gdmae_oracle as orc
from oracle import thirdparty as tp

pytestmark = pytest.mark.gpu
CASES = ["kitti_b2", "kitti_b2_m75", "waymo_b1"]


def dev():
    return torch.device("cuda:0")


def _stage_args(cfg):
    from pcdet.models.backbones_3d.spt_backbone import stage_plan_args
    return stage_plan_args(cfg.BACKBONE_3D.SST_BLOCK_LIST)


@pytest.mark.parametrize("name", CASES)
def test_voxelize_bit_exact(name):
    from gdmae_hip import plan
    z, ds, cfg, _ = load_case(name)
    pts = torch.from_numpy(z["points"])
    vox = plan.voxelize(pts.to(dev()), ds.point_cloud_range, ds.voxel_size, ds.grid_size, int(z["batch_size"]))
    keep, coords = orc.point_coords(pts, ds.point_cloud_range, ds.voxel_size, ds.grid_size)
    vc, inv, rank, cnt = orc.unique_pillars(coords, ds.grid_size)
    assert vox.N == int(z["keep_count"]) == int(keep.sum()) and vox.M == vc.shape[0]
    assert np.array_equal(vox.voxel_coords.cpu().numpy(), z["voxel_coords"])          # reference golden
    assert np.array_equal(vox.inverse.cpu().numpy(), z["inverse"])
    assert torch.equal(vox.points.cpu(), pts[keep]) and torch.equal(vox.point_coords.cpu(), coords)
    assert torch.equal(vox.point_rank.cpu().long(), rank)
    off = torch.cat([torch.zeros(1, dtype=torch.long), cnt.cumsum(0)])
    assert torch.equal(vox.pt_off.cpu().long(), off)
    order = torch.argsort(inv * (len(inv) + 1) + torch.arange(len(inv)))            # by (pillar, index)
    assert torch.equal(vox.pillar_pts.cpu().long(), order)
    mean = tp.scatter_mean(pts[keep][:, 1:], inv, vox.M)
    assert torch.equal(vox.pillar_mean.cpu(), mean), "per-pillar mean must be bit-identical (sequential canonical order)"
    so = torch.searchsorted(vc[:, 0].contiguous(), torch.arange(int(z["batch_size"]) + 1))
    assert torch.equal(vox.sample_off.cpu().long(), so)


def test_voxelize_edge_cases():
    from gdmae_hip import plan
    rng = np.random.default_rng(3)
    pcr, vs, grid = [0, 0, -1, 8, 8, 1], [0.5, 0.5, 2], [16, 16, 1]
    # (a) empty cloud, (b) single point, (c) one crowded pillar with > 256 points + scattered rest, (d) NaN / inf rows
    clouds = [np.zeros((0, 5), np.float32), np.array([[0, 1.2, 3.4, 0.1, 0.5]], np.float32)]
    crowd = np.concatenate([np.full((700, 1), 1.0), rng.uniform(2.0, 2.49, (700, 2)), rng.uniform(-1, 1, (700, 2))], 1)
    rest = np.concatenate([rng.integers(0, 2, (900, 1)).astype(np.float64), rng.uniform(-0.7, 8.3, (900, 2)),
                           rng.uniform(-1.5, 1.5, (900, 1)), rng.uniform(0, 1, (900, 1))], 1)
    mix = np.concatenate([crowd, rest]).astype(np.float32)
    mix = mix[rng.permutation(len(mix))]
    mix = mix[np.argsort(mix[:, 0], kind="stable")]
    bad = mix.copy()
    bad[5, 1] = np.nan
    bad[9, 2] = np.inf
    bad[11, 3] = -np.inf
    clouds += [mix, bad]
    for c in clouds:
        pts = torch.from_numpy(c)
        vox = plan.voxelize(pts.to(dev()), pcr, vs, grid, 2)
        keep, coords = orc.point_coords(pts, pcr, vs, grid)
        vc, inv, rank, cnt = orc.unique_pillars(coords, grid)
        assert vox.N == int(keep.sum()) and vox.M == vc.shape[0]
        assert torch.equal(vox.voxel_coords.cpu(), vc) and torch.equal(vox.inverse.cpu(), inv)
        assert torch.equal(vox.point_rank.cpu().long(), rank)
        if vox.M:
            assert torch.equal(vox.pillar_mean.cpu(), tp.scatter_mean(pts[keep][:, 1:], inv, vox.M))


@pytest.mark.parametrize("name", CASES)
def test_mask_and_partition_bit_exact(name):
    from gdmae_hip import plan
    z, ds, cfg, _ = load_case(name)
    B = int(z["batch_size"])
    vox = plan.voxelize(torch.from_numpy(z["points"]).to(dev()), ds.point_cloud_range, ds.voxel_size, ds.grid_size, B)
    ep = plan.encoder_plan(vox, *_stage_args(cfg), keep_frac=1 - float(z["mask_ratio"]),
                           noise=torch.from_numpy(z["noise"]).to(dev()))
    assert np.array_equal(ep.mask.cpu().numpy().astype(np.uint8), z["mask"])
    vis = np.flatnonzero(z["mask"] == 0)
    assert np.array_equal(ep.tok_pillar.cpu().numpy(), vis)
    for i, st in enumerate(ep.stages):
        assert np.array_equal(st.indices_byx().cpu().numpy(), z[f"st{i}_indices"]), f"stage {i} active set"
        for s, w in enumerate(st.windows):
            assert np.array_equal(w.tok_win.cpu().numpy(), z[f"st{i}_win_id{s}"])
            assert np.array_equal(w.tok_level.cpu().numpy(), z[f"st{i}_level{s}"])
            assert np.array_equal(w.tok_slot.cpu().numpy(), z[f"st{i}_slot{s}"])
            # CSR consistency: every token exactly once, windows contiguous, canonical order inside a window
            csr = w.csr_tok.cpu().numpy()
            assert np.array_equal(np.sort(csr), np.arange(st.n_tok))
            ws, wl = w.win_start.cpu().numpy(), w.win_len.cpu().numpy()
            assert wl.sum() == st.n_tok and sum(w.n_win) == len(ws)
            win_of = w.tok_win.cpu().numpy()
            for a, n in zip(ws[:50], wl[:50]):
                seg = csr[a:a + n]
                assert len(set(win_of[seg])) == 1 and np.all(np.diff(seg) > 0)
        # rulebooks against a brute-force lookup
        idx = st.indices_byx().cpu().long()
        key = {(int(b), int(y), int(x)): t for t, (b, y, x) in enumerate(idx.tolist())}
        nb = st.nbr_subm.cpu().numpy()
        for t in list(range(0, st.n_tok, max(1, st.n_tok // 200))):
            b, y, x = idx[t].tolist()
            exp = [key.get((b, y + ky - 1, x + kx - 1), -1) for ky in range(3) for kx in range(3)]
            assert nb[t].tolist() == exp
        if st.nbr_down is not None:
            pidx = ep.stages[i - 1].indices_byx().cpu().long()
            pkey = {(int(b), int(y), int(x)): t for t, (b, y, x) in enumerate(pidx.tolist())}
            nd, ndt = st.nbr_down.cpu().numpy(), st.nbr_down_t.cpu().numpy()
            for t in list(range(0, st.n_tok, max(1, st.n_tok // 200))):
                b, y, x = idx[t].tolist()
                exp = [pkey.get((b, 2 * y - 1 + ky, 2 * x - 1 + kx), -1) for ky in range(3) for kx in range(3)]
                assert nd[t].tolist() == exp
            # transposed rulebook is the exact inverse relation
            o, k = np.nonzero(nd >= 0)
            assert np.array_equal(ndt[nd[o, k], k], o)
            assert (ndt >= 0).sum() == (nd >= 0).sum()


def test_mask_ties_and_extremes():
    from gdmae_hip import lib as L
    d = dev()
    rng = np.random.default_rng(0)
    for L_, ratio in [(1, 0.5), (7, 0.85), (1000, 0.75), (5000, 0.5), (3000, 0.0), (40, 1.0)]:
        # coarse noise -> many exact ties, some straddling the keep boundary
        noise = torch.from_numpy(rng.integers(0, 50, L_).astype(np.float32) / 50)
        off = torch.tensor([0, L_], dtype=torch.int32, device=d)
        mask = torch.empty(L_, device=d)
        lk = torch.empty(1, dtype=torch.int32, device=d)
        L.call("gdmae_random_mask", L.ptr(noise.to(d)), L.ptr(off), 1, float(1 - ratio), L.ptr(mask), L.ptr(lk), L.stream())
        exp = orc.random_masking(L_, ratio, noise)
        assert torch.equal(mask.cpu(), exp), (L_, ratio)
        assert int(lk.item()) == int(L_ * (1 - ratio))


def test_segment_max_and_gt_grouping_match_oracle():
    from gdmae_hip import ops, plan
    z, ds, cfg, _ = load_case("kitti_b2")
    pts = torch.from_numpy(z["points"])
    vox = plan.voxelize(pts.to(dev()), ds.point_cloud_range, ds.voxel_size, ds.grid_size, int(z["batch_size"]))
    keep, coords = orc.point_coords(pts, ds.point_cloud_range, ds.voxel_size, ds.grid_size)
    vc, inv, rank, cnt = orc.unique_pillars(coords, ds.grid_size)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(vox.N, 64, generator=g)
    x[::7] = x[0]                                   # ties
    xr = x.clone().requires_grad_(True)
    out_ref, arg_ref = tp.scatter_max(xr, inv, vox.M)
    xg = x.to(dev()).requires_grad_(True)
    out = ops.SegmentMax.apply(xg, vox.pt_off, vox.pillar_pts, vox.inverse32)
    assert torch.equal(out.cpu(), out_ref.detach())
    w = torch.randn(vox.M, 64, generator=g)
    (out * w.to(dev())).sum().backward()
    (out_ref * w).sum().backward()
    assert torch.equal(xg.grad.cpu(), xr.grad)
    gt, gi = ops.group_gt_points(vox, 64, want_index=True)
    gt_ref, gi_ref = orc.group_gt_points(pts[keep][:, 1:4], inv, rank, cnt, 64)
    assert np.array_equal(gi.cpu().numpy(), z["gt_group_inds"]) and torch.equal(gi.cpu().long(), gi_ref)
    cen = orc.voxel_centers(vc[:, 1:], ds.voxel_size, ds.point_cloud_range)
    assert torch.equal(gt.cpu(), gt_ref - cen.unsqueeze(1))
    deco = ops.decorate_points(vox)
    sd = orc.seeded_state_dict(orc.param_shapes(cfg, 4), seed=1)
    _, _, deco_ref = orc.dyn_vfe(pts[keep], coords, inv, vox.M, sd, ds.point_cloud_range, ds.voxel_size)
    assert torch.equal(deco.cpu(), deco_ref), "point decoration must be bit-identical"


def test_sparse_conv_matches_oracle():
    from gdmae_hip import ops, plan
    z, ds, cfg, _ = load_case("kitti_b2_m75")
    B = int(z["batch_size"])
    vox = plan.voxelize(torch.from_numpy(z["points"]).to(dev()), ds.point_cloud_range, ds.voxel_size, ds.grid_size, B)
    ep = plan.encoder_plan(vox, *_stage_args(cfg), keep_frac=0.25, noise=torch.from_numpy(z["noise"]).to(dev()))
    g = torch.Generator().manual_seed(1)
    s0, s1 = ep.stages[0], ep.stages[1]
    x = torch.randn(s0.n_tok, 32, generator=g)
    w_sub = torch.randn(48, 3, 3, 32, generator=g) * 0.1
    w_dn = torch.randn(40, 3, 3, 32, generator=g) * 0.1
    idx0 = s0.indices_byx().cpu()
    for (wt, kind) in ((w_sub, "subm"), (w_dn, "down")):
        xr, wr = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
        if kind == "subm":
            ref = tp.subm_conv2d(xr, idx0, [s0.Y, s0.X], B, wr)
            nbr, nbr_t = s0.nbr_subm, torch.flip(s0.nbr_subm, dims=[1]).contiguous()
        else:
            ref, ridx, rshape = tp.sparse_conv2d(xr, idx0, [s0.Y, s0.X], B, wr)
            assert torch.equal(ridx, s1.indices_byx().cpu()) and rshape == [s1.Y, s1.X]
            nbr, nbr_t = s1.nbr_down, s1.nbr_down_t
        xg, wg = x.to(dev()).requires_grad_(True), wt.to(dev()).requires_grad_(True)
        out = ops.SparseConv3x3.apply(xg, wg, nbr, nbr_t)
        assert (out.cpu() - ref.detach()).abs().max() <= 2e-5 * ref.abs().max()
        go = torch.randn(ref.shape, generator=g)
        (out * go.to(dev())).sum().backward()
        (ref * go).sum().backward()
        assert (xg.grad.cpu() - xr.grad).abs().max() <= 2e-5 * xr.grad.abs().max()
        assert (wg.grad.cpu() - wr.grad).abs().max() <= 2e-5 * wr.grad.abs().max()
    # dense() / gather-at-sites round trip
    from pcdet.utils.spconv_utils import SparseConvTensor
    dense = SparseConvTensor(x.to(dev()), ep, 0).dense()
    assert torch.equal(dense.cpu(), tp.densify(x, idx0, [s0.Y, s0.X], B))


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("d,nhead", [(128, 8), (256, 8)])
def test_window_attention_fwd_bwd_matches_oracle(d, nhead, impl):
    """impl 0 (product default): bf16 rows on the bf16 matrix-core kernels at every level (attention_t16 / attention_t32),
    fp32 rows on the exact-fp32 MFMA kernels for T = 32 / 64 and the VALU kernel for T = 16; 1: VALU everywhere; 2: exact-fp32
    MFMA (T >= 32) / VALU (T = 16) also for bf16 rows.  The bf16-row results of the three implementations must agree with
    each other far below bf16 noise."""
    from gdmae_hip import lib as L
    from gdmae_hip import ops, plan
    L.call("gdmae_set_attention_impl", impl)
    from pcdet.models.backbones_3d.spt_backbone import SSTInputLayer
    z, ds, cfg, _ = load_case("waymo_b1")
    B = int(z["batch_size"])
    vox = plan.voxelize(torch.from_numpy(z["points"]).to(dev()), ds.point_cloud_range, ds.voxel_size, ds.grid_size, B)
    # mask only 20 % so that all three occupancy levels (T = 16/32/64) are populated
    ep = plan.encoder_plan(vox, *_stage_args(cfg), keep_frac=0.8, noise=torch.from_numpy(z["noise"]).to(dev()))
    st = ep.stages[0]
    bcfg = cfg.BACKBONE_3D.SST_BLOCK_LIST[0]
    di = orc._drop_info(bcfg)
    g = torch.Generator().manual_seed(5)
    coords = torch.cat([st.indices_byx().cpu().long()[:, :1], torch.zeros(st.n_tok, 1, dtype=torch.long),
                        st.indices_byx().cpu().long()[:, 1:]], 1)
    for shift in (0, 1):
        w = st.windows[shift]
        assert all(n > 0 for n in w.n_win), w.n_win
        part = orc.window_partition(coords, [st.X, st.Y, 1], [8, 8, 1], shift == 1, di)
        x = torch.randn(st.n_tok, d, generator=g)
        pos = orc.pos_embed_table(part["in_win"], d, [8, 8, 1], 1000.0)
        table = SSTInputLayer(bcfg.PREPROCESS).pos_table(d, dev())
        assert (table[w.tok_pos.long()].cpu() - pos).abs().max() < 1e-6
        pfx = "a."
        sd = {pfx + "in_proj_weight": torch.randn(3 * d, d, generator=g) / d ** 0.5,
              pfx + "in_proj_bias": torch.randn(3 * d, generator=g) * 0.1,
              pfx + "out_proj.weight": torch.eye(d), pfx + "out_proj.bias": torch.zeros(d),
              pfx + "tau": torch.full((1, 1, 1), 0.37 if shift == 0 else 0.004)}   # below tau_min in the 2nd pass
        leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = x.clone().requires_grad_(True)
        ref = orc.cosine_window_attention(xr, pos, part, di, leaf, pfx, nhead)
        go = torch.randn(ref.shape, generator=g)
        (ref * go).sum().backward()
        W, bI = sd[pfx + "in_proj_weight"], sd[pfx + "in_proj_bias"]
        qk = (torch.nn.functional.linear(x + pos, W[:2 * d], bI[:2 * d])).to(dev()).requires_grad_(True)
        v = (torch.nn.functional.linear(x, W[2 * d:], bI[2 * d:])).to(dev()).requires_grad_(True)
        tau = sd[pfx + "tau"].to(dev()).requires_grad_(True)
        out = ops.WindowCosineAttention.apply(qk, v, tau, w, nhead, 0.01)
        err = (out.cpu() - ref.detach()).abs().max() / ref.abs().max()
        assert err < 2e-5, err
        (out * go.to(dev())).sum().backward()
        # chain the oracle's dx through the projections to compare dqk / dv
        dW = leaf[pfx + "in_proj_weight"].grad
        dW_hip = torch.cat([qk.grad.cpu().t() @ (x + pos), v.grad.cpu().t() @ x], 0)
        assert (dW_hip - dW).abs().max() / dW.abs().max() < 5e-4
        db = leaf[pfx + "in_proj_bias"].grad
        db_hip = torch.cat([qk.grad.cpu().sum(0), v.grad.cpu().sum(0)])
        assert (db_hip - db).abs().max() / db.abs().max() < 5e-4
        dtau = leaf[pfx + "tau"].grad
        if shift == 0:
            assert abs(float(tau.grad.cpu().reshape(-1)[0] - dtau.reshape(-1)[0])) <= 2e-3 * abs(float(dtau.reshape(-1)[0])) + 1e-6
        else:
            assert float(tau.grad.abs().sum()) == 0.0 and float(dtau.abs().sum()) == 0.0
        # bf16 token rows (throughput mode): same kernels, bf16 HBM I/O, fp32 arithmetic
        qkb, vb = qk.detach().to(torch.bfloat16).requires_grad_(True), v.detach().to(torch.bfloat16).requires_grad_(True)
        outb = ops.WindowCosineAttention.apply(qkb, vb, tau.detach(), w, nhead, 0.01)
        assert outb.dtype == torch.bfloat16
        errb = (outb.float().cpu() - ref.detach()).abs().max() / ref.abs().max()
        assert errb < (3e-2 if shift == 0 else 0.5), errb      # tau = 0.004 -> clamp 0.01 amplifies bf16 q/k rounding 100x
        (outb.float() * go.to(dev())).sum().backward()
        assert torch.isfinite(qkb.grad.float()).all() and torch.isfinite(vb.grad.float()).all()
        if shift == 0:
            assert (vb.grad.float().cpu() - v.grad.cpu()).norm() < 3e-2 * v.grad.norm()
            # same bf16 inputs through the VALU kernels (fp32 arithmetic): implementation differences only
            L.call("gdmae_set_attention_impl", 1)
            qk1, v1 = qkb.detach().clone().requires_grad_(True), vb.detach().clone().requires_grad_(True)
            tau1 = tau.detach().clone().requires_grad_(True)
            out1 = ops.WindowCosineAttention.apply(qk1, v1, tau1, w, nhead, 0.01)
            (out1.float() * go.to(dev())).sum().backward()
            L.call("gdmae_set_attention_impl", impl)
            tau0 = tau.detach().clone().requires_grad_(True)
            qk0, v0 = qkb.detach().clone().requires_grad_(True), vb.detach().clone().requires_grad_(True)
            out0 = ops.WindowCosineAttention.apply(qk0, v0, tau0, w, nhead, 0.01)
            (out0.float() * go.to(dev())).sum().backward()
            for a, b, nm in ((out0, out1, "out"), (qk0.grad, qk1.grad, "dqk"), (v0.grad, v1.grad, "dv")):
                rel = float((a.float() - b.float()).norm() / b.float().norm())
                assert rel < 6e-3, (nm, rel)          # bf16 output rounding is 4e-3 per element
            t0, t1 = float(tau0.grad.sum()), float(tau1.grad.sum())
            assert abs(t0 - t1) <= 2e-2 * abs(t1) + 1e-5, (impl, shift, t0, t1)
    L.call("gdmae_set_attention_impl", 0)


@pytest.mark.parametrize("d,H", [(128, 8), (256, 8)])
def test_packed_window_attention_edge_cases(d, H):
    """The bf16 matrix-core attention kernels on hand-made window lists: full windows, single-token windows, groups whose tokens
    sum to exactly 16 (one packed pass) or to 17 (two passes), a tail group of fewer than four windows, and for the T = 32 / 64
    levels windows of exactly 17 / 32 / 33 / 64 tokens - against the lane-per-query VALU kernels (fp32 arithmetic) on the same bf16
    rows, forward and backward, through the per-level and the all-levels entry points."""
    from gdmae_hip import lib as L
    dv = dev()
    g = torch.Generator().manual_seed(d + H)
    levels = {16: [16, 1, 1, 14, 8, 8, 5, 4, 4, 3, 16, 16, 2, 9, 1, 1, 1, 1, 7],            # 19 windows: tail group of 3
              32: [17, 32, 20, 31, 25],
              64: [33, 64, 40, 63, 50, 64, 35]}
    lens = [n for T in (16, 32, 64) for n in levels[T]]
    n_tok = sum(lens)
    perm = torch.randperm(n_tok, generator=g).int()                       # tokens of a window are scattered rows
    win_start = torch.tensor([sum(lens[:i]) for i in range(len(lens))], dtype=torch.int32)
    win_len = torch.tensor(lens, dtype=torch.int32)
    n_win = [len(levels[T]) for T in (16, 32, 64)]
    qk = torch.randn(n_tok, 2 * d, generator=g).bfloat16().to(dv)
    v = torch.randn(n_tok, d, generator=g).bfloat16().to(dv)
    go = torch.randn(n_tok, d, generator=g).bfloat16().to(dv)
    tau = torch.full((1,), 0.2, device=dv)
    csr, ws, wl = perm.to(dv), win_start.to(dv), win_len.to(dv)
    nw_h, T_h = L.host_i32(n_win), L.host_i32([16, 32, 64])

    def run(impl, levels_entry):
        L.call("gdmae_set_attention_impl", impl)
        out = torch.zeros(n_tok, d, dtype=torch.bfloat16, device=dv)
        dqk = torch.zeros(n_tok, 2 * d, dtype=torch.bfloat16, device=dv)
        dvv = torch.zeros(n_tok, d, dtype=torch.bfloat16, device=dv)
        part = torch.full((sum(n_win) * H,), 123.0, device=dv)             # every slot must be written
        if levels_entry:
            # "lse": the product path - the forward leaves the rows' log-sum-exp, the backward takes it and the forward's output
            # (one launch per direction); True: without them (the backward re-derives the statistics, per-(window, head) kernels)
            lse = torch.full((n_tok, H), float("nan"), device=dv) if levels_entry == "lse" else None
            L.call("gdmae_window_attention_levels_fwd", L.ptr(qk), L.ptr(v), L.ptr(out), 1, L.ptr(csr), L.ptr(ws), L.ptr(wl), 3, nw_h, T_h, d, H,
                   L.ptr(tau), 0.01, L.ptr(lse), L.stream())
            L.call("gdmae_window_attention_levels_bwd", L.ptr(qk), L.ptr(v), L.ptr(go), L.ptr(dqk), L.ptr(dvv), 1, L.ptr(part), L.ptr(csr),
                   L.ptr(ws), L.ptr(wl), 3, nw_h, T_h, d, H, L.ptr(tau), 0.01, L.ptr(out if lse is not None else None), L.ptr(lse), L.stream())
        else:
            base = pb = 0
            for nw, T in zip(n_win, (16, 32, 64)):
                L.call("gdmae_window_attention_fwd", L.ptr(qk), L.ptr(v), L.ptr(out), 1, L.ptr(csr), L.ptr(ws[base:]), L.ptr(wl[base:]), nw, T,
                       d, H, L.ptr(tau), 0.01, L.stream())
                L.call("gdmae_window_attention_bwd", L.ptr(qk), L.ptr(v), L.ptr(go), L.ptr(dqk), L.ptr(dvv), 1, L.ptr(part[pb:]), L.ptr(csr),
                       L.ptr(ws[base:]), L.ptr(wl[base:]), nw, T, d, H, L.ptr(tau), 0.01, L.stream())
                base += nw
                pb += nw * H
        return out.float(), dqk.float(), dvv.float(), float(part.sum())

    try:
        ref = run(1, False)
        for entry in (False, True, "lse"):
            got = run(0, entry)
            for a, b, nm in zip(got[:3], ref[:3], ("out", "dqk", "dv")):
                rel = float((a - b).norm() / b.norm())
                assert rel < 6e-3, (nm, entry, rel)                       # bf16 output rounding is 4e-3 per element
                assert float((a - b).abs().max()) <= 0.06 * float(b.abs().max()), (nm, entry)
            assert abs(got[3] - ref[3]) <= 2e-2 * abs(ref[3]) + 1e-4, (entry, got[3], ref[3])
    finally:
        L.call("gdmae_set_attention_impl", 0)


@pytest.mark.parametrize("P1,P2", [(16, 64), (16, 40), (8, 64), (24, 33)])
def test_chamfer_matches_oracle(P1, P2):
    """(16, *) is the NUM_PRD_POINTS = 16 fast path (transposed-butterfly minima), the others the generic kernel."""
    from gdmae_hip import ops
    g = torch.Generator().manual_seed(2)
    M = 777
    pred = torch.randn(M, P1, 3, generator=g)
    gt = torch.randn(M, P2, 3, generator=g)
    gt[5, 1] = gt[5, 0]                     # exact tie between two ground-truth points -> lowest index wins
    w = (torch.rand(M, generator=g) < 0.8).float()
    pr = pred.clone().requires_grad_(True)
    ref, _ = tp.chamfer_distance(pr, gt, w)
    ref.backward()
    pg = pred.to(dev()).requires_grad_(True)
    loss = ops.ChamferLoss.apply(pg, gt.to(dev()), w.to(dev()))
    loss.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    assert (pg.grad.cpu() - pr.grad).abs().max() <= 1e-5 * pr.grad.abs().max()
    z = ops.ChamferLoss.apply(pg, gt.to(dev()), torch.zeros(M, device=dev()))
    assert float(z) == 0.0


@pytest.mark.parametrize("decoder_impl", ["sparse", "dense"])
@pytest.mark.parametrize("name", CASES + ["once_e_b1"])
def test_full_model_forward_backward_vs_reference_golden(name, decoder_impl):
    """Whole pre-training forward/backward through the pcdet-compatible modules, fp32 mode, with the
    sparse-aware decoder (product default) and with the reference's dense dataflow."""
    import logging
    from pcdet.models import build_network
    z, ds, cfg, shapes = load_case(name)
    torch.manual_seed(0)
    net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev())
    net.backbone_3d.decoder_impl = decoder_impl
    sd = orc.seeded_state_dict(shapes, seed=int(z["seed"]))
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(("running_" in m or "num_batches" in m or m == "global_step") for m in missing)
    net.train()
    bd = {"points": torch.from_numpy(z["points"]).to(dev()), "batch_size": int(z["batch_size"]),
          "mae_noise": torch.from_numpy(z["noise"]).to(dev())}
    ret, tb, disp = net(bd)
    loss = ret["loss"]
    rel = abs(float(loss) - float(z["loss"])) / float(z["loss"])
    assert rel < 1e-4, f"Chamfer loss {float(loss)} vs reference {float(z['loss'])}: rel {rel:.2e}"
    assert np.array_equal(bd["voxel_coords"].cpu().numpy(), z["voxel_coords"])
    assert np.array_equal(bd["voxel_mae_mask"].cpu().numpy().astype(np.uint8), z["mask"])
    assert_sampled_close(bd["pillar_features"], z["pillar_features_s"], z["pillar_features_c"], 1e-4, "pillar_features")
    for i in range(3):
        assert_sampled_close(bd["multi_scale_3d_features"][f"x_conv{i + 1}"].features, z[f"st{i}_features_s"],
                             z[f"st{i}_features_c"], 5e-4, f"stage {i}")
    assert_sampled_close(bd["spatial_features"], z["spatial_features_s"], z["spatial_features_c"], 5e-4, "spatial_features")
    fr = net.backbone_3d.forward_ret_dict
    assert_sampled_close(fr["pred_points"], z["pred_points_s"], z["pred_points_c"], 5e-4, "pred_points")
    assert_sampled_close(fr["gt_points"], z["gt_points_s"], z["gt_points_c"], 1e-6, "gt_points")
    loss.backward()
    names = sorted(shapes)
    params = dict(net.named_parameters())
    gn = np.array([float(params[k].grad.double().norm()) for k in names])
    # tau gradients are sums with heavy cancellation (|g| ~ 1e-4..1e-2): the oracle itself differs from the
    # reference by up to 3e-2 there (tests/golden/make_golden.py), so they only get a coarse check - relative to the
    # value and to the typical tau-gradient magnitude (the error depends on the summation order of the GEMM algorithms
    # that happened to be timed fastest on this box)
    ref = z["grad_norm"]
    is_tau = np.array([k.endswith("tau") for k in names])
    tol = np.where(is_tau, 2.5e-1, 2e-2)
    slack = np.where(is_tau, 2e-2 * np.median(ref[is_tau]), 0.0)
    badn = np.abs(gn - ref) > tol * ref + slack
    assert not badn.any(), [(names[i], gn[i], ref[i]) for i in np.flatnonzero(badn)]
    gh = np.stack([np.pad(params[k].grad.reshape(-1)[:8].cpu().numpy(), (0, max(0, 8 - params[k].grad.numel()))) for k in names])
    scale = np.abs(z["grad_head"]).max(axis=1, keepdims=True) + 1e-8
    bad = np.abs(gh - z["grad_head"]) / scale
    assert (bad.max(axis=1) <= tol * 5).all(), [(names[i], bad[i].max()) for i in np.flatnonzero(bad.max(axis=1) > tol * 5)]


def test_sst_ops_dropin_api_matches_sequential_kernels():
    """pcdet.ops.sst_ops.sst_ops_utils (the reference's own op API) vs a sequential evaluation of its kernels."""
    from pcdet.ops.sst_ops import sst_ops_utils
    rng = np.random.default_rng(9)
    g = np.concatenate([rng.integers(0, 500, 6000), np.full(300, 77), rng.integers(900, 1000, 50)])
    g = g[rng.permutation(len(g))]
    seen, rank = {}, np.empty(len(g), np.int64)
    for i, v in enumerate(g):
        rank[i] = seen.get(v, 0)
        seen[v] = rank[i] + 1
    out = sst_ops_utils.get_inner_win_inds(torch.from_numpy(g).to(dev()))
    assert np.array_equal(out.cpu().numpy(), rank)
    K = 64
    pts = torch.randn(len(g), 3)
    grouped = sst_ops_utils.group_inner_inds(pts.to(dev()), torch.from_numpy(g).to(dev()), K)
    M = int(g.max()) + 1
    exp = -np.ones((M, K), np.int64)
    for i, v in enumerate(g):
        if rank[i] < K:
            exp[v, rank[i]] = i
    for m in range(M):
        c = seen.get(m, 0)
        if 0 < c < K:
            exp[m, c:] = exp[m, np.arange(c, K) % c]
    assert grouped.shape == (M, K, 3)
    ref = pts[torch.from_numpy(exp)]           # -1 indexes the last row, exactly like the reference's points[group_inds]
    assert torch.equal(grouped.cpu(), ref)


def test_colstats_and_strided_rows():
    from gdmae_hip import decoder as gdec
    g = torch.Generator().manual_seed(4)
    for dtype, C, R in [(torch.float32, 128, 5003), (torch.bfloat16, 128, 70001), (torch.bfloat16, 384, 9000),
                        (torch.float32, 384, 777), (torch.float32, 64, 1)]:
        x = (torch.randn(R, C, generator=g) * 3 + 1).to(dtype)
        s1, s2 = gdec.colstats(x.to(dev()))
        xd = x.double()
        assert torch.allclose(s1.cpu(), xd.sum(0), rtol=1e-6, atol=1e-6 * R)
        assert torch.allclose(s2.cpu(), (xd * xd).sum(0), rtol=1e-6, atol=1e-6 * R)
    table = torch.randn(1000, 384, generator=g).to(torch.bfloat16)
    idx = torch.randperm(1000, generator=g)[:300].int()
    sl = gdec._gather_slice(table.to(dev()), idx.to(dev()), 128, 128)
    assert torch.equal(sl.cpu(), table[idx.long(), 128:256])
    dst = table.clone().to(dev())
    src = torch.randn(300, 128, generator=g).to(torch.bfloat16)
    gdec._scatter_slice(src.to(dev()), idx.to(dev()), dst, 256)
    exp = table.clone()
    exp[idx.long(), 256:384] = src
    assert torch.equal(dst.cpu(), exp)


def test_sparse_decoder_equals_dense_decoder_and_updates_running_stats():
    """A/B on the same weights: the exact sparse-aware decoder vs the dense torch dataflow (loss, pillar features,
    every decoder / encoder gradient, BN running statistics)."""
    import copy
    import logging
    from pcdet.models import build_network
    z, ds, cfg, shapes = load_case("kitti_b2_m75")
    res = {}
    for impl in ("sparse", "dense"):
        torch.manual_seed(0)
        net = build_network(cfg, 3, ds, logging.getLogger("t")).to(dev())
        net.load_state_dict(orc.seeded_state_dict(shapes, seed=5), strict=False)
        net.backbone_3d.decoder_impl = impl
        net.train()
        bd = {"points": torch.from_numpy(z["points"]).to(dev()), "batch_size": int(z["batch_size"]),
              "mae_noise": torch.from_numpy(z["noise"]).to(dev())}
        ret, _, _ = net(bd)
        ret["loss"].backward()
        res[impl] = (float(ret["loss"].detach()), bd["voxel_features"].detach().cpu(), bd["spatial_features"].detach().cpu(),
                     {k: p.grad.detach().cpu() for k, p in net.named_parameters()},
                     {k: v.detach().cpu().clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k})
    ls, vs, sfs, gs, rs = res["sparse"]
    ld, vd, sfd, gd, rd = res["dense"]
    assert abs(ls - ld) <= 2e-6 * abs(ld)
    assert (vs - vd).abs().max() <= 2e-5 * vd.abs().max()
    assert (sfs - sfd).abs().max() <= 2e-5 * sfd.abs().max()
    for k in gs:
        tol = 5e-2 if k.endswith("tau") else 2e-3
        assert (gs[k] - gd[k]).norm() <= tol * gd[k].norm() + 1e-7, (k, float((gs[k] - gd[k]).norm() / gd[k].norm()))
    for k in rs:
        assert torch.allclose(rs[k].float(), rd[k].float(), rtol=1e-4, atol=1e-6), k


@pytest.mark.parametrize("d", [128, 256])
@pytest.mark.parametrize("bdt", [torch.float32, torch.bfloat16])
def test_fused_add_layernorm_matches_torch(d, bdt):
    from gdmae_hip import ops
    g = torch.Generator().manual_seed(8)
    n = 3001
    a = torch.randn(n, d, generator=g) * 2
    b = (torch.randn(n, d, generator=g)).to(bdt)
    ln = torch.nn.LayerNorm(d)
    with torch.no_grad():
        ln.weight.copy_(torch.rand(d, generator=g) + 0.5)
        ln.bias.copy_(torch.randn(d, generator=g) * 0.1)
    ar, br = a.clone().requires_grad_(True), b.float().clone().requires_grad_(True)
    ref = ln(ar + br)
    go = torch.randn(n, d, generator=g)
    (ref * go).sum().backward()
    lng = torch.nn.LayerNorm(d).to(dev())
    lng.load_state_dict(ln.state_dict())
    ag, bg = a.to(dev()).requires_grad_(True), b.to(dev()).requires_grad_(True)
    out = ops.add_layer_norm(ag, bg, lng)
    (out * go.to(dev())).sum().backward()
    assert (out.cpu() - ref.detach()).abs().max() < 2e-5
    assert (ag.grad.cpu() - ar.grad).abs().max() < 5e-5 * ar.grad.abs().max() + 1e-6
    tol = 1e-2 if bdt == torch.bfloat16 else 5e-5
    assert (bg.grad.float().cpu() - br.grad).abs().max() < tol * br.grad.abs().max() + 1e-6
    assert (lng.weight.grad.cpu() - ln.weight.grad).abs().max() < 1e-4 * ln.weight.grad.abs().max()
    assert (lng.bias.grad.cpu() - ln.bias.grad).abs().max() < 1e-4 * ln.bias.grad.abs().max()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_fused_vfe_bn_relu_and_segment_max_match_torch(dt):
    """gdmae_hip.vfe: fused BatchNorm1d(train)+ReLU rows and the fused BN+ReLU+per-pillar-max tail vs torch modules."""
    from gdmae_hip import plan, vfe as gvfe
    z, ds, cfg, _ = load_case("kitti_b2")
    pts = torch.from_numpy(z["points"])
    vox = plan.voxelize(pts.to(dev()), ds.point_cloud_range, ds.voxel_size, ds.grid_size, int(z["batch_size"]))
    keep, coords = orc.point_coords(pts, ds.point_cloud_range, ds.voxel_size, ds.grid_size)
    vc, inv, rank, cnt = orc.unique_pillars(coords, ds.grid_size)
    g = torch.Generator().manual_seed(3)
    N, C = vox.N, 128
    x = (torch.randn(N, C, generator=g) * 2 + 0.3).to(dt)
    gamma = torch.rand(C, generator=g) + 0.5
    gamma[3] = -0.7                                   # negative scale: max must follow relu(a x + b), not a*max(x)
    beta = torch.randn(C, generator=g) * 0.2
    tol = 2e-2 if dt == torch.bfloat16 else 2e-4
    # ---- rows
    xr = x.float().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.relu(torch.nn.functional.batch_norm(xr, None, None, gr, br, True, 0.01, 1e-3))
    go = torch.randn(N, C, generator=g)
    (ref * go).sum().backward()
    xg = x.to(dev()).requires_grad_(True)
    gg, bg = gamma.to(dev()).requires_grad_(True), beta.to(dev()).requires_grad_(True)
    out, mean, var = gvfe.BNReLURows.apply(xg, gg, bg, 1e-3)
    (out.float() * go.to(dev())).sum().backward()
    assert (out.float().cpu() - ref.detach()).abs().max() <= tol * ref.abs().max()
    assert (mean.cpu() - x.float().mean(0)).abs().max() < 1e-4 and (var.cpu() - x.float().var(0, unbiased=False)).abs().max() < 1e-3
    assert (xg.grad.float().cpu() - xr.grad).norm() <= tol * xr.grad.norm()
    assert (gg.grad.cpu() - gr.grad).norm() <= tol * gr.grad.norm() and (bg.grad.cpu() - br.grad).norm() <= tol * br.grad.norm()
    # ---- fused tail
    xr = x.float().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    h = torch.relu(torch.nn.functional.batch_norm(xr, None, None, gr, br, True, 0.01, 1e-3))
    ref, _ = tp.scatter_max(h, inv, vox.M)
    go = torch.randn(vox.M, C, generator=g)
    (ref * go).sum().backward()
    xg = x.to(dev()).requires_grad_(True)
    gg, bg = gamma.to(dev()).requires_grad_(True), beta.to(dev()).requires_grad_(True)
    out, mean, var = gvfe.BNReLUSegmentMax.apply(xg, gg, bg, 1e-3, vox.pt_off, vox.pillar_pts, vox.inverse32)
    (out * go.to(dev())).sum().backward()
    assert (out.cpu() - ref.detach()).abs().max() <= tol * ref.abs().max()
    assert (xg.grad.float().cpu() - xr.grad).norm() <= tol * xr.grad.norm()
    assert (gg.grad.cpu() - gr.grad).norm() <= tol * gr.grad.norm() and (bg.grad.cpu() - br.grad).norm() <= tol * br.grad.norm()


@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("impl", ["native", "python"])
def test_fused_encoder_layer_equals_autograd_layer(autocast, impl):
    """gdmae_hip.encoder (native one-call executor / op-by-op Function, both with hand-written backward) vs the same
    layer op by op through autograd."""
    from gdmae_hip import encoder as genc
    from gdmae_hip import plan
    genc.IMPL = impl
    from pcdet.models.backbones_3d.spt_backbone import SSTInputLayer
    from pcdet.models.model_utils.sst_basic_block import EncoderLayer
    z, ds, cfg, _ = load_case("waymo_b1")
    B = int(z["batch_size"])
    vox = plan.voxelize(torch.from_numpy(z["points"]).to(dev()), ds.point_cloud_range, ds.voxel_size, ds.grid_size, B)
    ep = plan.encoder_plan(vox, *_stage_args(cfg), keep_frac=0.8, noise=torch.from_numpy(z["noise"]).to(dev()))
    st = ep.stages[1]
    bcfg = cfg.BACKBONE_3D.SST_BLOCK_LIST[1]
    d = 256
    table = SSTInputLayer(bcfg.PREPROCESS).pos_table(d, dev())
    torch.manual_seed(3)
    layer = EncoderLayer(d, 8, 512, 0.0, "gelu", layer_cfg={"cosine": True, "tau_min": 0.01}).to(dev())
    with torch.no_grad():
        layer.win_attn.self_attn.tau.fill_(0.6)
        layer.win_attn.self_attn.in_proj_bias.normal_(0, 0.1)
        layer.norm1.weight.uniform_(0.5, 1.5)
        layer.norm2.bias.normal_(0, 0.1)
    x = torch.randn(st.n_tok, d, device=dev())
    go = torch.randn(st.n_tok, d, device=dev())
    res = {}
    for fused in (False, True):
        layer.fused = fused
        layer.zero_grad()
        xi = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            y = layer(xi, table, st.windows[1])
        (y.float() * go).sum().backward()
        res[fused] = (y.detach().float(), xi.grad.clone(), {k: p.grad.clone() for k, p in layer.named_parameters()})
    genc.IMPL = "native"
    y0, dx0, g0 = res[False]
    y1, dx1, g1 = res[True]
    tol = 3e-2 if autocast else 2e-4
    assert (y1 - y0).abs().max() <= tol * y0.abs().max()
    assert (dx1 - dx0).norm() <= tol * dx0.norm()
    for k in g0:
        t = 0.15 if k.endswith("tau") else tol
        assert (g1[k] - g0[k]).norm() <= t * g0[k].norm() + 1e-6, (k, float((g1[k] - g0[k]).norm() / g0[k].norm()))


def test_prefetched_plan_is_identical_to_inline_plan():
    """PlanPrefetch (side stream, capacity-sized buffers, async counts readback) must yield exactly the inline
    geometry plan and therefore bit-identical loss; also while the main stream is busy."""
    import dataclasses
    import logging
    from pcdet.models import build_network
    from gdmae_hip import plan as gplan
    z, ds, cfg, shapes = load_case("kitti_b2")
    torch.manual_seed(0)
    net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev())
    net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
    net.train()
    pts = torch.from_numpy(z["points"]).to(dev())
    noise = torch.from_numpy(z["noise"]).to(dev())
    B = int(z["batch_size"])

    def same(a, b, path):
        if isinstance(a, torch.Tensor):
            assert a.shape == b.shape and torch.equal(a, b), path
        elif dataclasses.is_dataclass(a):
            for f in dataclasses.fields(a):
                same(getattr(a, f.name), getattr(b, f.name), f"{path}.{f.name}")
        elif isinstance(a, (list, tuple)):
            assert len(a) == len(b), path
            for i, (x, y) in enumerate(zip(a, b)):
                same(x, y, f"{path}[{i}]")
        else:
            assert a == b, (path, a, b)

    bb = net.backbone_3d
    vox0 = gplan.voxelize(pts, bb.point_cloud_range, bb.voxel_size, bb.grid_size, B)
    from pcdet.models.backbones_3d.spt_backbone import stage_plan_args
    ep0 = gplan.encoder_plan(vox0, *stage_plan_args(bb.model_cfg.SST_BLOCK_LIST), keep_frac=1 - bb.mask_ratio, noise=noise)
    busy = torch.randn(4096, 4096, device=dev())
    for _ in range(20):
        busy = busy @ busy * 1e-3                     # main stream busy while the plan is built
    noise_cap = torch.cat([noise, torch.rand(pts.shape[0] - noise.numel(), device=dev())])   # capacity-sized noise
    pf = bb.prefetch_plan(pts, B, noise=noise_cap)
    vox1, ep1 = pf.finish()
    for name in ("N", "M", "points", "point_coords", "inverse", "inverse32", "voxel_coords", "pillar_cell", "pt_off", "pillar_pts",
                 "point_rank", "sample_off", "pillar_mean", "cell2pillar", "points_pm", "row_pillar"):
        same(getattr(vox0, name), getattr(vox1, name), f"vox.{name}")
    same(ep0.mask, ep1.mask, "mask")
    same(ep0.tok_pillar, ep1.tok_pillar, "tok_pillar")
    for i, (a, b) in enumerate(zip(ep0.stages, ep1.stages)):
        for f in dataclasses.fields(a):
            if f.name == "map":
                assert torch.equal(a.map, b.map)
            else:
                same(getattr(a, f.name), getattr(b, f.name), f"stage{i}.{f.name}")
        # the tables the one-call plan builds in its own kernels: transposed submanifold rulebook, full-resolution sites
        same(torch.flip(a.nbr_subm, dims=[1]).contiguous(), b._nbr_subm_t, f"stage{i}.nbr_subm_t")
        up = bb.grid_size[1] // a.Y
        if up > 1:
            assert b._up_sites[0] == up
            same(gplan.upsample_cells(a.tok_cell, a.Y, a.X, int(up)).reshape(-1), b._up_sites[1], f"stage{i}.up_sites")
    srcs = bb._dec_sources()
    dt0 = gplan.decoder_tiles(ep0, srcs, int(bb.grid_size[1]), int(bb.grid_size[0]))
    dt1 = ep1.dec_tiles
    assert dt1 is not None and dt0.n_act == dt1.n_act and dt0.sources == dt1.sources
    same(dt0.tile_slot, dt1.tile_slot, "dec.tile_slot")
    same(dt0.tile_list, dt1.tile_list, "dec.tile_list")
    ret0, _, _ = net({"points": pts, "batch_size": B, "mae_noise": noise})
    ret1, _, _ = net({"points": pts, "batch_size": B, "_gdmae_vox": vox1, "_gdmae_plan": ep1})
    assert float(ret0["loss"]) == float(ret1["loss"])


# bench (16-bit) mode against the REFERENCE's fp32 goldens.  Round 6: the decoder's forward products (deconvolution rows, tile
# convolution) multiply fp16 instead of bf16 operands - the bf16 rounding of those WEIGHTS was the one systematic term (the same error at
# every site: tools/weight_rounding_full_size.py, tools/oracle_rounding_injection.py) - and the loss went from 5.2e-4 / 2.2e-5 / 2.3e-5 /
# 2.8e-4 to 3.8e-6 / 9.1e-5 / 3.7e-5 / 1.5e-4, and - with DynVFE's rows between its two layers in fp16 as well - the samples of this build
# are 1.4e-4 / 2.3e-4 / 8.1e-5 / 1.5e-4.  What is left on these SMALL cases (1 - 2 frames, 3 - 16 k pillars) is the rounding of the bf16
# activations, which averages over the pillars of a batch: a build with fp16 operands in EVERY forward product (sparse convolutions,
# in-projection, out-projection, feed-forward block; measured, not kept) lands at 2.0e-4 / 3.1e-4 / 4.9e-5 / 5.4e-5 - another sample of the
# same noise, and the fp32 oracle with exact weights and bf16-rounded activations at -1.3e-4 / +1.3e-4 / -4.1e-5 / +1.3e-5.  north_star's
# 1e-4 is asserted where the noise has averaged out: 8 full-size frames (test_full_size_properties.py LOSS_REL, two weight seeds: 5.6e-5,
# 3.8e-5); here the bound is the small-case noise floor, the same for every case.
# The gradient-norm bound of kitti_b2 is a NOISE floor too: when the BatchNorm statistics of the sparse-conv blocks moved
# into the convolution's epilogue (same sums of the same rounded values in another order: they change by 1e-7 relative, checked to
# 1e-6 by test_spconv_implicit_gemm_matches_gathered_product), its worst parameters - the in-projection / out-projection / LayerNorm
# biases of stage 2's first block, column sums over a few hundred rows that cancel almost completely - went from 4.4 % to 9.1 % while
# the other cases stayed at 3 - 4 % (tools/ab_bench_mode_golden.py with GDMAE_SPCONV_STATS=0 / 1: 4.4 | 9.1, 4.0 | 3.6, 2.7 | 3.4 %).
BENCH_LOSS_REL = {"kitti_b2": 4e-4, "kitti_b2_m75": 4e-4, "waymo_b1": 4e-4, "once_e_b1": 4e-4}
BENCH_NORM_REL = {"kitti_b2": 0.18, "kitti_b2_m75": 0.08, "waymo_b1": 0.07, "once_e_b1": 0.09}
BENCH_TAU_ABS = 0.5


def test_bench_mode_loss_deviation_is_scatter_not_bias_and_the_decoder_meets_1e4():
    """What the 16-bit mode's loss deviation on a SMALL case is made of (kitti_b2: 2 frames, 3 272 pillars), over six masking-noise seeds
    against the fp32 oracle run with the same noise (tools/bench_mode_seed_scatter.py / bench_mode_precision_split.py print the full
    tables, profiles/r06_loss_deviation_*.txt keep them):  (1) the whole bench mode: a scatter around ~0 - |mean| <= 1.5e-4 (measured
    -1e-5 over these six seeds, +5e-5 over eight), std <= 4e-4 (2.0e-4; 3.3e-4 before DynVFE's rows became fp16);  (2) everything but the
    encoder stages in 16 bits - DynVFE on fp16 rows, the fp16-operand decoder - with the three stages in fp32: every seed within
    2.5e-4 (measured max 1.3e-4, mean 1e-5, std 7e-5): what is left of the scatter is the bf16 activation stream of the stages, and
    it averages out with the batch (<= 5.6e-5 at 8 full-size frames, test_full_size_properties);  (3) the DECODER alone in 16 bits
    behind an fp32 DynVFE and fp32 stages: every seed within 2e-4 (max 1.1e-4, mean 3e-6) - the part of the path that carried the
    systematic error meets north_star's bound case by case;  (4) everything outside autocast under the same module tree: 1e-6 (the
    fp32 mode, measured 1e-7).  The parts are taken out of the autocast region by wrapping their forwards here; the product code is
    not touched."""
    import logging
    from gdmae_hip import configs, optim, decoder as gdec
    from pcdet.models import build_network
    from pcdet.utils.spconv_utils import replace_feature
    import pcdet.models.backbones_3d.spt_backbone_mae as mae_mod
    z, ds, cfg, shapes = load_case("kitti_b2")
    sd = orc.seeded_state_dict(shapes, seed=int(z["seed"]))
    pts = torch.from_numpy(z["points"])
    B, M = int(z["batch_size"]), int(z["noise"].shape[0])
    torch.manual_seed(0)
    net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev())
    net.load_state_dict(sd, strict=False)
    net.train()
    opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)      # noqa: F841  (flat buffers + bf16 shadows: the bench mode)
    fp32_parts = set()

    def part(fn, tag, to32=None, to16=None):
        def wrapped(*a, **k):
            if tag in fp32_parts:
                with torch.autocast("cuda", enabled=False):
                    return fn(*(to32(a) if to32 else a), **k)
            return fn(*(to16(a) if to16 else a), **k)
        return wrapped

    sp32 = lambda a: (replace_feature(a[0], a[0].features.float()),) + tuple(a[1:])
    hid = lambda dt: (lambda a: tuple(a[:3]) + ([replace_feature(h, h.features.to(dt)) for h in a[3]],) + tuple(a[4:]))
    orig_dec = gdec.sparse_decoder
    try:
        net.vfe.forward = part(net.vfe.forward, "vfe")
        for i, blk in enumerate(net.backbone_3d.sst_blocks):
            blk.forward = part(blk.forward, f"stage{i}", sp32)
        # (fp32 stage outputs would send a 16-bit decoder's deconvolutions down the bf16-weight fallback path: hand it bf16 rows)
        mae_mod.gdec.sparse_decoder = part(orig_dec, "decoder", hid(torch.float32), hid(torch.bfloat16))
        K = 6
        noises = [torch.rand(M, generator=torch.Generator().manual_seed(1000 + s)) for s in range(K)]
        refs = []
        for nz in noises:
            with torch.no_grad():
                o = orc.forward(pts, B, cfg, {k: v.clone() for k, v in sd.items()}, ds.point_cloud_range, ds.voxel_size, ds.grid_size, noise=nz)
            refs.append(float(o["loss"]))
        res = {}
        for label, on in (("bench", set()), ("stages32", {"stage0", "stage1", "stage2"}), ("decoder16", {"vfe", "stage0", "stage1", "stage2"}),
                          ("all32", {"vfe", "stage0", "stage1", "stage2", "decoder"})):
            fp32_parts.clear()
            fp32_parts.update(on)
            dv = []
            for nz, ref in zip(noises, refs):
                bd = {"points": pts.to(dev()), "batch_size": B, "mae_noise": nz.to(dev())}
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                    ret, _, _ = net(bd)
                dv.append((float(ret["loss"]) - ref) / ref)
            res[label] = np.array(dv)
            print(f"[loss deviation over {K} mask seeds, kitti_b2] {label}: mean {res[label].mean():+.2e} std {res[label].std():.2e} max {np.abs(res[label]).max():.2e}")
    finally:
        mae_mod.gdec.sparse_decoder = orig_dec
    assert abs(res["bench"].mean()) <= 1.5e-4 and res["bench"].std() <= 4e-4, res["bench"]
    assert np.abs(res["stages32"]).max() <= 2.5e-4, res["stages32"]
    assert np.abs(res["decoder16"]).max() <= 2e-4, res["decoder16"]
    assert np.abs(res["all32"]).max() <= 1e-6, res["all32"]


@pytest.mark.parametrize("name", CASES + ["once_e_b1"])
def test_bench_mode_gradients_reach_every_parameter(name):
    """The configuration bench.py times (flat optimizer with bf16 weight shadows + bf16 autocast + fused layers) against the fp32
    goldens generated from the unmodified reference: every parameter must receive its gradient (a detached shadow silently dropping
    one is 'work skipped'); loss, per-parameter gradient norms and temperature gradients within 2 x the measured bf16 deviations."""
    import logging
    from gdmae_hip import configs, optim
    from pcdet.models import build_network
    z, ds, cfg, shapes = load_case(name)
    torch.manual_seed(0)
    net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev())
    net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
    net.train()
    opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)
    opt.zero_grad()
    bd = {"points": torch.from_numpy(z["points"]).to(dev()), "batch_size": int(z["batch_size"]),
          "mae_noise": torch.from_numpy(z["noise"]).to(dev())}
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ret, _, _ = net(bd)
    loss_rel = abs(float(ret["loss"].detach()) - float(z["loss"])) / float(z["loss"])
    ret["loss"].backward()
    names = sorted(shapes)
    params = dict(net.named_parameters())
    gn = np.array([float(params[k].grad.double().norm()) for k in names])
    assert np.isfinite(gn).all()
    assert (gn > 0).all(), [names[i] for i in np.flatnonzero(gn == 0)]
    rel = np.abs(gn - z["grad_norm"]) / (z["grad_norm"] + 1e-12)
    # tau gradients are heavily cancelling sums of O(1e-4) on an ABSOLUTE noise floor set by the bf16 q / k / v rows (DESIGN.md section
    # 5): bounded against the largest |dtau| of the case, like at full size (test_full_size_properties.py TAU_ABS)
    nt = np.array([not k.endswith("tau") for k in names])
    tau_dev = np.abs(gn[~nt] - z["grad_norm"][~nt]).max() / z["grad_norm"][~nt].max()
    print(f"[bench mode vs fp32 golden, {name}] loss rel {loss_rel:.3e}, worst gradient-norm deviation (tau excluded) {rel[nt].max():.3e}, "
          f"tau: worst | |g| - |g_ref| | / max |g_ref| {tau_dev:.3e}")
    assert loss_rel <= BENCH_LOSS_REL[name], loss_rel
    assert (rel[nt] <= BENCH_NORM_REL[name]).all(), [(names[i], rel[i]) for i in np.flatnonzero(nt & (rel > BENCH_NORM_REL[name]))]
    assert tau_dev <= BENCH_TAU_ABS, tau_dev


@pytest.mark.parametrize("bdt", [torch.float32, torch.bfloat16])
def test_border_sums_match_torch(bdt):
    """gdmae_border_sums (edge / corner sums of the conv output map and of the pillar rows on the border)."""
    from gdmae_hip import lib as L
    torch.manual_seed(5)
    B, H, W, C = 3, 20, 28, 128
    Y = torch.randn(B, H, W, C, device=dev()).to(bdt)
    cells = torch.randperm(B * H * W, device=dev())[:700].sort().values.int()
    cells = torch.unique(torch.cat([cells, torch.tensor([0, W - 1, (H - 1) * W, H * W - 1, H * W + 5, 2 * H * W + W * 3],
                                                        device=dev(), dtype=torch.int32)])).int()
    M = cells.numel()
    rows = torch.randn(M, C, device=dev())
    out = torch.empty(16, C, dtype=torch.float64, device=dev())
    ws = torch.empty(L.load().gdmae_border_sums_workspace_bytes(B, C), dtype=torch.uint8, device=dev())
    L.call("gdmae_border_sums", L.ptr(Y.view(-1, C)), int(bdt == torch.bfloat16), None, None, L.ptr(rows), L.ptr(cells), M, B, H, W, C,
           L.ptr(out), L.ptr(ws), L.stream())
    Yd = Y.double()
    refY = torch.stack([Yd[:, 0].sum((0, 1)), Yd[:, H - 1].sum((0, 1)), Yd[:, :, 0].sum((0, 1)), Yd[:, :, W - 1].sum((0, 1)),
                        Yd[:, 0, 0].sum(0), Yd[:, 0, W - 1].sum(0), Yd[:, H - 1, 0].sum(0), Yd[:, H - 1, W - 1].sum(0)])
    x = cells.long() % W
    y = (cells.long() // W) % H
    y0, yl, x0, xl = y == 0, y == H - 1, x == 0, x == W - 1
    masks = torch.stack([y0, yl, x0, xl, y0 & x0, y0 & xl, yl & x0, yl & xl]).double()
    refR = masks @ rows.double()
    assert int(masks.sum()) > 20
    assert torch.allclose(out[:8], refY, rtol=1e-5, atol=1e-4)
    assert torch.allclose(out[8:], refR, rtol=1e-5, atol=1e-4)


def test_stage_executor_equals_per_layer_calls():
    """gdmae_encoder_stage_fwd/bwd (4 layers per call, prep_tokens / add3 folded into the neighbouring LayerNorm passes)
    vs one gdmae_encoder_layer call per layer, in the bench configuration: identical arithmetic -> identical results."""
    import logging
    from gdmae_hip import configs, optim
    from gdmae_hip import encoder as genc
    from pcdet.models import build_network
    from gdmae_hip import lib as glib
    z, ds, cfg, shapes = load_case("waymo_b1")
    res = {}
    glib.call("gdmae_encoder_set_layer_path", 0)      # the launch-per-product sequence: same rounding points as the per-layer calls
    for stage in (True, False):
        genc.STAGE = stage
        torch.manual_seed(0)
        net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev())
        net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
        net.train()
        opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)
        opt.zero_grad()
        bd = {"points": torch.from_numpy(z["points"]).to(dev()), "batch_size": int(z["batch_size"]),
              "mae_noise": torch.from_numpy(z["noise"]).to(dev())}
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ret, _, _ = net(bd)
        ret["loss"].backward()
        res[stage] = (float(ret["loss"]), opt.flat_grad.clone())
    genc.STAGE = True
    glib.call("gdmae_encoder_set_layer_path", -1)
    assert res[True][0] == res[False][0]
    g1, g0 = res[True][1], res[False][1]
    assert float((g1 - g0).norm()) <= 1e-6 * float(g0.norm()), float((g1 - g0).norm() / g0.norm())


@pytest.mark.parametrize("cin,s,n", [(128, 1, 4097), (256, 2, 5000), (256, 4, 3001), (128, 2, 63)])
def test_deconv_rows_match_conv_transpose(cin, s, n):
    """gdmae_deconv_rows_* (csrc/rows_gemm.hip: the decoder's ConvTranspose2d(k = s, stride s) blocks on token rows, forward /
    input gradient / weight gradient, row counts that are not multiples of any tile) against F.conv_transpose2d on the same bf16
    values in fp64: every token placed on its own cell of an (n, 1) map, so the s x s output patch of a token is row block
    (token, dy * s + dx) of P."""
    import torch.nn.functional as F
    from gdmae_hip import lib as L
    cout = 128
    g = torch.Generator().manual_seed(cin + s)
    w = (torch.randn(cin, cout, s, s, generator=g) * 0.05).to(dev())
    x = torch.randn(n, cin, generator=g).to(dev()).to(torch.bfloat16)
    dP = torch.randn(n * s * s, cout, generator=g).to(dev()).to(torch.bfloat16)
    lib = L.load()
    nb = lib.gdmae_deconv_rows_packed_bytes(cin, cout, s)
    pf, pb = (torch.empty(nb, dtype=torch.uint8, device=dev()) for _ in range(2))
    L.call("gdmae_deconv_rows_pack", L.ptr(w), cin, cout, s, L.ptr(pf), L.ptr(pb), L.stream())
    P = torch.empty(n * s * s, cout, dtype=torch.bfloat16, device=dev())
    L.call("gdmae_deconv_rows_fwd", L.ptr(x), n, cin, cout, s, L.ptr(pf), L.ptr(P), L.stream())
    dX = torch.empty(n, cin, dtype=torch.bfloat16, device=dev())
    L.call("gdmae_deconv_rows_bwd_input", L.ptr(dP), n, cin, cout, s, L.ptr(pb), L.ptr(dX), L.stream())
    dW = torch.full((cin, cout, s, s), 0.5, dtype=torch.float32, device=dev())          # accumulated into
    ws = torch.empty(lib.gdmae_deconv_rows_dw_workspace_bytes(n, cin, cout, s), dtype=torch.uint8, device=dev())
    L.call("gdmae_deconv_rows_bwd_weight", L.ptr(x), L.ptr(dP), n, cin, cout, s, L.ptr(dW), L.ptr(ws), L.stream())
    # reference: the same bf16 operand values in fp64 through torch's transposed convolution and its autograd
    wq = w.to(torch.bfloat16).double().cpu().requires_grad_(True)
    xq = x.double().cpu().requires_grad_(True)
    y = F.conv_transpose2d(xq.t().reshape(1, cin, n, 1), wq, stride=s)                     # (1, cout, n s, s)
    ref = y[0].reshape(cout, n, s, s).permute(1, 2, 3, 0).reshape(n * s * s, cout)       # row (token, dy * s + dx)
    assert float((P.double().cpu() - ref.detach()).abs().max()) <= 8e-3 * float(ref.detach().abs().max())     # bf16 rounding of the result
    (ref * dP.double().cpu()).sum().backward()
    assert float((dX.double().cpu() - xq.grad).abs().max()) <= 8e-3 * float(xq.grad.abs().max())
    got = dW.double().cpu() - 0.5
    assert float((got - wq.grad).abs().max()) <= 2e-5 * float(wq.grad.abs().max()) + 1e-6 * n, float((got - wq.grad).abs().max())
    dW2 = torch.full_like(dW, 0.5)
    L.call("gdmae_deconv_rows_bwd_weight", L.ptr(x), L.ptr(dP), n, cin, cout, s, L.ptr(dW2), L.ptr(ws), L.stream())
    assert torch.equal(dW, dW2)                                                           # fixed summation order


@pytest.mark.parametrize("M,N,K", [(4099, 256, 128), (777, 48, 128), (130, 2304, 1152), (20000, 64, 11 * 8)])
def test_fp32_gemm_is_exact_fp32_on_own_kernel(M, N, K):
    """gdmae_gemm with fp32 operands (the parity mode's token / im2col products: csrc/gemm_f32.hip, exact-fp32 MFMA, fixed order)
    against the fp64 product: all four transpose combinations the library's callers use, bias, ragged extents; and the split-K
    weight-gradient entry.  1e-6 relative to the row scale = fp32 accumulation of K <= 1152 terms; bit-identical when repeated."""
    from gdmae_hip import lib as L
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).to(dev())
    b = torch.randn(K, N, generator=g).to(dev())
    bias = torch.randn(N, generator=g).to(dev())
    ws = torch.empty(L.load().gdmae_gemm_workspace_bytes(), dtype=torch.uint8, device=dev())
    ref = a.double() @ b.double()
    tol = 2e-6 * float(ref.abs().max()) * max(1.0, (K / 128) ** 0.5)
    for ta, tb in ((0, 0), (0, 1), (1, 0), (1, 1)):
        A = a.t().contiguous() if ta else a
        B = b.t().contiguous() if tb else b
        for bs in (None, bias):
            out = torch.full((M, N), float("nan"), device=dev())
            L.call("gdmae_gemm", L.ptr(A), L.ptr(B), L.ptr(out), M, N, K, ta, tb, 0, 0, None if bs is None else L.ptr(bs), L.ptr(ws), L.stream())
            want = ref if bs is None else ref + bs.double()
            assert float((out.double() - want).abs().max()) <= tol, (ta, tb, bs is not None, float((out.double() - want).abs().max()), tol)
            out2 = torch.empty_like(out)
            L.call("gdmae_gemm", L.ptr(A), L.ptr(B), L.ptr(out2), M, N, K, ta, tb, 0, 0, None if bs is None else L.ptr(bs), L.ptr(ws), L.stream())
            assert torch.equal(out, out2)
    # split-K weight gradient: C (m, n) = A^T B over the long row extent
    if N % 8 == 0 and K % 8 == 0:
        wsk = torch.empty(L.load().gdmae_gemm_tn_splitk_workspace_bytes(M, K, N), dtype=torch.uint8, device=dev())
        x2 = torch.randn(M, N, generator=g).to(dev())
        c = torch.empty(K, N, device=dev())
        L.call("gdmae_gemm_tn_splitk", L.ptr(a), L.ptr(x2), L.ptr(c), M, K, N, 0, 0, L.ptr(wsk), L.stream())
        want = a.double().t() @ x2.double()
        assert float((c.double() - want).abs().max()) <= 3e-6 * float(want.abs().max()) * max(1.0, (M / 1024) ** 0.5)


@pytest.mark.parametrize("n", [131072 + 37, 700])
def test_pred_head_matches_fp64_linear(n):
    """gdmae_pred_head_* (nn.Linear(128 -> 48) on fp32 rows, csrc/rows_gemm.hip): the forward (three-term split-bf16 product, fp32 bias)
    against the fp64 product of the UNROUNDED fp32 operands to 2e-5 of the largest value (a bf16-operand product sits at 4e-3); the
    backward against the fp64 products of the bf16-rounded operands: input gradient to fp32 round-off, weight / bias gradients
    accumulated, repeatable."""
    from gdmae_hip import lib as L
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, 128, generator=g).to(dev())
    W = (torch.randn(48, 128, generator=g) * 0.1).to(dev())
    b = torch.randn(48, generator=g).to(dev())
    dy = torch.randn(n, 48, generator=g).to(dev()).to(torch.bfloat16)
    lib = L.load()
    packed = torch.empty(lib.gdmae_pred_head_packed_bytes(), dtype=torch.uint8, device=dev())
    L.call("gdmae_pred_head_pack", L.ptr(W), L.ptr(b), 128, 48, L.ptr(packed), L.stream())
    y = torch.empty(n, 48, dtype=torch.bfloat16, device=dev())
    xb = torch.empty(n, 128, dtype=torch.bfloat16, device=dev())
    yf = torch.empty(n, 48, dtype=torch.float32, device=dev())
    L.call("gdmae_pred_head_fwd", L.ptr(x), n, 48, L.ptr(packed), L.ptr(y), L.ptr(xb), L.ptr(yf), L.stream())
    assert torch.equal(xb, x.to(torch.bfloat16)) and torch.equal(y, yf.to(torch.bfloat16))
    yf2 = torch.empty_like(yf)
    L.call("gdmae_pred_head_fwd", L.ptr(x), n, 48, L.ptr(packed), None, None, L.ptr(yf2), L.stream())
    assert torch.equal(yf2, yf)
    ref = x.double() @ W.double().t() + b.double()
    assert float((yf.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    xq, Wq = x.to(torch.bfloat16).double(), W.to(torch.bfloat16).double()
    dx = torch.empty(n, 128, dtype=torch.float32, device=dev())
    dW = torch.full((48, 128), 0.25, dtype=torch.float32, device=dev())
    db = torch.full((48,), -1.0, dtype=torch.float32, device=dev())
    ws = torch.empty(lib.gdmae_pred_head_bwd_workspace_bytes(n), dtype=torch.uint8, device=dev())
    L.call("gdmae_pred_head_bwd", L.ptr(dy), 0, None, None, None, L.ptr(xb), n, 48, L.ptr(packed), L.ptr(dx), L.ptr(dW), L.ptr(db), L.ptr(ws), L.stream())
    assert float((dx.double() - dy.double() @ Wq).abs().max()) <= 1e-5 * float((dy.double() @ Wq).abs().max())
    rw, rb = dy.double().t() @ xq, dy.double().sum(0)
    assert float(((dW.double() - 0.25) - rw).abs().max()) <= 2e-5 * float(rw.abs().max()) + 1e-7 * n
    assert float(((db.double() + 1.0) - rb).abs().max()) <= 2e-5 * float(rb.abs().max()) + 1e-7 * n
    dW2, db2 = torch.full_like(dW, 0.25), torch.full_like(db, -1.0)
    L.call("gdmae_pred_head_bwd", L.ptr(dy), 0, None, None, None, L.ptr(xb), n, 48, L.ptr(packed), None, L.ptr(dW2), L.ptr(db2), L.ptr(ws), L.stream())
    assert torch.equal(dW, dW2) and torch.equal(db, db2)
    # fp32 gradient rows (what the Chamfer backward hands over): rounded inside the launch, identical results
    dyf, dyb = dy.float(), torch.empty_like(dy)
    dx3, dW3, db3 = torch.empty_like(dx), torch.full_like(dW, 0.25), torch.full_like(db, -1.0)
    L.call("gdmae_pred_head_bwd", L.ptr(dyf), 1, L.ptr(dyb), None, None, L.ptr(xb), n, 48, L.ptr(packed), L.ptr(dx3), L.ptr(dW3), L.ptr(db3), L.ptr(ws),
           L.stream())
    assert torch.equal(dyb, dy) and torch.equal(dx3, dx) and torch.equal(dW3, dW) and torch.equal(db3, db)
    # ... with the two device scalars of the Chamfer mean folded in: dY = (2 x 0.25) x unscaled rows
    sa, sb = torch.tensor([2.0], device=dev()), torch.tensor([0.25], device=dev())
    L.call("gdmae_pred_head_bwd", L.ptr(dyf * 2.0), 1, L.ptr(dyb), L.ptr(sa), L.ptr(sb), L.ptr(xb), n, 48, L.ptr(packed), L.ptr(dx3), L.ptr(dW3),
           L.ptr(db3), L.ptr(ws), L.stream())
    assert torch.equal(dyb, dy) and torch.equal(dx3, dx)


def test_packed_weight_images_follow_weight_changes_between_optimizer_steps():
    """The packed MFMA weight images (encoder layers, sparse convolutions, decoder conv_out) are refreshed by the optimizer once
    per step; a weight change through torch in between - load_state_dict after a first forward (mid-training resume) - keeps every
    data_ptr, so a registry keyed by pointers alone would run the implicit sparse convolutions and the token GEMMs on the OLD
    weights while the bf16 shadows see the new ones.  Loss and gradients after the reload must equal those of a model built with
    the new weights from the start; a change through .data needs FlatAdamOneCycle.refresh_weights()."""
    import logging
    from gdmae_hip import configs, optim
    from pcdet.models import build_network
    z, ds, cfg, shapes = load_case("waymo_b1")
    sd_a = orc.seeded_state_dict(shapes, seed=int(z["seed"]))
    sd_b = orc.seeded_state_dict(shapes, seed=int(z["seed"]) + 5)

    def run(net, opt):
        opt.zero_grad()
        bd = {"points": torch.from_numpy(z["points"]).to(dev()), "batch_size": int(z["batch_size"]),
              "mae_noise": torch.from_numpy(z["noise"]).to(dev())}
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ret, _, _ = net(bd)
        ret["loss"].backward()
        return float(ret["loss"]), opt.flat_grad.clone()

    def fresh(sd):
        torch.manual_seed(0)
        net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev()).train()
        net.load_state_dict(sd, strict=False)
        return net, optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)

    net_b, opt_b = fresh(sd_b)
    l_ref, g_ref = run(net_b, opt_b)
    net, opt = fresh(sd_a)
    l_a, _ = run(net, opt)                                   # registers + packs every image from weights A
    assert abs(l_a - l_ref) > 1e-3 * abs(l_ref)              # the two weight sets really differ
    net.load_state_dict(sd_b, strict=False)                  # in place: same storage, version counters bumped
    l1, g1 = run(net, opt)
    assert l1 == l_ref, (l1, l_ref)
    assert torch.equal(g1, g_ref)
    # ... and through .data (no version bump) with the explicit refresh
    with torch.no_grad():
        for k, p in net.named_parameters():
            p.data.copy_(sd_a[k].to(p.device))
    opt.refresh_weights()
    l2, _ = run(net, opt)
    assert l2 == l_a, (l2, l_a)


def test_plan_layouts_are_cached_per_capacity_bucket():
    """Real batches differ in point count at every step: the plan layout (arena offsets) is cached per capacity bucket of 16384
    points in a bounded LRU, the exact count travels with the call, and results do not depend on which layout served the call; the
    mask ratio is part of the key (a second model with another ratio and the same shapes gets its own entry)."""
    from gdmae_hip import configs, plan as gplan, synth
    from pcdet.models.backbones_3d.spt_backbone import stage_plan_args
    cfg, ds, skw = configs.named_config("A")
    pts = torch.from_numpy(synth.synth_batch(3, 2, ds.point_cloud_range, **skw)).to(dev())
    args = (ds.point_cloud_range, ds.voxel_size, ds.grid_size, 2, *stage_plan_args(cfg.BACKBONE_3D.SST_BLOCK_LIST))
    n0 = pts.shape[0]
    gplan._PLAN_SHAPES.clear()
    outs = []
    for n in (n0, n0 - 1, n0 - 777, n0 - 16385):
        noise = torch.rand(n0, generator=torch.Generator().manual_seed(5)).to(dev())
        vox, ep = gplan.PlanPrefetch(pts[:n].contiguous(), *args, keep_frac=0.5, noise=noise).finish()
        ref_v = gplan.voxelize(pts[:n].contiguous(), ds.point_cloud_range, ds.voxel_size, ds.grid_size, 2)
        assert vox.N == ref_v.N and vox.M == ref_v.M and torch.equal(vox.voxel_coords, ref_v.voxel_coords)
        assert torch.equal(vox.inverse, ref_v.inverse)
        outs.append((vox.N, vox.M, ep.stages[0].n_tok))
    assert len(gplan._PLAN_SHAPES) == 2, list(gplan._PLAN_SHAPES)      # three counts share a bucket, the fourth is one bucket lower
    gplan.PlanPrefetch(pts, *args, keep_frac=0.25).finish()
    assert len(gplan._PLAN_SHAPES) == 3
    # many prefetches in flight before any finish(): every one owns its pinned read-back buffer
    pend = [gplan.PlanPrefetch(pts[:n0 - 100 * i].contiguous(), *args, keep_frac=0.5) for i in range(12)]
    got = [p.finish()[0].N for p in pend]
    want = [gplan.voxelize(pts[:n0 - 100 * i].contiguous(), ds.point_cloud_range, ds.voxel_size, ds.grid_size, 2).N for i in range(12)]
    assert got == want
    for i in range(40):                                                # the cache stays bounded
        gplan._plan_shape(20000 + 16384 * i, pts.shape[1], 2, *args[:3], *args[4:], 0.5, None)
    assert len(gplan._PLAN_SHAPES) <= gplan._PLAN_SHAPES_MAX


@pytest.mark.parametrize("name", ["waymo_b1", "once_e_b1"])
def test_fused_layer_path_matches_launch_per_product_path(name):
    """The stage as three fused launches per layer and direction with a bf16 residual stream (csrc/layer_fused.hip,
    gdmae_encoder_set_layer_path(1): d = 128 and d = 256 stages) against the launch-per-product sequence with the fp32 stream
    (path 0) in the bench configuration, and both against the fp32 reference golden: the difference between the two paths is the
    bf16 rounding of the stream (2^-9 per half layer on O(1) rows), i.e. of the size of either path's own distance to fp32."""
    import logging
    from gdmae_hip import configs, optim
    from gdmae_hip import lib as glib
    from pcdet.models import build_network
    z, ds, cfg, shapes = load_case(name)
    res = {}
    try:
        for path in (0, 1):
            glib.call("gdmae_encoder_set_layer_path", path)
            torch.manual_seed(0)
            net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev())
            net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
            net.train()
            opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)
            opt.zero_grad()
            bd = {"points": torch.from_numpy(z["points"]).to(dev()), "batch_size": int(z["batch_size"]),
                  "mae_noise": torch.from_numpy(z["noise"]).to(dev())}
            with torch.autocast("cuda", dtype=torch.bfloat16):
                ret, _, _ = net(bd)
            ret["loss"].backward()
            names = sorted(shapes)
            params = dict(net.named_parameters())
            res[path] = (float(ret["loss"]), opt.flat_grad.clone(), np.array([float(params[k].grad.double().norm()) for k in names]), names)
    finally:
        glib.call("gdmae_encoder_set_layer_path", -1)
    l0, g0, n0, names = res[0]
    l1, g1, n1, _ = res[1]
    ref = float(z["loss"])
    print(f"[{name}] loss fp32 golden {ref:.6f}  path0 {l0:.6f} ({abs(l0 - ref) / ref:.2e})  path1 {l1:.6f} ({abs(l1 - ref) / ref:.2e})")
    cos = float((g0 * g1).sum() / (g0.norm() * g1.norm()))
    print(f"[{name}] flat gradient: |g1 - g0| / |g0| = {float((g1 - g0).norm() / g0.norm()):.3e}, cosine {cos:.5f}")
    rel0 = np.abs(n0 - z["grad_norm"]) / (z["grad_norm"] + 1e-12)
    rel1 = np.abs(n1 - z["grad_norm"]) / (z["grad_norm"] + 1e-12)
    nt = np.array([not k.endswith("tau") for k in names])
    print(f"[{name}] worst per-parameter gradient-norm deviation from the fp32 golden (tau excluded): path0 {rel0[nt].max():.3e}  path1 {rel1[nt].max():.3e}")
    assert np.isfinite(g1.cpu().numpy()).all() and (n1 > 0).all()
    assert abs(l1 - ref) <= 2e-2 * ref and abs(l1 - l0) <= 1e-2 * ref
    assert cos >= 0.97
    assert (rel1[nt] <= 0.12).all(), [(names[i], rel1[i]) for i in np.flatnonzero((rel1 > 0.12) & nt)]


def test_finetune_backbone_vs_reference_golden():
    """Next row f1 (encoder part): DynVFE + SPTBackbone without masking (all pillars are tokens, every occupancy level
    populated, dense decoder dataflow) against the reference golden (tests/golden/make_golden_finetune.py): active sets
    bit-exact, stage features / spatial_features / every parameter-gradient norm within fp32 round-off."""
    from gdmae_hip import configs
    from pcdet.models.backbones_3d import SPTBackbone
    from pcdet.models.backbones_3d.vfe import DynVFE
    z = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "finetune_kitti_b2.npz")))
    ds = configs.SyntheticDatasetInfo(**configs.KITTI)
    cfg = configs.gdmae_ssl_model_cfg(eval_metric="kitti")
    F = int(z["num_point_features"])
    vfe = DynVFE(model_cfg=cfg.VFE, num_point_features=F, voxel_size=ds.voxel_size, point_cloud_range=ds.point_cloud_range,
                 grid_size=ds.grid_size)
    bb = SPTBackbone(model_cfg=configs.gdmae_finetune_backbone_cfg(eval_metric="kitti"), input_channels=vfe.get_output_feature_dim(),
                     grid_size=ds.grid_size, voxel_size=ds.voxel_size, point_cloud_range=ds.point_cloud_range)

    class Wrap(torch.nn.Module):
        def __init__(s):
            super().__init__()
            s.vfe, s.backbone_3d = vfe, bb
    net = Wrap().to(dev())
    names = [str(n) for n in z["param_names"]]
    shapes = {n: tuple(int(v) for v in s if v > 0) for n, s in zip(names, z["param_shapes"])}
    assert {k: tuple(v.shape) for k, v in net.named_parameters()} == shapes
    net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
    net.train()
    B = int(z["batch_size"])
    bd = bb(vfe({"points": torch.from_numpy(z["points"]).to(dev()), "batch_size": B}))
    assert np.array_equal(bd["voxel_coords"].cpu().numpy(), z["voxel_coords"])
    for i in range(3):
        t = bd["multi_scale_3d_features"][f"x_conv{i + 1}"]
        assert np.array_equal(t.indices.cpu().numpy(), z[f"st{i}_indices"])
        assert_sampled_close(t.features, z[f"st{i}_features_s"], z[f"st{i}_features_c"], 5e-4, f"stage {i}")
    sf = bd["spatial_features"]
    assert_sampled_close(sf, z["spatial_features_s"], z["spatial_features_c"], 5e-4, "spatial_features")
    wgt = torch.randn(tuple(sf.shape), generator=torch.Generator().manual_seed(int(z["seed"]) + 1)).to(dev())
    ((sf * wgt).sum() / sf.numel()).backward()
    g = dict(net.named_parameters())
    gn = np.array([float(g[k].grad.double().norm()) for k in names])
    ref = z["grad_norm"]
    is_tau = np.array([k.endswith("tau") for k in names])
    # tau gradients are scalar sums with heavy cancellation (1e-8 .. 3e-6 here, one of them 100x below the others): their
    # error is set by the summation order of whichever GEMM algorithms were timed fastest, so it is measured against the
    # typical tau-gradient magnitude as well as against the value itself
    slack = np.where(is_tau, 2e-2 * np.median(ref[is_tau]), 0.0)
    tol = np.where(is_tau, 2.5e-1, 2e-2)
    bad = np.abs(gn - ref) > tol * ref + slack
    assert not bad.any(), [(names[i], gn[i], ref[i]) for i in np.flatnonzero(bad)]


def test_native_conv_block_equals_op_by_op_block():
    """gdmae_conv_block_fwd/bwd (sparse conv + BatchNorm1d + ReLU as one call per direction) vs the op-by-op path
    (ops.SparseConv3x3 + vfe.BNReLURows) in the bench configuration: same arithmetic, same kernels."""
    import logging
    from gdmae_hip import configs, optim
    from pcdet.models import build_network
    from pcdet.utils.spconv_utils import SparseSequential
    from gdmae_hip import lib as glib
    z, ds, cfg, shapes = load_case("kitti_b2_m75")
    res = {}
    # the op-by-op reference runs its im2col products through hipBLASLt: pin its algorithm choice (the heuristic's first, no timing)
    # so that the reference is the same in every process and test order
    glib.call("gdmae_gemm_tuning", 0)
    try:
        for native in (True, False):
            SparseSequential.native_block = native
            torch.manual_seed(0)
            net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev())
            net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
            net.train()
            opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)
            opt.zero_grad()
            bd = {"points": torch.from_numpy(z["points"]).to(dev()), "batch_size": int(z["batch_size"]),
                  "mae_noise": torch.from_numpy(z["noise"]).to(dev())}
            with torch.autocast("cuda", dtype=torch.bfloat16):
                ret, _, _ = net(bd)
            ret["loss"].backward()
            rs = torch.cat([v.float().reshape(-1) for k, v in net.state_dict().items() if "running_" in k])
            res[native] = (float(ret["loss"]), opt.flat_grad.clone(), rs)
    finally:
        SparseSequential.native_block = True
        glib.call("gdmae_gemm_tuning", -1)
    # With timing-based selection (the default: csrc/gemm.hip) a split-K variant that sums bf16 partials may win depending on what ran
    # before, and the REFERENCE then moves by ~2e-4 in the loss and 4 % in the gradients (seen: 16.19421 vs 16.19674); the native path is
    # deterministic (own kernels, bit-identical when repeated, insensitive to stale memory: tools/probe/garbage_probe.py)
    assert abs(res[True][0] - res[False][0]) <= 5e-4 * abs(res[False][0])
    assert abs(res[True][0] - float(z["loss"])) <= 2e-3 * float(z["loss"])
    g1, g0 = res[True][1], res[False][1]
    # gradients: with the heuristic's first algorithm (pinned above) the two bf16 paths differ by 4.1 % of the gradient norm - the
    # elementwise chaos of bf16 activations (arg-max / ReLU flips, DESIGN.md section 5: 12 % between bf16 and fp32 on these small
    # goldens); an algorithm with the native kernel's summation order used to agree to 2e-3 when timing happened to select it
    assert float((g1 - g0).norm()) <= 8e-2 * float(g0.norm()), float((g1 - g0).norm() / g0.norm())
    assert torch.allclose(res[True][2], res[False][2], rtol=2e-3, atol=1e-5)


@pytest.mark.parametrize("autocast", [False, True])
@pytest.mark.parametrize("name", ["kitti_b2", "waymo_b1", "once_e_b1"])
def test_vfe_point_layer_equals_op_by_op_layer(name, autocast):
    """gdmae_vfe_point_layer_fwd/bwd (decoration + Linear + BatchNorm1d + ReLU with the pre-activation recomputed in
    MFMA accumulators) and, in bf16 mode, gdmae_vfe_max_layer_fwd/bwd (Linear + BatchNorm1d + ReLU + pillar max, same
    idea) vs the op-by-op DynVFE layers (decorate kernel, hipBLASLt GEMM, fold + row kernels):
    pillar features, every VFE parameter gradient and the running statistics.
    fp32: 1e-4 relative (summation order).  bf16 mode: the op-by-op path rounds the decorated features (absolute
    coordinates of up to 75 m) and the pre-activation to bf16, the fused one keeps both in fp32 - so the check is
    that the fused result is at least as close to the fp32 result as the op-by-op bf16 result is."""
    import logging
    from pcdet.models import build_network
    from pcdet.models.backbones_3d.vfe.dyn_vfe import DynVFE
    z, ds, cfg, shapes = load_case(name)

    def run(fused, ac):
        DynVFE.point_layer = DynVFE.max_layer = fused
        torch.manual_seed(0)
        net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev())
        net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
        net.train()
        vfe = net.vfe
        bd = {"points": torch.from_numpy(z["points"]).to(dev()), "batch_size": int(z["batch_size"])}
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
            out = vfe(bd)["pillar_features"]
        gen = torch.Generator(device="cpu").manual_seed(1)
        up = torch.randn(out.shape, generator=gen).to(dev())
        (out.float() * up).sum().backward()
        res = {"out": out.detach().float().clone()}
        res.update({k: p.grad.detach().float().clone() for k, p in vfe.named_parameters()})
        rs = {k: v.detach().float().clone() for k, v in vfe.state_dict().items() if "running_" in k or "num_batches" in k}
        return res, rs

    def rel(a, b):
        return float((a - b).norm()) / float(b.norm())

    try:
        fused, fused_rs = run(True, autocast)
        plain, plain_rs = run(False, autocast)
        exact, exact_rs = run(False, False) if autocast else (plain, plain_rs)
    finally:
        DynVFE.point_layer = DynVFE.max_layer = True
    assert set(fused) == set(plain) and len(fused) == 7
    for k in plain:
        assert fused[k].shape == plain[k].shape and torch.isfinite(fused[k]).all() and float(plain[k].norm()) > 0, k
        if autocast:
            assert rel(fused[k], exact[k]) <= 1.25 * rel(plain[k], exact[k]) + 1e-3, (k, rel(fused[k], exact[k]),
                                                                                      rel(plain[k], exact[k]))
        else:
            assert rel(fused[k], plain[k]) <= 1e-4, (k, rel(fused[k], plain[k]))
    for k, r0 in plain_rs.items():
        if autocast:
            assert rel(fused_rs[k], exact_rs[k]) <= 1.25 * rel(r0, exact_rs[k]) + 1e-5, k
        else:
            assert torch.allclose(fused_rs[k], r0, rtol=1e-5, atol=1e-6), k


@pytest.mark.parametrize("name", ["kitti_b2", "once_e_b1"])
def test_vfe_fp16_rows_are_closer_to_fp32_than_bf16_rows(name):
    """Round 6: both fused DynVFE layers as one autograd node with FP16 rows between them (gdmae_hip.vfe.PointLayers12Max: layer 1 writes
    fp16, gdmae_vfe_max_layer_*_f16 multiply fp16 rows by the fp16 rounding of the fp32 master weights; gradients stay bf16) against the
    two-node form with bf16 rows (GDMAE_VFE_F16=0) and the op-by-op fp32 layers: the pillar features of the fp16 form are at least 3 x
    closer to fp32 (measured 6 - 8 x: 11 instead of 8 significand bits, and the pillar maximum passes one point's value on unaveraged),
    every parameter gradient at least as close, running statistics alike; 64 -> 128 (kitti_b2) and config E's 64 -> 256 as two blocks."""
    import logging
    from pcdet.models import build_network
    import pcdet.models.backbones_3d.vfe.dyn_vfe as dv
    z, ds, cfg, shapes = load_case(name)

    def run(f16, fused, ac):
        dv.VFE_F16 = f16
        dv.DynVFE.point_layer = dv.DynVFE.max_layer = fused
        torch.manual_seed(0)
        net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev())
        net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
        net.train()
        bd = {"points": torch.from_numpy(z["points"]).to(dev()), "batch_size": int(z["batch_size"])}
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=ac):
            out = net.vfe(bd)["pillar_features"]
        up = torch.randn(out.shape, generator=torch.Generator(device="cpu").manual_seed(1)).to(dev())
        (out.float() * up).sum().backward()
        res = {"out": out.detach().float().clone()}
        res.update({k: p.grad.detach().float().clone() for k, p in net.vfe.named_parameters()})
        res.update({k: v.detach().float().clone() for k, v in net.vfe.state_dict().items() if "running_" in k})
        return res

    rel = lambda a, b: float((a - b).norm()) / float(b.norm())
    try:
        h16, b16, exact = run(True, True, True), run(False, True, True), run(True, False, False)
    finally:
        dv.VFE_F16 = True
        dv.DynVFE.point_layer = dv.DynVFE.max_layer = True
    assert set(h16) == set(b16) == set(exact)
    e16, eb = rel(h16["out"], exact["out"]), rel(b16["out"], exact["out"])
    print(f"[DynVFE {name}] pillar features vs fp32: fp16 rows {e16:.2e}, bf16 rows {eb:.2e}")
    assert e16 <= eb / 3, (e16, eb)
    for k in exact:
        assert torch.isfinite(h16[k]).all() and h16[k].shape == exact[k].shape, k
        assert rel(h16[k], exact[k]) <= 1.1 * rel(b16[k], exact[k]) + 1e-4, (k, rel(h16[k], exact[k]), rel(b16[k], exact[k]))


@pytest.mark.parametrize("sizes", [[1, 700, 3, 1, 2600, 17, 2, 2, 1, 90, 31, 33, 1], [1] * 300, [3, 12000, 2, 7000, 1]])
def test_vfe_max_layer_f16_entries(sizes):
    """gdmae_vfe_max_layer_fwd_f16 / _bwd_f16 through the C ABI on crowded and tiny pillars: fp16 rows, fp32 master weights.  Forward against
    relu(BatchNorm1d_train(y1 f16(W)^T)) reduced per pillar in fp64 torch at 3e-4 (fp16 products, fp32 accumulation; the bf16 entries
    hold 2e-3), arg = the FIRST maximal row of its pillar (every second pillar is made of identical rows); backward (dy1 in bf16, dW,
    dgamma, dbeta) against autograd with the gradient routed to that first row."""
    from gdmae_hip import lib as L
    d = dev()
    gen = torch.Generator().manual_seed(len(sizes) * 11 + sizes[0])
    N, M, C = sum(sizes), len(sizes), 128
    y1 = torch.randn(N, 64, generator=gen).abs()
    off = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32)
    for p in range(1, M, 2):
        y1[off[p]:off[p + 1]] = y1[off[p]]
    y1 = y1.half()
    W = torch.randn(128, 64, generator=gen) * 0.2
    rowpil = torch.repeat_interleave(torch.arange(M, dtype=torch.int32), torch.tensor(sizes))
    gamma, beta = torch.rand(128, generator=gen) + 0.5, torch.randn(128, generator=gen) * 0.3
    up = torch.randn(M, 128, generator=gen)
    y1d, Wd, rpd, offd, gd_, bd_, upd = (t.to(d).contiguous() for t in (y1, W, rowpil, off, gamma, beta, up))
    out = torch.empty(M, C, dtype=torch.float32, device=d)
    arg = torch.full((M, C), -7, dtype=torch.int32, device=d)
    stats, ab, mv = torch.empty(2 * C, dtype=torch.float64, device=d), torch.empty(2 * C, device=d), torch.empty(2 * C, device=d)
    ws = torch.empty(L.load().gdmae_vfe_max_layer_workspace_bytes(), dtype=torch.uint8, device=d)
    L.call("gdmae_vfe_max_layer_fwd_f16", L.ptr(y1d), N, L.ptr(Wd), L.ptr(offd), L.ptr(rpd), M, L.ptr(gd_), L.ptr(bd_), 1e-3, 0.0, None, None,
           None, L.ptr(stats), L.ptr(ab), L.ptr(mv), L.ptr(out), L.ptr(arg), L.ptr(ws), L.stream())
    gm, dy1 = torch.empty_like(out), torch.empty(N, 64, dtype=torch.bfloat16, device=d)
    dg, db, dW = torch.empty(C, device=d), torch.empty(C, device=d), torch.empty(C, 64, device=d)
    L.call("gdmae_vfe_max_layer_bwd_f16", L.ptr(y1d), N, L.ptr(Wd), L.ptr(rpd), M, L.ptr(gd_), L.ptr(stats), L.ptr(ab), L.ptr(out), L.ptr(arg),
           L.ptr(upd), L.ptr(gm), L.ptr(dy1), L.ptr(dg), L.ptr(db), L.ptr(dW), 0, L.ptr(ws), L.stream())
    torch.cuda.synchronize()
    yr, Wr = y1.double().requires_grad_(), W.half().double().requires_grad_()
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    h = yr @ Wr.t()
    mean, var = h.mean(0), h.var(0, unbiased=False)
    v = torch.relu((h - mean) / torch.sqrt(var + 1e-3) * gr + br)
    ref = torch.stack([v[off[p]:off[p + 1]].max(0).values for p in range(M)])
    o = out.cpu().double()
    assert torch.allclose(o, ref.detach(), rtol=3e-4, atol=3e-4), float((o - ref).abs().max())
    a = arg.cpu().long()
    lo, hi = off[:-1].long()[:, None], off[1:].long()[:, None]
    assert bool(((a >= lo) & (a < hi)).all()), "arg-max row outside its pillar"
    for p in range(1, M, 2):
        assert bool((a[p] == int(off[p])).all()), p
    first = torch.stack([v[off[p]:off[p + 1]].max(0).indices + int(off[p]) for p in range(M)])
    (v.gather(0, first) * up.double()).sum().backward()
    for nm, got, want, tol in (("dy1", dy1, yr.grad, 3e-2), ("dW", dW, Wr.grad, 2e-2), ("dgamma", dg, gr.grad, 2e-2), ("dbeta", db, br.grad, 2e-2)):
        e = float((got.cpu().double() - want).norm()) / max(float(want.norm()), 1e-12)
        assert e <= tol, (nm, e)


@pytest.mark.parametrize("sizes", [[1, 700, 3, 1, 2600, 17, 2, 2, 1, 90, 31, 33, 1], [5000], [1] * 300, [40, 40, 40, 41, 39, 1, 500, 16, 16, 16, 16],
                                   [3, 12000, 2, 7000, 1]])
def test_vfe_max_layer_crowded_pillars(sizes):
    """gdmae_vfe_max_layer_fwd on pillars far larger than a worker's row range (k_v2_max gives every half-wave the same number of
    rows and k_v2_max_fix joins the pieces of a pillar that crosses range boundaries - a 1 000-point pillar next to the sensor is
    the normal case at 0.32 m): out against relu(BatchNorm1d_train(y1 W^T)) reduced per pillar in fp32 torch (2e-3: bf16 products,
    fp32 accumulation), arg a row of ITS pillar whose value reaches the maximum, and - every second pillar made of identical rows -
    the FIRST row of the pillar on ties, bit for bit (torch_scatter's scatter_max semantics as dyn_vfe.py:112 uses it)."""
    from gdmae_hip import vfe as gvfe
    d = dev()
    gen = torch.Generator().manual_seed(len(sizes) * 7 + sizes[0])
    N, M = sum(sizes), len(sizes)
    y1 = torch.randn(N, 64, generator=gen).abs()
    off = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32)
    for p in range(1, M, 2):                                   # identical rows: every row of the pillar ties in every column
        y1[off[p]:off[p + 1]] = y1[off[p]]
    y1 = y1.bfloat16()
    W = (torch.randn(128, 64, generator=gen) * 0.2)
    rowpil = torch.repeat_interleave(torch.arange(M, dtype=torch.int32), torch.tensor(sizes))
    gamma, beta = torch.rand(128, generator=gen) + 0.5, torch.randn(128, generator=gen) * 0.3
    y1d_, Wd_, gd__, bd__ = y1.to(d).requires_grad_(), W.to(d).requires_grad_(), gamma.to(d).requires_grad_(), beta.to(d).requires_grad_()
    out, mf, vf = gvfe.PointLayer2Max.apply(y1d_, rowpil.to(d), Wd_, gd__, bd__, 1e-3, off.to(d), None)
    up = torch.randn(M, 128, generator=gen)
    (out * up.to(d)).sum().backward()
    torch.cuda.synchronize()
    out = out.detach()
    yr, Wr = y1.double().requires_grad_(), W.bfloat16().double().requires_grad_()
    gr, br = gamma.double().requires_grad_(), beta.double().requires_grad_()
    h = yr @ Wr.t()
    mean, var = h.mean(0), h.var(0, unbiased=False)
    v = torch.relu((h - mean) / torch.sqrt(var + 1e-3) * gr + br)
    ref = torch.stack([v[off[p]:off[p + 1]].max(0).values for p in range(M)])
    o = out.cpu().double()
    assert torch.allclose(o, ref.detach(), rtol=2e-3, atol=2e-3), float((o - ref).abs().max())
    # backward (k_v2_gstats / k_v2_dy / k_v2_dw: per-pillar (arg, gm) chunks scattered to the arg-max rows through the LDS window;
    # [1] * 300 puts 32 pillars into a tile = four chunks): the reference routes the gradient to the FIRST maximal row as well
    first = torch.stack([v[off[p]:off[p + 1]].max(0).indices + int(off[p]) for p in range(M)])
    (v.gather(0, first) * up.double()).sum().backward()
    for name, got, want, tol in (("dy1", y1d_.grad, yr.grad, 3e-2), ("dW", Wd_.grad, Wr.grad, 2e-2), ("dgamma", gd__.grad, gr.grad, 2e-2),
                                 ("dbeta", bd__.grad, br.grad, 2e-2)):
        e = float((got.cpu().double() - want).norm()) / max(float(want.norm()), 1e-12)
        assert e <= tol, (name, e)
    v, ref = v.detach(), ref.detach()
    # the arg-max rows are an internal output: re-run the C entry to read them
    from gdmae_hip import lib as L
    C = 128
    wb = W.bfloat16().to(d).contiguous()
    o2 = torch.empty(M, C, dtype=torch.float32, device=d)
    arg = torch.full((M, C), -7, dtype=torch.int32, device=d)
    stats, ab, mv = torch.empty(2 * C, dtype=torch.float64, device=d), torch.empty(2 * C, device=d), torch.empty(2 * C, device=d)
    ws = torch.empty(L.load().gdmae_vfe_max_layer_workspace_bytes(), dtype=torch.uint8, device=d)
    y1d, rpd, offd, gd_, bd_ = y1.to(d), rowpil.to(d), off.to(d), gamma.to(d), beta.to(d)
    L.call("gdmae_vfe_max_layer_fwd", L.ptr(y1d), N, L.ptr(wb), L.ptr(offd), L.ptr(rpd), M, L.ptr(gd_), L.ptr(bd_), 1e-3, 0.0, None,
           None, None, L.ptr(stats), L.ptr(ab), L.ptr(mv), L.ptr(o2), L.ptr(arg), L.ptr(ws), L.stream())
    torch.cuda.synchronize()
    assert torch.equal(o2, out)
    a = arg.cpu().long()
    lo, hi = off[:-1].long()[:, None], off[1:].long()[:, None]
    assert bool(((a >= lo) & (a < hi)).all()), "arg-max row outside its pillar"
    assert torch.allclose(v.gather(0, a), ref, rtol=2e-3, atol=2e-3)
    for p in range(1, M, 2):
        assert bool((a[p] == int(off[p])).all()), (p, sizes[p], a[p].unique())


def _random_sources(B, H, W, dens, seed, operand_dtype=torch.float16):
    """Random token sets of three source stages (strides 1, 2, 4) with their dense cell -> token maps, deconvolution
    rows P (bf16) and folded BatchNorm affines; plus the dense input map the reference dataflow would build, rounded to the
    operand type of the tile convolution (fp16 since round 6)."""
    g = torch.Generator().manual_seed(seed)
    maps, Ps, a_l, b_l, ups = [], [], [], [], [1, 2, 4]
    Z = torch.empty(B, H, W, 384)
    for i, s in enumerate(ups):
        Hs, Ws = H // s, W // s
        act = torch.rand(B * Hs * Ws, generator=g) < dens[i]
        cells = torch.nonzero(act).flatten()
        m = torch.full((B * Hs * Ws,), -1, dtype=torch.int32)
        m[cells] = torch.arange(cells.numel(), dtype=torch.int32)
        P = (torch.randn(cells.numel() * s * s, 128, generator=g) * 1.5).bfloat16()
        a = torch.rand(128, generator=g) + 0.5
        b = torch.randn(128, generator=g) * 0.5
        zg = torch.relu(b).to(operand_dtype).float().expand(B, H, W, 128).clone()
        # one rounding like the kernel's fmaf (a * P + b in fp64, then to fp32): an fp32 ulp decides one in 2^13 fp16 roundings
        rows = torch.relu((P.double() * a.double() + b.double()).float()).to(operand_dtype).float().view(cells.numel(), s, s, 128)
        bb = cells // (Hs * Ws)
        yy = (cells // Ws) % Hs
        xx = cells % Ws
        for dy in range(s):
            for dx in range(s):
                zg[bb, yy * s + dy, xx * s + dx] = rows[:, dy, dx]
        Z[..., 128 * i:128 * (i + 1)] = zg
        maps.append(m), Ps.append(P), a_l.append(a), b_l.append(b)
    return maps, Ps, a_l, b_l, ups, Z


@pytest.mark.parametrize("B,H,W,dens", [(2, 40, 40, (0.02, 0.03, 0.03)), (3, 44, 36, (0.01, 0.0, 0.02)), (1, 16, 24, (0.0, 0.0, 0.0)),
                                        (2, 24, 32, (0.3, 0.3, 0.5))])
def test_conv3x3_tiles_matches_dense_conv(B, H, W, dens):
    """The 16-bit-MFMA 3x3 convolution over the active tiles (gdmae_decoder_tiles + gdmae_conv3x3_tiles_pack/_fwd) against
    F.conv2d in fp64 on the SAME fp16-rounded operands (the kernel multiplies fp16 operands and rounds its output to bf16): active-tile set vs a brute-force dilation, every site of the map
    (active tiles and border-class constants) within bf16 output rounding, fused BatchNorm statistics, and the two readers
    (gather at cells, dense expansion).  Edge cases: maps that are not multiples of the tile, an empty stage, no active
    site at all, nearly full maps."""
    import torch.nn.functional as F
    from gdmae_hip import lib as L
    lib = L.load()
    d = dev()
    maps, Ps, a_l, b_l, ups, Z = _random_sources(B, H, W, dens, seed=B * 1000 + H)
    gen = torch.Generator().manual_seed(7)
    w = torch.randn(128, 384, 3, 3, generator=gen) * 0.05
    gamma, beta = torch.rand(128, generator=gen) + 0.5, torch.randn(128, generator=gen)
    TH, TW = (H + 7) // 8, (W + 7) // 8
    # ---- active tiles
    mapsd = [m.to(d) for m in maps]
    slot = torch.empty(B * TH * TW, dtype=torch.int32, device=d)
    tlist = torch.empty(B * TH * TW, dtype=torch.int32, device=d)
    nact = torch.zeros(1, dtype=torch.int32, device=d)
    ws = torch.empty(lib.gdmae_decoder_tiles_workspace_bytes(B, H, W), dtype=torch.uint8, device=d)
    L.call("gdmae_decoder_tiles", L.host_ptrs(mapsd), L.host_i32(ups), 3, B, H, W, L.ptr(slot), L.ptr(tlist), L.ptr(nact), L.ptr(ws),
           L.stream())
    n_act = int(nact.item())
    act = torch.zeros(B, H, W, dtype=torch.bool)
    for m, s in zip(maps, ups):
        act |= (m.view(B, H // s, W // s) >= 0).repeat_interleave(s, 1).repeat_interleave(s, 2)
    dil = F.max_pool2d(act.float()[:, None], 3, 1, 1)[:, 0] > 0
    pad = torch.zeros(B, TH * 8, TW * 8, dtype=torch.bool)
    pad[:, :H, :W] = dil
    tile_ref = pad.view(B, TH, 8, TW, 8).any(4).any(2).flatten()
    assert torch.equal(slot.cpu() >= 0, tile_ref)
    assert n_act == int(tile_ref.sum())
    assert torch.equal(tlist[:n_act].cpu().long(), torch.nonzero(tile_ref).flatten())
    assert torch.equal(slot.cpu()[tile_ref].long(), torch.arange(n_act))
    # ---- convolution + statistics
    wd = w.to(d).contiguous()
    Psd, ad, bd_ = [P.to(d) for P in Ps], [a.to(d) for a in a_l], [b.to(d) for b in b_l]
    Wp = torch.empty(lib.gdmae_conv3x3_tiles_packed_bytes(3), dtype=torch.uint8, device=d)
    bgz = torch.empty(384, dtype=torch.bfloat16, device=d)
    ybg = torch.empty(9, 128, dtype=torch.bfloat16, device=d)
    L.call("gdmae_conv3x3_tiles_pack", L.ptr(wd), 128, 384, L.host_ptrs(bd_), 3, L.ptr(Wp), L.ptr(bgz), L.ptr(ybg), L.stream())
    assert torch.equal(bgz.cpu().float(), torch.cat([torch.relu(b).bfloat16().float() for b in b_l]))
    Yc = torch.full((max(n_act, 1) * 64, 128), float("nan"), dtype=torch.bfloat16, device=d)
    stats = torch.empty(256, dtype=torch.float64, device=d)
    ab = torch.empty(256, dtype=torch.float32, device=d)
    mv = torch.empty(256, dtype=torch.float32, device=d)
    rm, rv, nb = torch.zeros(128, device=d), torch.ones(128, device=d), torch.zeros(1, dtype=torch.int64, device=d)
    ws2 = torch.empty(lib.gdmae_conv3x3_tiles_workspace_bytes(n_act), dtype=torch.uint8, device=d)
    gd, bed = gamma.to(d), beta.to(d)
    L.call("gdmae_conv3x3_tiles_fwd", L.host_ptrs(Psd), L.host_ptrs(mapsd), L.host_ptrs(ad), L.host_ptrs(bd_), L.host_i32(ups), 3,
           L.ptr(Wp), L.ptr(ybg), L.ptr(tlist), n_act, B, H, W, L.ptr(Yc), L.ptr(gd), L.ptr(bed), 1e-3, 0.01, L.ptr(rm), L.ptr(rv),
           L.ptr(nb), L.ptr(stats), L.ptr(ab), L.ptr(mv), L.ptr(ws2), L.stream())
    yd = torch.empty(B * H * W, 128, dtype=torch.bfloat16, device=d)
    L.call("gdmae_tiles_to_dense", L.ptr(Yc), L.ptr(slot), L.ptr(ybg), B, H, W, 128, 2, L.ptr(yd), L.stream())
    ref = F.conv2d(Z.permute(0, 3, 1, 2).double(), w.half().double(), None, 1, 1).permute(0, 2, 3, 1).reshape(B * H * W, 128)
    got = yd.cpu().double()
    scale = ref.abs().max()
    # bf16 output rounding (2^-9 relative) + fp32 accumulation order over K = 3456
    err = (got - ref).abs()
    assert float((err / (ref.abs() + 1e-3 * scale)).max()) < 6e-3, float((err / (ref.abs() + 1e-3 * scale)).max())
    # statistics of the ROUNDED map over all sites
    mean, var = got.mean(0), got.var(0, unbiased=False)
    assert torch.allclose(mv[:128].cpu().double(), mean, rtol=1e-5, atol=1e-5 * float(scale))
    assert torch.allclose(mv[128:].cpu().double(), var, rtol=1e-4, atol=1e-6 * float(scale) ** 2)
    a_ref = gamma.double() / torch.sqrt(var + 1e-3)
    assert torch.allclose(ab[:128].cpu().double(), a_ref, rtol=1e-4)
    assert torch.allclose(ab[128:].cpu().double(), beta.double() - a_ref * mean, rtol=1e-4, atol=1e-4)
    n = B * H * W
    assert torch.allclose(rm.cpu().double(), 0.01 * mean, rtol=1e-4, atol=1e-6)
    assert torch.allclose(rv.cpu().double(), 0.99 + 0.01 * var * n / (n - 1), rtol=1e-4)
    assert int(nb.item()) == 1
    # ---- gather at cells (through the tile map, class constants outside the active tiles)
    cells = torch.randperm(n, generator=gen)[:min(n, 500)].int().sort().values.to(d)
    rows = torch.empty(cells.numel(), 128, dtype=torch.bfloat16, device=d)
    L.call("gdmae_tiles_gather_rows", L.ptr(Yc), L.ptr(slot), L.ptr(ybg), L.ptr(cells), cells.numel(), H, W, 128, 2, L.ptr(rows), L.stream())
    assert torch.equal(rows, yd[cells.long()])


@pytest.mark.parametrize("name", ["kitti_b2_m75", "waymo_b1"])
def test_tile_conv_decoder_equals_dense_conv_decoder_bf16(name):
    """A/B in throughput mode (bf16 autocast) on the same weights: decoder conv_out as the library's tile convolution vs the
    materialised map + F.conv2d (both consume the same bf16-rounded operands; they differ by accumulation order and by the
    closed-form treatment of the constant regions): loss, pillar rows, dense spatial_features, running statistics, and
    every gradient.  bf16 activations make gradients chaotic at the 10 % level elementwise (arg-max / ReLU flips), so the
    gradient bound is relative to that noise: the tile path must be as close to the fp32 gradients as the dense-conv
    bf16 path is (measured: |tiles - dense| 4 % median / 7 % max, |bf16 - fp32| 12 % median / 23 % max of the norm)."""
    import logging
    from pcdet.models import build_network
    z, ds, cfg, shapes = load_case(name)
    res = {}
    for impl in ("tiles", "dense", "fp32"):
        torch.manual_seed(0)
        net = build_network(cfg, 3, ds, logging.getLogger("t")).to(dev())
        net.load_state_dict(orc.seeded_state_dict(shapes, seed=5), strict=False)
        net.backbone_3d.decoder_conv_impl = "dense" if impl == "fp32" else impl
        net.train()
        bd = {"points": torch.from_numpy(z["points"]).to(dev()), "batch_size": int(z["batch_size"]),
              "mae_noise": torch.from_numpy(z["noise"]).to(dev())}
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=impl != "fp32"):
            ret, _, _ = net(bd)
        ret["loss"].backward()
        res[impl] = (float(ret["loss"].detach()), bd["voxel_features"].detach().float().cpu(),
                     bd["spatial_features"].detach().float().cpu(),
                     {k: p.grad.detach().double().cpu() for k, p in net.named_parameters()},
                     {k: v.detach().cpu().clone() for k, v in net.state_dict().items() if "running" in k or "num_batches" in k})
    lt, vt, sft, gt, rt = res["tiles"]
    ld, vd, sfd, gd, rd = res["dense"]
    gf = res["fp32"][3]
    assert abs(lt - ld) <= 2e-3 * abs(ld), (lt, ld)
    assert (vt - vd).abs().max() <= 2e-2 * vd.abs().max()
    assert (sft - sfd).abs().max() <= 2e-2 * sfd.abs().max()
    worst_td, worst_excess = ("", 0.0), ("", -1.0)
    for k in gt:
        if k.endswith("tau"):
            continue            # noise-dominated in bf16 mode
        n = float(gf[k].norm()) + 1e-30
        d_td, d_tf, d_df = (float((a - b).norm()) / n for a, b in ((gt[k], gd[k]), (gt[k], gf[k]), (gd[k], gf[k])))
        worst_td = max(worst_td, (k, d_td), key=lambda t: t[1])
        worst_excess = max(worst_excess, (k, d_tf - 1.3 * d_df), key=lambda t: t[1])
    print("tile-vs-dense gradient distance: worst %s %.4f; worst excess over 1.3 x dense-vs-fp32: %s %.4f" % (*worst_td, *worst_excess))
    # both runs are deterministic, but any change of a bf16 kernel's rounding re-rolls the chaos: observed excess over versions of
    # the attention kernels -0.001 ... +0.03
    assert worst_td[1] <= 0.15 and worst_excess[1] <= 0.05, (worst_td, worst_excess)
    for k in rt:
        assert torch.allclose(rt[k].float(), rd[k].float(), rtol=2e-3, atol=1e-5), k


@pytest.mark.parametrize("rows,M,N", [(2048, 128, 128), (4096, 256, 128), (6144, 256, 512), (2048, 512, 256)])
def test_dw_gemm_matches_fp32_product(rows, M, N):
    """Hand-written TN product of the weight gradients (dw_grouped.hip): dW = G^T X and db = column sums of G for bf16 rows
    against the fp32 product of the same bf16 values (fp32 accumulation in a different order: 1e-5 of the largest entry)."""
    from gdmae_hip import lib as L
    g = torch.Generator().manual_seed(rows + M + N)
    G = (torch.randn(rows, M, generator=g) * 0.5).to(torch.bfloat16).to(dev())
    X = torch.randn(rows, N, generator=g).to(torch.bfloat16).to(dev())
    G[rows - 300:] = 0                                   # zero-padded tail rows, as the layer executor passes them
    dW = torch.empty(M, N, dtype=torch.float32, device=dev())
    db = torch.empty(M, dtype=torch.float32, device=dev())
    ws = torch.empty(L.load().gdmae_dw_gemm_workspace_bytes(rows, M, N), dtype=torch.uint8, device=dev())
    L.call("gdmae_dw_gemm", L.ptr(G), L.ptr(X), rows, M, N, L.ptr(dW), L.ptr(db), L.ptr(ws), L.stream())
    ref = G.double().t() @ X.double()
    assert (dW.double() - ref).abs().max() <= 1e-5 * ref.abs().max()
    refb = G.double().sum(0)
    assert (db.double() - refb).abs().max() <= 1e-5 * refb.abs().max() + 1e-6
    dW2 = torch.empty_like(dW)
    L.call("gdmae_dw_gemm", L.ptr(G), L.ptr(X), rows, M, N, L.ptr(dW2), None, L.ptr(ws), L.stream())
    assert torch.equal(dW, dW2), "deterministic"


def _pack_weight(w, transpose=False):
    """Packed (fragment-ordered) image of a 2-D fp32 weight through gdmae_tok_gemm_pack."""
    from gdmae_hip import lib as L
    M, K = (w.shape[1], w.shape[0]) if transpose else w.shape
    dst = torch.empty(M * K, dtype=torch.bfloat16, device=w.device)
    jobs = torch.tensor([w.data_ptr(), dst.data_ptr(), M, K, w.shape[1], int(transpose)], dtype=torch.int64).to(w.device)
    L.call("gdmae_tok_gemm_pack", L.ptr(jobs), 1, L.stream())
    torch.cuda.synchronize()
    return dst


@pytest.mark.parametrize("d,with_bias", [(128, True), (256, True), (256, False)])
def test_tok_gemm_qkv_is_bit_identical_to_separate_products(d, with_bias):
    """gdmae_tok_gemm_qkv (q, k and v projections of a layer as three jobs of one launch, the k half addressed inside the
    packed (2d, d) image) against two gdmae_tok_gemm calls with the plain epilogue: same k order per output element ->
    identical bits; row counts that are / are not a multiple of 8 tiles (the launch pads its grid to whole XCD groups)."""
    from gdmae_hip import lib as L
    d_ = dev()
    g = torch.Generator().manual_seed(31 * d + int(with_bias))
    for n_pad in (64, 1024, 1984):
        X = torch.randn(n_pad, d, generator=g).bfloat16().to(d_)
        Xp = (X.float().cpu() + torch.randn(n_pad, d, generator=g)).bfloat16().to(d_)
        Win = (torch.randn(3 * d, d, generator=g) / d ** 0.5).to(d_)
        b3 = torch.randn(3 * d, generator=g).bfloat16().to(d_) if with_bias else None
        Wp_qk, Wp_v = _pack_weight(Win[:2 * d].contiguous()), _pack_weight(Win[2 * d:].contiguous())
        qk, v = (torch.full((n_pad, 2 * d), 7.0, dtype=torch.bfloat16, device=d_), torch.full((n_pad, d), 7.0, dtype=torch.bfloat16, device=d_))
        L.call("gdmae_tok_gemm_qkv", L.ptr(Xp), L.ptr(X), L.ptr(Wp_qk), L.ptr(Wp_v), L.ptr(b3), n_pad, d, L.ptr(qk), L.ptr(v), L.stream())
        qk_ref, v_ref = torch.empty_like(qk), torch.empty_like(v)
        for Xi, Wp, b, N, out in ((Xp, Wp_qk, b3[:2 * d] if with_bias else None, 2 * d, qk_ref), (X, Wp_v, b3[2 * d:] if with_bias else None, d, v_ref)):
            L.call("gdmae_tok_gemm", L.ptr(Xi), L.ptr(Wp), L.ptr(b), n_pad, n_pad, d, N, 0, L.ptr(out), None, None, None, None, None, 1e-5,
                   None, None, None, None, None, None, L.stream())
        torch.cuda.synchronize()
        assert torch.equal(qk.view(torch.int16), qk_ref.view(torch.int16)), (d, n_pad)
        assert torch.equal(v.view(torch.int16), v_ref.view(torch.int16)), (d, n_pad)
        ref = Xp.float() @ Win[:2 * d].bfloat16().float().t() + (b3[:2 * d].float() if with_bias else 0.0)
        assert float((qk.float() - ref).abs().max()) < 0.05 * float(ref.abs().max())


@pytest.mark.parametrize("K,N", [(128, 128), (128, 256), (256, 128), (256, 256), (256, 512), (512, 256)])
def test_tok_gemm_epilogues_match_torch(K, N):
    """gdmae_tok_gemm (bf16 MFMA token GEMM with fused row epilogues) against torch on the same bf16-rounded operands:
    plain (+bias), bias + GELU (h and gelu(h)), GELU backward, residual + LayerNorm (+ bf16 copies, + positional copy).
    The product is accumulated in fp32 in a different order than the reference GEMM: results agree to bf16 rounding of
    the output (2^-8 relative) for the bf16 outputs and to 2e-3 of the row scale for the LayerNorm output."""
    from gdmae_hip import lib as L
    d_ = dev()
    g = torch.Generator().manual_seed(K * 7 + N)
    n, n_pad = 1000, 1024
    X = torch.zeros(n_pad, K)
    X[:n] = torch.randn(n, K, generator=g)
    Xb = X.bfloat16().to(d_)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(d_)
    bias = torch.randn(N, generator=g).bfloat16().to(d_)
    Wp = _pack_weight(W)
    ref = Xb.float() @ W.bfloat16().float().t() + bias.float()            # fp32 accumulate of bf16 operands

    def call(epi, out0=None, out1=None, aux=None, res=None, gamma=None, beta=None, y=None, stats=None, ybf=None, pos=None, tp=None,
             ypos=None, b=bias):
        L.call("gdmae_tok_gemm", L.ptr(Xb), L.ptr(Wp), L.ptr(b), n, n_pad, K, N, epi, L.ptr(out0), L.ptr(out1), L.ptr(aux), L.ptr(res),
               L.ptr(gamma), L.ptr(beta), 1e-5, L.ptr(y), L.ptr(stats), L.ptr(ybf), L.ptr(pos), L.ptr(tp), L.ptr(ypos), L.stream())

    def close_bf(a, b_, what):
        err = (a.float() - b_).abs() / (b_.abs() + 0.05 * b_.abs().max())
        assert float(err.max()) < 1.2e-2, (what, float(err.max()))

    out = torch.empty(n_pad, N, dtype=torch.bfloat16, device=d_)
    call(0, out0=out)
    close_bf(out, ref, "plain")
    # transposed packing: the same product from W^T stored (K, N)
    Wt = W.t().contiguous()
    Wp_t = _pack_weight(Wt, transpose=True)
    out_t = torch.empty_like(out)
    L.call("gdmae_tok_gemm", L.ptr(Xb), L.ptr(Wp_t), L.ptr(bias), n, n_pad, K, N, 0, L.ptr(out_t), None, None, None, None, None, 1e-5,
           None, None, None, None, None, None, L.stream())
    assert torch.equal(out_t, out)
    # bias + GELU
    h, ga = torch.empty_like(out), torch.empty_like(out)
    call(1, out0=h, out1=ga)
    assert torch.equal(h, out)
    # GELU of the ROUNDED h, as the unfused path; erf to 1.5e-7 absolute (Abramowitz-Stegun 7.1.26), far below the bf16
    # rounding of the result: at most one bf16 ulp apart from the exact-erf GELU, and only on rounding boundaries
    gr = torch.nn.functional.gelu(h.float())
    dg = (ga.float() - gr).abs()
    assert float((dg / (gr.abs() + 1e-3)).max()) < 2 ** -7 and float((ga != gr.bfloat16()).float().mean()) < 1e-2
    # GELU backward: dh = (X W^T) * gelu'(aux)
    aux = torch.randn(n_pad, N, generator=g).bfloat16().to(d_)
    dh = torch.empty_like(out)
    call(2, out0=dh, aux=aux, b=None)
    prod = (Xb.float() @ W.bfloat16().float().t())
    a = aux.float()
    gp = 0.5 * (1 + torch.erf(a * 0.7071067811865476)) + a * torch.exp(-0.5 * a * a) * 0.3989422804014327
    close_bf(dh, prod.bfloat16().float() * gp, "gelu_bwd")
    if N <= 256:
        res = torch.randn(n, N, generator=g).to(d_)
        gamma, beta = (torch.rand(N, generator=g) + 0.5).to(d_), torch.randn(N, generator=g).to(d_)
        pos = torch.randn(64, N, generator=g).to(d_)
        tp = torch.randint(0, 64, (n,), generator=g).int().to(d_)
        y = torch.empty(n, N, device=d_)
        stats = torch.empty(n, 2, device=d_)
        ybf = torch.zeros(n_pad, N, dtype=torch.bfloat16, device=d_)
        ypos = torch.zeros(n_pad, N, dtype=torch.bfloat16, device=d_)
        f = torch.empty(n_pad, N, dtype=torch.bfloat16, device=d_)
        call(3, out0=f, res=res, gamma=gamma, beta=beta, y=y, stats=stats, ybf=ybf, pos=pos, tp=tp, ypos=ypos)
        assert torch.equal(f, out)
        s = res + f[:n].float()
        yr = torch.nn.functional.layer_norm(s, (N,), gamma, beta, 1e-5)
        assert float((y - yr).abs().max()) < 2e-5 * float(yr.abs().max()) + 1e-5     # LayerNorm of the SAME rounded branch
        assert torch.allclose(stats[:, 0], s.mean(1), atol=1e-5) and torch.allclose(stats[:, 1], torch.rsqrt(s.var(1, unbiased=False) + 1e-5), rtol=1e-4)
        assert torch.equal(ybf[:n], y.bfloat16()) and float(ybf[n:].abs().max()) == 0
        assert torch.equal(ypos[:n], (y + pos[tp.long()]).bfloat16())
        # LayerNorm backward fused behind the product: against the row kernel (gdmae_add_layernorm_bwd) fed with the rounded
        # product as its 2nd / 3rd gradient piece - same arithmetic per row, so dx is bit-identical; the parameter-gradient
        # partials are summed over different row groups
        for with_dy2 in (False, True):
            dy = torch.randn(n, N, generator=g).to(d_)
            dy2 = torch.randn(n_pad, N, generator=g).bfloat16().to(d_) if with_dy2 else None
            la = torch.randn(n, N, generator=g).to(d_)
            lb = torch.randn(n_pad, N, generator=g).bfloat16().to(d_)
            ssum = la + lb[:n].float()
            st = torch.stack([ssum.mean(1), torch.rsqrt(ssum.var(1, unbiased=False) + 1e-5)], 1).contiguous()
            rows = L.load().gdmae_tok_gemm_ln_bwd_rows(N)
            part = torch.empty(n_pad // rows, 3, N, device=d_)
            dx, dxb = torch.empty(n, N, device=d_), torch.zeros(n_pad, N, dtype=torch.bfloat16, device=d_)
            L.call("gdmae_tok_gemm_ln_bwd", L.ptr(Xb), L.ptr(Wp), n, n_pad, K, N, L.ptr(dy), L.ptr(dy2), L.ptr(la), L.ptr(lb), L.ptr(st),
                   L.ptr(gamma), L.ptr(dx), L.ptr(dxb), L.ptr(part), L.stream())
            prod_b = torch.empty(n_pad, N, dtype=torch.bfloat16, device=d_)
            call(0, out0=prod_b, b=None)
            dx_r, sums = torch.empty(n, N, device=d_), torch.empty(3, N, device=d_)
            ws = torch.empty(L.load().gdmae_add_layernorm_workspace_bytes(N), dtype=torch.uint8, device=d_)
            if with_dy2:
                dsum = torch.empty(n, N, device=d_)      # the row kernel's C ABI takes two pieces: fold dy + dy2 like the fused path
                dsum.copy_(dy + dy2[:n].float())
                L.call("gdmae_add_layernorm_bwd", L.ptr(la), L.ptr(lb), 1, L.ptr(gamma), L.ptr(st), L.ptr(dsum), L.ptr(prod_b), 1, n, N,
                       L.ptr(dx_r), None, L.ptr(sums), L.ptr(ws), L.stream())
            else:
                L.call("gdmae_add_layernorm_bwd", L.ptr(la), L.ptr(lb), 1, L.ptr(gamma), L.ptr(st), L.ptr(dy), L.ptr(prod_b), 1, n, N,
                       L.ptr(dx_r), None, L.ptr(sums), L.ptr(ws), L.stream())
            assert torch.equal(dx, dx_r), float((dx - dx_r).abs().max())
            assert torch.equal(dxb[:n], dx.bfloat16())
            ps = part.double().sum(0)
            assert float((ps - sums.double()).abs().max()) <= 1e-5 * float(sums.abs().max()) + 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_shortcut", [False, True])
def test_dense_bn_relu_through_row_kernels_matches_torch(dtype, with_shortcut):
    """gdmae_hip.dense.conv_bn_relu (fine-tune row f1: BatchNorm2d(train) + ReLU [+ identity shortcut] of a dense channels-last map
    through gdmae_bn_fold / gdmae_rows_affine_relu[_add] / gdmae_rows_bwd_stats / gdmae_rows_bwd) against the module sequence it
    replaces: output, running statistics, and the gradients of the input, the shortcut, gamma and beta."""
    import copy
    import torch.nn as nn
    from gdmae_hip import dense as gdense
    torch.manual_seed(3)
    B, C, Y, X = 2, 64, 37, 29
    block = nn.Sequential(nn.Conv2d(C, C, 3, padding=1, bias=False), nn.BatchNorm2d(C, eps=1e-3, momentum=0.01), nn.ReLU()).to(dev())
    with torch.no_grad():
        block[1].weight.uniform_(0.5, 1.5)
        block[1].bias.normal_()
    ref = copy.deepcopy(block)
    x0 = torch.randn(B, C, Y, X, device=dev()).to(memory_format=torch.channels_last)
    g0 = torch.randn(B, C, Y, X, device=dev()).to(memory_format=torch.channels_last)
    outs = []
    for mod, fused in ((block, True), (ref, False)):
        x = x0.clone().requires_grad_()
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            if fused:
                y = gdense.conv_bn_relu(mod, x, shortcut=x if with_shortcut else None)
                rm, rv = mod[1].running_mean, mod[1].running_var
            else:
                # the module sequence written out (training-mode BatchNorm2d in fp32 on the convolution's output, rounded back to its
                # dtype; MIOpen's own NHWC batch norm is not used as the reference: it crashes on this map size in bf16)
                bn = mod[1]
                c = mod[0](x)
                cf = c.float()
                mean, var = cf.mean((0, 2, 3)), cf.var((0, 2, 3), unbiased=False)
                z = (cf - mean[None, :, None, None]) * torch.rsqrt(var + bn.eps)[None, :, None, None]
                z = (z * bn.weight[None, :, None, None] + bn.bias[None, :, None, None]).to(c.dtype)
                y = torch.relu(z)
                y = y + x.to(y.dtype) if with_shortcut else y
                cnt = cf.numel() / C
                rm = bn.momentum * mean.detach()
                rv = (1 - bn.momentum) * torch.ones_like(var) + bn.momentum * var.detach() * cnt / (cnt - 1)
        (y.float() * g0).sum().backward()
        outs.append((y.float(), x.grad.float(), mod[1].weight.grad, mod[1].bias.grad, mod[0].weight.grad, rm, rv))
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    names = ["out", "dx", "dgamma", "dbeta", "dW", "running_mean", "running_var"]
    for nm, a, b in zip(names, outs[0], outs[1]):
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= tol * scale, (nm, float((a - b).abs().max()), scale)
    assert int(block[1].num_batches_tracked) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_dense_bn_relu_cat_equals_separate_blocks(dtype):
    """gdmae_hip.dense.conv_bn_relu_cat (the decoder's three deblocks writing BatchNorm + ReLU straight into the column slices of the
    concatenated channels-last map) against torch.cat of the single-block path: output, running statistics and every gradient."""
    import copy
    import torch.nn as nn
    from gdmae_hip import dense as gdense
    torch.manual_seed(5)
    B, Y, X = 2, 24, 40
    spec = [(64, 128, 1), (128, 128, 2), (256, 128, 4)]
    blocks = [nn.Sequential(nn.ConvTranspose2d(ci, co, s, stride=s, bias=False), nn.BatchNorm2d(co, eps=1e-3, momentum=0.01), nn.ReLU())
              .to(dev()) for ci, co, s in spec]
    with torch.no_grad():
        for b in blocks:
            b[1].weight.uniform_(0.5, 1.5)
            b[1].bias.normal_()
    ref = copy.deepcopy(blocks)
    xs0 = [torch.randn(B, ci, Y // s, X // s, device=dev()).to(memory_format=torch.channels_last) for ci, _, s in spec]
    g0 = torch.randn(B, 384, Y, X, device=dev()).to(memory_format=torch.channels_last)
    res = []
    for mods, fused in ((blocks, True), (ref, False)):
        xs = [x.clone().requires_grad_() for x in xs0]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            y = gdense.conv_bn_relu_cat(mods, xs) if fused else torch.cat([gdense.conv_bn_relu(m, x) for m, x in zip(mods, xs)], dim=1)
        assert y.shape == (B, 384, Y, X)
        (y.float() * g0).sum().backward()
        res.append([y.float()] + [x.grad for x in xs] + [m[1].weight.grad for m in mods] + [m[1].bias.grad for m in mods] +
                   [m[0].weight.grad for m in mods] + [m[1].running_mean for m in mods] + [m[1].running_var for m in mods])
    names = ["out"] + [f"{k}{i}" for k in ("dx", "dgamma", "dbeta", "dW", "running_mean", "running_var") for i in range(3)]
    for nm, a, b in zip(names, *res):         # (the library's transposed convolutions are not bit-reproducible call to call)
        tol = (1e-5 if dtype == torch.float32 else 2e-2) * (float(b.float().abs().max()) + 1e-6)
        assert float((a.float() - b.float()).abs().max()) <= tol, (nm, float((a.float() - b.float()).abs().max()), tol)


@pytest.mark.parametrize("d", [128, 256])
def test_one_launch_ffn_equals_two_token_gemms(d):
    """gdmae_tok_gemm_ffn (linear1 + GELU + linear2 + residual + LayerNorm 2 in one launch, gelu(h) in LDS only) against the two
    launches it replaces (gdmae_tok_gemm epilogue 1, then epilogue 3): every output bit for bit, ragged row count, with and
    without the optional copies; the GELU-backward epilogue re-creates gelu(h) bit for bit."""
    from gdmae_hip import lib as L
    g = torch.Generator().manual_seed(d)
    ff, n, n_pad = 2 * d, 1237, 1280
    d_ = dev()

    def pack(w):
        M, K = w.shape
        dst = torch.empty(M * K, dtype=torch.bfloat16, device=d_)
        jobs = torch.tensor([w.data_ptr(), dst.data_ptr(), M, K, K, 0], dtype=torch.int64).to(d_)
        L.call("gdmae_tok_gemm_pack", L.ptr(jobs), 1, L.stream())
        return dst

    X = torch.randn(n_pad, d, generator=g).bfloat16().to(d_)
    W1, W2 = (torch.randn(ff, d, generator=g) / d ** 0.5).to(d_), (torch.randn(d, ff, generator=g) / ff ** 0.5).to(d_)
    W1p, W2p = pack(W1), pack(W2)
    b1, b2 = torch.randn(ff, generator=g).bfloat16().to(d_), torch.randn(d, generator=g).bfloat16().to(d_)
    res = torch.randn(n, d, generator=g).to(d_)
    gamma, beta = (torch.rand(d, generator=g) + 0.5).to(d_), torch.randn(d, generator=g).to(d_)
    pos = torch.randn(64, d, generator=g).to(d_)
    tp = torch.randint(0, 64, (n,), generator=g).int().to(d_)

    def outs():
        return dict(y=torch.zeros(n, d, device=d_), st=torch.zeros(n, 2, device=d_), ybf=torch.zeros(n_pad, d, dtype=torch.bfloat16, device=d_),
                    ypos=torch.zeros(n_pad, d, dtype=torch.bfloat16, device=d_), f=torch.zeros(n_pad, d, dtype=torch.bfloat16, device=d_),
                    h=torch.zeros(n_pad, ff, dtype=torch.bfloat16, device=d_))
    for full in (True, False):
        a, b = outs(), outs()
        gact = torch.empty(n_pad, ff, dtype=torch.bfloat16, device=d_)
        L.call("gdmae_tok_gemm", L.ptr(X), L.ptr(W1p), L.ptr(b1), n_pad, n_pad, d, ff, 1, L.ptr(a["h"]), L.ptr(gact), None, None, None, None,
               1e-5, None, None, None, None, None, None, L.stream())
        L.call("gdmae_tok_gemm", L.ptr(gact), L.ptr(W2p), L.ptr(b2), n, n_pad, ff, d, 3, L.ptr(a["f"]), None, None, L.ptr(res), L.ptr(gamma),
               L.ptr(beta), 1e-5, L.ptr(a["y"]), L.ptr(a["st"]), L.ptr(a["ybf"]) if full else None, L.ptr(pos) if full else None,
               L.ptr(tp) if full else None, L.ptr(a["ypos"]) if full else None, L.stream())
        L.call("gdmae_tok_gemm_ffn", L.ptr(X), L.ptr(W1p), L.ptr(b1), L.ptr(W2p), L.ptr(b2), n, n_pad, d, L.ptr(b["h"]), L.ptr(res), L.ptr(gamma),
               L.ptr(beta), 1e-5, L.ptr(b["y"]), L.ptr(b["st"]), L.ptr(b["ybf"]) if full else None, L.ptr(pos) if full else None,
               L.ptr(tp) if full else None, L.ptr(b["ypos"]) if full else None, L.ptr(b["f"]), L.stream())
        for k in a:
            assert torch.equal(a[k], b[k]), (k, full)
        assert float(a["y"].abs().max()) > 0.5
        # backward side: dh and gelu(h) from the GELU-backward epilogue
        dY = torch.randn(n_pad, d, generator=g).bfloat16().to(d_)
        W2t = pack(W2.t().contiguous())
        dh0, dh1, g1 = (torch.empty(n_pad, ff, dtype=torch.bfloat16, device=d_) for _ in range(3))
        for o1, dh in ((None, dh0), (g1, dh1)):
            L.call("gdmae_tok_gemm", L.ptr(dY), L.ptr(W2t), None, n_pad, n_pad, d, ff, 2, L.ptr(dh), L.ptr(o1), L.ptr(b["h"]), None, None, None,
                   1e-5, None, None, None, None, None, None, L.stream())
        assert torch.equal(dh0, dh1) and torch.equal(g1, gact)


@pytest.mark.parametrize("cin,cout,src_f32", [(128, 128, False), (128, 256, True), (256, 128, False), (256, 256, True), (256, 256, False)])
def test_spconv_implicit_gemm_matches_gathered_product(cin, cout, src_f32):
    """gdmae_spconv (implicit GEMM over a rulebook, packed per-tap weight images) against the explicit formulation it replaces:
    gather the 9 neighbour rows (zeros where the rulebook says -1), round them to bf16, multiply by the bf16 weights with fp32
    accumulation, round to bf16.  Ragged row count (not a multiple of the 32 / 64-row tile), missing taps, repeated sources;
    forward images and the transposed (input-gradient) images."""
    import ctypes as C
    from gdmae_hip import lib as L
    g = torch.Generator().manual_seed(cin + cout)
    n_src, n = 1500, 1237
    X = torch.randn(n_src, cin, generator=g).to(dev())
    nbr = torch.randint(-1, n_src, (n, 9), generator=g).int()
    nbr[::7] = -1                                              # rows without any active tap
    nbr[5] = 3                                                 # all taps read the same source row
    nbr = nbr.to(dev())
    W = (torch.randn(cout, 3, 3, cin, generator=g) * 0.05).to(dev())
    lib = L.load()
    for transposed in (0, 1):
        ci, co = (cin, cout) if not transposed else (cout, cin)      # the input-gradient convolution swaps the roles
        Xs = X if not transposed else torch.randn(n_src, ci, generator=g).to(dev())
        src = Xs if src_f32 else Xs.bfloat16()
        packed = torch.empty(lib.gdmae_spconv_packed_bytes(cin, cout), dtype=torch.uint8, device=dev())
        jobs = (C.c_longlong * 54)()
        L.call("gdmae_spconv_pack_jobs", L.ptr(W), cin, cout, transposed, L.ptr(packed), jobs)
        jd = torch.tensor(list(jobs), dtype=torch.int64).to(dev())
        L.call("gdmae_tok_gemm_pack", L.ptr(jd), 9, L.stream())
        Y = torch.empty(n, co, dtype=torch.bfloat16, device=dev())
        L.call("gdmae_spconv", L.ptr(src.contiguous()), int(src_f32), L.ptr(nbr), L.ptr(packed), n, ci, co, L.ptr(Y), 0, L.stream())
        xb = Xs.bfloat16().float()
        Wb = W.bfloat16().float()
        ref = torch.zeros(n, co, device=dev())
        for k in range(9):
            rows = torch.where(nbr[:, k:k + 1] >= 0, xb[nbr[:, k].clamp(min=0).long()], torch.zeros(1, device=dev()))
            Wk = Wb[:, k // 3, k % 3, :]                       # (cout, cin)
            ref += rows @ (Wk.t() if not transposed else Wk)
        err = float((Y.float() - ref).abs().max())
        assert err <= 2 ** -7 * float(ref.abs().max()), (transposed, err, float(ref.abs().max()))     # one bf16 rounding of the result
        assert float(Y[::7].abs().max()) == 0.0
        # the same launch with the BatchNorm statistics as its epilogue: identical rows, partial rows that add up to the column sums
        # of the ROUNDED rows (fp32 partials over <= 128 rows, combined in fp64: 1e-6)
        rpw = lib.gdmae_spconv_stat_rows(ci, co, int(src_f32))
        nblk = (n + rpw - 1) // rpw
        part = torch.full((nblk, 2, co), float("nan"), device=dev())
        Y2 = torch.empty_like(Y)
        L.call("gdmae_spconv_stats", L.ptr(src.contiguous()), int(src_f32), L.ptr(nbr), L.ptr(packed), n, ci, co, L.ptr(Y2), L.ptr(part), L.stream())
        assert torch.equal(Y2, Y)
        s = part.double().sum(0).cpu()
        yd = Y.double().cpu()
        assert torch.allclose(s[0], yd.sum(0), rtol=1e-6, atol=1e-6 * float(yd.abs().sum(0).max()))
        assert torch.allclose(s[1], (yd * yd).sum(0), rtol=1e-6, atol=0)


@pytest.mark.gpu
def test_bf16_stores_round_to_nearest_even_like_torch():
    """Every bf16 epilogue converts with the hardware instruction (common.h gd_pack_bf16): bit-identical to torch's cast (round to
    nearest even) on random bit patterns, exact ties, subnormals and infinities; NaN stays NaN.  (-0.0 is left out: the carrier
    kernel adds its optional operands to +0.)"""
    from gdmae_hip import lib as L
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    bits = torch.randint(-2**31 + 1, 2**31 - 1, (1 << 20,), generator=g, dtype=torch.int64).to(torch.int32)
    ties = (torch.randint(0, 1 << 16, (1 << 16,), generator=g, dtype=torch.int64) << 16 | 0x8000).to(torch.int32)       # exactly half way
    near = ties + torch.randint(-1, 2, ties.shape, generator=g, dtype=torch.int32)
    special = torch.tensor([0, 0x7F800000, -8388608, 1, 0x00007FFF, 0x00008000, 0x00008001, 0x007FFFFF, 0x7F7FFFFF, 0x7F7F8000,
                            0x7FC00000, 0x7F800001], dtype=torch.int32)
    a = torch.cat([bits, ties, near, special]).view(torch.float32).to(dev)
    out = torch.empty(a.numel(), dtype=torch.bfloat16, device=dev)
    L.call("gdmae_add3_to", L.ptr(a), None, 0, None, 0, a.numel(), L.ptr(out), 1, L.stream())
    torch.cuda.synchronize()
    ref = a.to(torch.bfloat16)
    nan = torch.isnan(a)
    assert torch.equal(torch.isnan(out), nan)
    assert torch.equal(out[~nan].view(torch.int16), ref[~nan].view(torch.int16))


def test_lazy_chamfer_scale_is_used_only_when_nothing_else_sees_the_gradient():
    """ops.ChamferLoss hands the prediction head an UNSCALED gradient with the mean's scalars on the side (no scaling pass) - only while
    the head's backward is the sole observer of d loss / d pred_points (spt_backbone_mae._only_the_pred_head_sees_the_gradient).  A
    tensor hook on pred_points switches to the scaled gradient: the hook sees d loss / d pred, the parameter gradients do not change."""
    import logging
    from gdmae_hip import configs, optim
    from pcdet.models import build_network
    from pcdet.models.backbones_3d import spt_backbone_mae as M
    z, ds, cfg, shapes = load_case("kitti_b2_m75")
    res = {}
    for hook in (False, True):
        torch.manual_seed(0)
        net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev())
        net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
        net.train()
        opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)
        opt.zero_grad()
        seen = []
        orig = net.backbone_3d.target_assigner

        def hooked(bd, orig=orig, seen=seen, hook=hook):
            r = orig(bd)
            if hook:
                r['pred_points'].register_hook(lambda g: seen.append(g.detach().clone()))
            res[("lazy", hook)] = bool(r['pred_lazy_scale']) and M._only_the_pred_head_sees_the_gradient(r['pred_points'])
            return r
        net.backbone_3d.target_assigner = hooked
        bd = {"points": torch.from_numpy(z["points"]).to(dev()), "batch_size": int(z["batch_size"]), "mae_noise": torch.from_numpy(z["noise"]).to(dev())}
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ret, _, _ = net(bd)
        ret["loss"].backward()
        res[hook] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        if hook:
            assert len(seen) == 1 and float(seen[0].abs().sum()) > 0
            # the scaled gradient: |d loss / d pred| sums to O(1 / pillars), the unscaled one to O(1) per pillar
            res["hook_grad_sum"] = float(seen[0].abs().sum())
    assert res[("lazy", False)] is True and res[("lazy", True)] is False
    for k in res[False]:
        a, b = res[False][k].double(), res[True][k].double()
        assert float((a - b).norm()) <= 2e-2 * float(b.norm()) + 1e-12, k       # bf16 rounding of the scaled vs unscaled gradient rows
