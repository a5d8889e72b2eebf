"""GPU parity at BASELINE.json's full size (config B: 8 synthetic Waymo-shape frames, ~1.4 M points, ~127 k pillars).

Numeric parity: ONE full-size frame (180 k points) runs through the CPU oracle in a few seconds, so the HIP fp32 path is
compared with it directly (indices / mask bit-exact, Chamfer loss 1e-4, sampled features); at 8 frames the throughput
configuration bench.py times (bf16 autocast + fused layers + tile convolution + flat optimizer) is compared with the HIP
fp32 parity mode on the same weights and inputs.  Everything else through size-independent properties of the domain:

* voxelization: the CSR is a permutation grouped by pillar, offsets / inverse / coordinates / means agree with each other
  and with an independent torch computation (unique cells, index_add means);
* masking: exact per-sample keep counts; token sets and window partitions: every token exactly once, window sizes
  within the capacity of their level;
* window attention: with v = 1 every output row is 1 (softmax rows sum to one), for every level and implementation;
* Chamfer: zero for identical sets; DynVFE point layers: invariance of the pillar features to a permutation of the
  input points (fused layers, bf16 mode);
* the whole training step: finite loss and gradients for every parameter, bit-identical when repeated (deterministic
  kernels, no atomics in floating point).
"""
import logging

import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def scene():
    from gdmae_hip import configs, plan as gplan, synth
    from pcdet.models.backbones_3d.spt_backbone import stage_plan_args
    cfg, ds, skw = configs.named_config("B", mask_ratio=0.75)
    B = 8
    pts = torch.from_numpy(synth.synth_batch(4242, B, ds.point_cloud_range, **skw)).to(dev())
    vox = gplan.voxelize(pts, ds.point_cloud_range, ds.voxel_size, ds.grid_size, B)
    ep = gplan.encoder_plan(vox, *stage_plan_args(cfg.BACKBONE_3D.SST_BLOCK_LIST), keep_frac=0.25)
    return cfg, ds, skw, B, pts, vox, ep


def test_voxelization_invariants_at_full_size(scene):
    cfg, ds, skw, B, pts, vox, ep = scene
    N, M = vox.N, vox.M
    assert N > 1_000_000 and M > 100_000
    gx, gy, gz = vox.grid
    # kept points = the in-range points, original order
    lo = torch.tensor(vox.lo, device=dev())
    vs = torch.tensor(vox.vs, device=dev())
    hi = lo + vs * torch.tensor([gx, gy, gz], device=dev(), dtype=torch.float32)
    xyz = pts[:, 1:4]
    keep = ((xyz >= lo) & (xyz < hi)).all(1)
    assert int(keep.sum()) == N
    assert torch.equal(vox.points, pts[keep])
    # pillars = unique occupied cells in lexicographic (b, z, y, x) order
    c = vox.point_coords
    key = ((c[:, 0] * gz + c[:, 1]) * gy + c[:, 2]) * gx + c[:, 3]
    uk, inv = torch.unique(key, return_inverse=True)
    assert uk.numel() == M
    assert torch.equal(inv, vox.inverse) and torch.equal(vox.inverse32.long(), vox.inverse)
    vk = ((vox.voxel_coords[:, 0] * gz + vox.voxel_coords[:, 1]) * gy + vox.voxel_coords[:, 2]) * gx + vox.voxel_coords[:, 3]
    assert torch.equal(vk, uk) and torch.equal(vox.pillar_cell.long(), uk)
    # CSR: a permutation of the points, grouped by pillar, ascending ids inside a pillar
    csr = vox.pillar_pts.long()
    assert torch.equal(torch.sort(csr).values, torch.arange(N, device=dev()))
    cnt = torch.bincount(vox.inverse, minlength=M)
    assert int(vox.pt_off[0]) == 0 and torch.equal(vox.pt_off[1:].long(), torch.cumsum(cnt, 0))
    rowpil = vox.inverse[csr]
    assert bool((rowpil[1:] >= rowpil[:-1]).all())
    same = rowpil[1:] == rowpil[:-1]
    assert bool((csr[1:][same] > csr[:-1][same]).all())
    assert torch.equal(vox.row_pillar.long(), rowpil) and torch.equal(vox.points_pm, vox.points[csr])
    # scatter-mean (the kernel sums in canonical order; torch's index_add order differs: fp32 tolerance 1e-4 m)
    F = vox.n_cols - 1
    ref = torch.zeros(M, F, device=dev(), dtype=torch.float64).index_add_(0, vox.inverse, vox.points[:, 1:].double())
    ref = (ref / cnt[:, None]).float()
    assert torch.allclose(vox.pillar_mean, ref, rtol=0, atol=1e-4)
    assert int(vox.sample_off[-1]) == M and (cnt > 0).all()


def test_mask_tokens_and_windows_at_full_size(scene):
    cfg, ds, skw, B, pts, vox, ep = scene
    M = vox.M
    so = vox.sample_off.tolist()
    mask = ep.mask[:M]
    assert set(torch.unique(mask).tolist()) <= {0.0, 1.0}
    for b in range(B):
        n = so[b + 1] - so[b]
        assert int((mask[so[b]:so[b + 1]] == 0).sum()) == int(n * 0.25)      # common_utils.random_masking: int(L * keep)
    vis = torch.nonzero(mask == 0)[:, 0]
    assert torch.equal(ep.tok_pillar.long(), vis)
    for st in ep.stages:
        assert st.n_tok > 0 and bool((st.tok_cell[1:] > st.tok_cell[:-1]).all())
        assert torch.equal(st.map[st.tok_cell.long()], torch.arange(st.n_tok, device=dev(), dtype=torch.int32))
        assert int((st.map >= 0).sum()) == st.n_tok
        for wp in st.windows:
            assert torch.equal(torch.sort(wp.csr_tok.long()).values, torch.arange(st.n_tok, device=dev()))
            assert sum(wp.n_tok) == st.n_tok and int(wp.win_len.sum()) == st.n_tok
            base = 0
            for lvl, nw in enumerate(wp.n_win):
                ln = wp.win_len[base:base + nw]
                assert nw == 0 or (int(ln.min()) >= 1 and int(ln.max()) <= wp.max_tokens[lvl])
                base += nw
            # tokens of one window are contiguous in csr_tok and share tok_win
            tw = wp.tok_win[wp.csr_tok.long()]
            assert int((tw[1:] != tw[:-1]).sum()) + 1 == sum(wp.n_win)


@pytest.mark.parametrize("impl", [0, 1, 2])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_attention_rows_sum_to_one_at_full_size(scene, impl, dt):
    from gdmae_hip import lib as L, ops
    cfg, ds, skw, B, pts, vox, ep = scene
    st = ep.stages[1]
    d, nhead = 256, 8
    gen = torch.Generator(device="cpu").manual_seed(3)
    qk = torch.randn(st.n_tok, 2 * d, generator=gen).to(dev()).to(dt)
    v = torch.ones(st.n_tok, d, device=dev(), dtype=dt)
    tau = torch.full((1, 1, 1), 0.07, device=dev())
    L.call("gdmae_set_attention_impl", impl)
    try:
        for wp in st.windows:
            out = ops.WindowCosineAttention.apply(qk, v, tau, wp, nhead, 0.01)
            assert torch.allclose(out.float(), torch.ones_like(out, dtype=torch.float32), rtol=0, atol=2e-5 if dt == torch.float32 else 8e-3)
    finally:
        L.call("gdmae_set_attention_impl", 2)


def test_chamfer_and_point_layers_at_full_size(scene):
    from gdmae_hip import ops, plan as gplan
    from pcdet.models import build_network
    cfg, ds, skw, B, pts, vox, ep = scene
    gt = ops.group_gt_points(vox, 64)[:, :16].contiguous()
    w = torch.ones(vox.M, device=dev())
    loss = ops.ChamferLoss.apply(gt.clone(), gt, w)
    assert float(loss) == 0.0
    # pillar features do not depend on the order of the input points (fused point layers, bf16 mode)
    torch.manual_seed(0)
    net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev()).train()
    perm = torch.randperm(pts.shape[0], generator=torch.Generator(device="cpu").manual_seed(5)).to(dev())
    outs = []
    for p in (pts, pts[perm]):
        bd = {"points": p, "batch_size": B}
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            outs.append(net.vfe(bd)["pillar_features"].float())
    assert outs[0].shape == (vox.M, 128)
    # the pillar means and the BatchNorm statistics are sums in a different order: fp32 / bf16-input rounding only
    assert float((outs[0] - outs[1]).abs().max()) <= 2e-2 * float(outs[0].abs().max())
    assert float((outs[0] - outs[1]).norm()) <= 2e-3 * float(outs[0].norm())


def test_train_step_is_finite_and_deterministic_at_full_size(scene):
    from gdmae_hip import configs, optim
    from pcdet.models import build_network
    cfg, ds, skw, B, pts, vox, ep = scene
    gen = torch.Generator(device="cpu").manual_seed(11)
    noise = torch.rand(vox.M, generator=gen).to(dev())
    res = []
    for rep in range(2):
        torch.manual_seed(7)
        net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev()).train()
        net.sync_loss_scalar = False
        net.backbone_3d.dense_spatial_features = False
        opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(B), total_steps=10)
        opt.zero_grad()
        bd = {"points": pts, "batch_size": B, "mae_noise": noise}
        with torch.autocast("cuda", dtype=torch.bfloat16):
            ret, _, _ = net(bd)
        ret["loss"].backward()
        res.append((ret["loss"].detach().clone(), opt.flat_grad.clone()))
        if rep == 0:
            missing = [k for k, p in net.named_parameters() if p.grad is None]
            assert not missing, missing
    loss, grad = res[0]
    assert torch.isfinite(loss) and 0.0 < float(loss) < 100.0
    assert torch.isfinite(grad).all() and float(grad.norm()) > 0
    assert torch.equal(res[0][0], res[1][0]), (float(res[0][0]), float(res[1][0]))
    assert torch.equal(res[0][1], res[1][1]), float((res[0][1] - res[1][1]).abs().max())


@pytest.mark.parametrize("config,min_points", [("B", 170_000), ("E", 55_000)])
def test_one_full_size_frame_matches_the_oracle(config, min_points):
    """Configs B (Waymo-shape, 180 k points, 12 layers 128 / 256 / 256) and E (ONCE-shape, 60 k points, 6 layers d = 256) at full
    size, ONE frame (the oracle needs a few seconds for it): HIP fp32 path vs oracle/gdmae_oracle.forward on the same seeded frame,
    weights and masking noise - voxel indices, inverse map, token mask, stage active sets bit-exact, Chamfer loss within 1e-4
    relative (north_star), stage / decoder features within 5e-4."""
    from gdmae_hip import configs, synth
    from oracle import gdmae_oracle as orc
    from pcdet.models import build_network
    cfg, ds, skw = configs.named_config(config, mask_ratio=0.75)
    pts = synth.synth_batch(31337, 1, ds.point_cloud_range, **skw)
    assert pts.shape[0] > min_points
    shapes = orc.param_shapes(cfg, ds.point_feature_encoder.num_point_features)
    sd = orc.seeded_state_dict(shapes, seed=21)
    o = orc.forward(torch.from_numpy(pts), 1, cfg, {k: v.clone() for k, v in sd.items()}, ds.point_cloud_range, ds.voxel_size,
                    ds.grid_size, noise_seed=9)
    M = o["voxel_coords"].shape[0]
    noise = torch.rand(M, generator=torch.Generator().manual_seed(9))
    net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev()).train()
    net.load_state_dict(sd, strict=False)
    bd = {"points": torch.from_numpy(pts).to(dev()), "batch_size": 1, "mae_noise": noise.to(dev())}
    ret, _, _ = net(bd)
    assert torch.equal(bd["voxel_coords"].cpu(), o["voxel_coords"])
    assert torch.equal(bd["point_inverse_indices"].cpu(), o["point_inverse_indices"])
    assert torch.equal(bd["voxel_mae_mask"].cpu(), o["voxel_mae_mask"])
    ep = bd["_gdmae_plan"]
    for i, tr in enumerate(o["stage_trace"]):
        assert torch.equal(ep.stages[i].indices_byx().cpu().long(), tr["coords"][:, [0, 2, 3]]), f"stage {i} active set"
        f = bd["multi_scale_3d_features"][f"x_conv{i + 1}"].features.detach().float().cpu()
        assert float((f - tr["features"]).abs().max()) <= 5e-4 * float(tr["features"].abs().max()), f"stage {i} features"
    sf = bd["spatial_features"].detach().float().cpu()
    assert float((sf - o["spatial_features"]).abs().max()) <= 5e-4 * float(o["spatial_features"].abs().max())
    fr = net.backbone_3d.forward_ret_dict
    assert torch.equal(fr["gt_points"].cpu(), o["gt_points"])
    rel = abs(float(ret["loss"]) - float(o["loss"])) / abs(float(o["loss"]))
    assert rel < 1e-4, (float(ret["loss"]), float(o["loss"]), rel)


# bounds of the bench (bf16) mode against the fp32 parity mode at full size: <= 2 x the deviations measured on MI355X (printed by the
# test; DESIGN.md section 5 keeps the measured values) so that a regression in a fused epilogue cannot hide under a loose bound
# measured (round 4; two states of the bench path - before / after the decoder's library GEMMs were replaced): loss 9.9e-5 / 1.26e-4,
# gradient norms 3.1e-2 / 1.7e-2, cosine 0.99412 / 0.99417.  The temperature gradients are sums of O(1e-4) terms that cancel to 2e-5 ..
# 2e-4 and sit on an ABSOLUTE noise floor set by the bf16 q / k / v rows: 5.6e-2 / 1.5e-1 of the largest |dtau| (vector 4.8e-2 / 1.0e-1)
# for those two arithmetically equivalent states, 1.7e-1 in round 3 - their bound stays at the floor, not at 2 x one sample of it
# round 6: LOSS_REL is north_star's own 1e-4 (the decoder's forward products take fp16 operands: the bf16 rounding of the deconvolution
# and conv_out WEIGHTS - the same error at every site - was +7.7e-5 + 6.2e-5 of round 5's 1.5e-4, tools/weight_rounding_full_size.py);
# measured 5.6e-5 (weight seed 7) and 3.8e-5 (seed 3) in the final build (3.6e-5 / 2.5e-5 before DynVFE's rows became fp16: samples)
LOSS_REL, NORM_REL, COS_MIN, TAU_ABS, TAU_L2 = 1.0e-4, 0.065, 0.988, 0.20, 0.20


@pytest.mark.parametrize("weight_seed", [7, 3])
def test_bench_mode_matches_fp32_mode_at_full_size(scene, weight_seed):
    """8 full-size frames, the exact configuration bench.py times (bf16 autocast, fused VFE layers, stage executor, tile
    convolution, flat optimizer with bf16 weight shadows) against the HIP fp32 parity mode with the dense decoder dataflow
    (the mode held to 1e-4 of the reference above and in test_hip_parity) on the same weights, frames and masking noise:
    identical geometry; loss within north_star's 1e-4 (LOSS_REL; round 6 measured 5.6e-5 / 3.8e-5 for the two weight seeds, round 5 1.4e-4),
    every parameter's gradient norm and direction within NORM_REL 6.5 %, COS_MIN 0.988 (2 x the measured 2.3 %, 0.9944).  tau (one scalar per layer whose
    gradient is a heavily cancelling sum over all (window, head, query, key) terms: 2e-5 .. 2e-4 here) carries an ABSOLUTE noise
    floor from the bf16 q/k/v rows: it is bounded by |dtau_bf16 - dtau_fp32| <= TAU_ABS (20 %) of the largest |dtau_fp32| over the
    layers + 10 % of its own value, and the 12-vector of tau gradients by a relative L2 error of TAU_L2 (20 %); measured 5.6 - 15 % /
    4.8 - 10 % over rounds 4 and 5."""
    import numpy as np
    from gdmae_hip import configs, optim
    from pcdet.models import build_network
    cfg, ds, skw, B, pts, vox, ep = scene
    noise = torch.rand(vox.M, generator=torch.Generator(device="cpu").manual_seed(11)).to(dev())
    res = {}
    for mode in ("bench", "fp32"):
        torch.manual_seed(weight_seed)
        net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev()).train()
        net.sync_loss_scalar = False
        names = [n for n, _ in net.named_parameters()]
        if mode == "bench":
            net.backbone_3d.dense_spatial_features = False
            opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(B), total_steps=10)
            opt.zero_grad()
        else:
            net.backbone_3d.decoder_impl = "dense"
        bd = {"points": pts, "batch_size": B, "mae_noise": noise}
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=mode == "bench"):
            ret, _, _ = net(bd)
        ret["loss"].backward()
        res[mode] = (float(ret["loss"].detach()), {n: p.grad.detach().double().cpu() for n, p in net.named_parameters()},
                     bd["voxel_mae_mask"].clone(), bd["voxel_coords"].clone())
        del net, bd, ret
        torch.cuda.empty_cache()
    lb, gb, mb, cb = res["bench"]
    lf, gf, mf, cf = res["fp32"]
    assert torch.equal(mb, mf) and torch.equal(cb, cf)
    taus = [n for n in gf if n.endswith("tau")]
    tau_scale = max(float(gf[n].abs().max()) for n in taus)
    bad = []
    worst = {"norm": 0.0, "cos": 1.0, "tau": 0.0}
    for n in gf:
        a, b = gb[n].reshape(-1), gf[n].reshape(-1)
        if n.endswith("tau"):
            worst["tau"] = max(worst["tau"], abs(float(a[0] - b[0])) / tau_scale)
            if abs(float(a[0] - b[0])) > TAU_ABS * tau_scale + 0.10 * abs(float(b[0])):
                bad.append((n, float(a[0]), float(b[0])))
            continue
        na, nb = float(a.norm()), float(b.norm())
        cos = float((a * b).sum()) / (na * nb + 1e-300)
        worst["norm"], worst["cos"] = max(worst["norm"], abs(na - nb) / nb), min(worst["cos"], cos)
        if abs(na - nb) > NORM_REL * nb or cos < COS_MIN:
            bad.append((n, na, nb, cos))
    tb, tf = torch.stack([gb[n][0] for n in taus]), torch.stack([gf[n][0] for n in taus])
    tau_l2 = float((tb - tf).norm()) / float(tf.norm())
    print(f"[bench vs fp32, full size] loss rel {abs(lb - lf) / abs(lf):.3e}; worst gradient-norm deviation {worst['norm']:.3e}, worst cosine "
          f"{worst['cos']:.5f}; tau: worst |d| / max|dtau| {worst['tau']:.3e}, vector L2 rel {tau_l2:.3e}")
    assert abs(lb - lf) <= LOSS_REL * abs(lf), (lb, lf)
    if weight_seed != 7:
        return        # the gradient bounds are 2 x what seed 7 measures (tau: its noise floor); the second seed pins the LOSS bound only
    assert tau_l2 <= TAU_L2, (tb.tolist(), tf.tolist())
    assert not bad, bad


def test_one_call_plan_equals_per_operator_plan_at_full_size(scene):
    """The geometry plan as ONE library call (gdmae_geometry_plan: single-launch look-back scans with hundreds of tiles, merged
    launches, one arena) against the per-operator entry points on 8 full-size frames with the same masking noise: every index
    structure bit-identical - voxel table, CSR, ranks, means, pillar-major rows, mask, token sets, maps, rulebooks incl. the
    transposed ones, window partitions of both shifts, upsampled sites, active decoder tiles."""
    import dataclasses
    from gdmae_hip import plan as gplan
    from pcdet.models.backbones_3d.spt_backbone import stage_plan_args
    cfg, ds, skw, B, pts, vox0, _ = scene
    noise = torch.rand(vox0.M, generator=torch.Generator().manual_seed(5)).to(dev())
    args = stage_plan_args(cfg.BACKBONE_3D.SST_BLOCK_LIST)
    ep0 = gplan.encoder_plan(vox0, *args, keep_frac=0.25, noise=noise, dec_sources=[0, 1, 2])
    cap_noise = torch.cat([noise, torch.rand(pts.shape[0] - noise.numel(), device=dev())])
    for rep in range(2):                                   # twice: the arena / look-back states of a previous plan are reused memory
        vox1, ep1 = gplan.PlanPrefetch(pts, ds.point_cloud_range, ds.voxel_size, ds.grid_size, B, *args, keep_frac=0.25, noise=cap_noise,
                                       dec_sources=[0, 1, 2]).finish()
        assert (vox1.N, vox1.M) == (vox0.N, vox0.M)
        for name in ("points", "point_coords", "inverse", "inverse32", "voxel_coords", "pillar_cell", "pt_off", "pillar_pts", "point_rank",
                     "sample_off", "pillar_mean", "cell2pillar", "points_pm", "row_pillar"):
            assert torch.equal(getattr(vox0, name), getattr(vox1, name)), name
        assert torch.equal(ep0.mask, ep1.mask) and torch.equal(ep0.tok_pillar, ep1.tok_pillar)
        for i, (a, b) in enumerate(zip(ep0.stages, ep1.stages)):
            assert (a.B, a.Y, a.X, a.n_tok) == (b.B, b.Y, b.X, b.n_tok)
            for name in ("tok_cell", "map", "nbr_subm", "nbr_down", "nbr_down_t", "_nbr_subm_t"):
                x, y = getattr(a, name), getattr(b, name)
                assert (x is None) == (y is None) and (x is None or torch.equal(x, y)), (i, name)
            assert (a._up_sites is None) == (b._up_sites is None)
            if a._up_sites is not None:
                assert a._up_sites[0] == b._up_sites[0] and torch.equal(a._up_sites[1], b._up_sites[1])
            for k, (wa, wb) in enumerate(zip(a.windows, b.windows)):
                for f in dataclasses.fields(wa):
                    x, y = getattr(wa, f.name), getattr(wb, f.name)
                    assert torch.equal(x, y) if isinstance(x, torch.Tensor) else x == y, (i, k, f.name)
        assert ep0.dec_tiles.n_act == ep1.dec_tiles.n_act and torch.equal(ep0.dec_tiles.tile_slot, ep1.dec_tiles.tile_slot)
        assert torch.equal(ep0.dec_tiles.tile_list, ep1.dec_tiles.tile_list)
        del vox1, ep1
