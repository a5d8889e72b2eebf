"""Generate the golden vectors under tests/golden/ (BUILD CONTAINER ONLY: reads /root/reference).

    python tests/golden/make_golden.py

For every case it (1) runs the UNMODIFIED reference modules (imported through ref_harness.py) on a
seeded synthetic batch with a seeded state_dict and injected masking noise, (2) runs the CPU oracle
(oracle/gdmae_oracle.py) on the same inputs and asserts agreement, and (3) stores inputs' seeds and
the reference's outputs as a small .npz.  Integer outputs are stored in full; large float tensors
are stored as strided samples plus double-precision checksums.

Fixtures are data (inputs/expected outputs) - no reference source text is stored.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "gd-mae_amd"))

import ref_harness as rh  # noqa: E402

rh.install()
from oracle import gdmae_oracle as orc  # noqa: E402

# the product-side config builders must not see the harness' fake 'pcdet' namespace: load by path
import importlib.util  # noqa: E402


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


_cfgmod = _load(os.path.join(REPO, "gd-mae_amd", "pcdet", "config.py"), "_our_pcdet_config")
sys.modules["pcdet.config"] = _cfgmod          # what gdmae_hip.configs imports
configs = _load(os.path.join(REPO, "gd-mae_amd", "gdmae_hip", "configs.py"), "_our_configs")
synth = _load(os.path.join(REPO, "gd-mae_amd", "gdmae_hip", "synth.py"), "_our_synth")
del sys.modules["pcdet.config"]


def to_plain(x):
    if isinstance(x, dict):
        return {k: to_plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [to_plain(v) for v in x]
    return x


def sample(t: torch.Tensor, n=4096):
    f = t.detach().reshape(-1).double()
    step = max(1, f.numel() // n)
    return f[::step][:n].float().numpy(), np.array([float(f.sum()), float(f.abs().sum()), float((f * f).sum())])


CASES = {
    # name: (dataset, yaml, batch, synth kwargs, mask ratio override, seeds)
    "kitti_b2": dict(ds=configs.KITTI, yaml="cfgs/kitti_models/gd_mae_ssl.yaml", B=2,
                     synth=dict(beams=24, azimuths=300, extra=500, features=4), ratio=None, seed=11),
    "kitti_b2_m75": dict(ds=configs.KITTI, yaml="cfgs/kitti_models/gd_mae_ssl.yaml", B=2,
                         synth=dict(beams=24, azimuths=300, extra=500, features=4), ratio=0.75, seed=12),
    "waymo_b1": dict(ds=configs.WAYMO, yaml="cfgs/waymo_models/gd_mae_ssl.yaml", B=1,
                     synth=dict(beams=32, azimuths=500, extra=1500, features=5), ratio=0.75, seed=13),
    # BASELINE config E (ONCE-shape, 6-layer SRA, d = 256 in every stage, VFE 64 -> 256): the ONCE yaml with the
    # BASELINE-defined overrides of SURVEY section 8d applied to the reference's own config object
    "once_e_b1": dict(ds=configs.ONCE, yaml="cfgs/once_models/gd_mae_ssl.yaml", B=1,
                      synth=dict(beams=24, azimuths=400, extra=1200, features=4), ratio=0.75, seed=14, config_e=True),
}


def apply_config_e(model_cfg):
    model_cfg.VFE.MLPS = [[64, 256]]
    for blk in model_cfg.BACKBONE_3D.SST_BLOCK_LIST:
        blk.ENCODER.NUM_BLOCKS = 1
        blk.ENCODER.D_MODEL = 256
        blk.ENCODER.DIM_FEEDFORWARD = 512
    for src in model_cfg.BACKBONE_3D.FEATURES_SOURCE:
        model_cfg.BACKBONE_3D.FUSE_LAYER[src].NUM_FILTER = 256


def run_case(name, c):
    print(f"== {name}")
    ycfg = rh.load_yaml_cfg(c["yaml"])
    model_cfg = ycfg.MODEL
    if c["ratio"] is not None:
        model_cfg.BACKBONE_3D.MASK_CONFIG.RATIO = c["ratio"]
    if c.get("config_e"):
        apply_config_e(model_cfg)
        ours = configs.named_config("E", mask_ratio=model_cfg.BACKBONE_3D.MASK_CONFIG.RATIO)[0]
    else:
        ours = configs.gdmae_ssl_model_cfg(mask_ratio=model_cfg.BACKBONE_3D.MASK_CONFIG.RATIO,
                                           eval_metric=model_cfg.POST_PROCESSING.EVAL_METRIC)
    ref_plain, our_plain = to_plain(model_cfg), to_plain(ours)
    for k in ("USE_GROUND_MASK", "DIS_THRESH", "NUM_ABOVE_GROUND"):   # dead keys of the ONCE yaml
        ref_plain["BACKBONE_3D"]["MASK_CONFIG"].pop(k, None)
    assert ref_plain == our_plain, "config builder drifted from the reference yaml MODEL section"

    ds = configs.SyntheticDatasetInfo(**c["ds"])
    assert list(ycfg.DATA_CONFIG.POINT_CLOUD_RANGE) == [float(v) for v in c["ds"]["point_cloud_range"]] or \
        [float(v) for v in ycfg.DATA_CONFIG.POINT_CLOUD_RANGE] == [float(v) for v in c["ds"]["point_cloud_range"]]
    assert [float(v) for v in ycfg.DATA_CONFIG.DATA_PROCESSOR[-1].VOXEL_SIZE] == [float(v) for v in c["ds"]["voxel_size"]]
    F = c["ds"]["num_point_features"]
    points = torch.from_numpy(synth.synth_batch(c["seed"], c["B"], ds.point_cloud_range, **c["synth"]))
    # inject a few out-of-range / boundary points so the range mask is exercised
    extra = points[:6].clone()
    pcr = [float(v) for v in ds.point_cloud_range]
    extra[0, 3] = pcr[5] + 0.5      # z above range -> dropped
    extra[1, 3] = pcr[2] - 0.3      # z slightly below lo -> truncates to 0, KEPT
    extra[2, 3] = pcr[2] - 7.0      # far below -> dropped
    extra[3, 1] = pcr[3]            # x == hi -> index == grid -> dropped
    extra[4, 1] = pcr[0]            # x == lo -> kept (index 0)
    extra[5, 2] = pcr[4] - 1e-4
    points = torch.cat([points, extra], 0)
    order = torch.argsort(points[:, 0], stable=True)  # keep frames grouped (collate order)
    points = points[order].contiguous()

    dyn = rh.ref("pcdet.models.backbones_3d.vfe.dyn_vfe")
    mae = rh.ref("pcdet.models.backbones_3d.spt_backbone_mae")
    cu = rh.ref("pcdet.utils.common_utils")
    vfe = dyn.DynVFE(model_cfg=model_cfg.VFE, num_point_features=F, voxel_size=ds.voxel_size,
                     point_cloud_range=ds.point_cloud_range, grid_size=ds.grid_size)
    bb = mae.SPTBackboneMAE(model_cfg=model_cfg.BACKBONE_3D, input_channels=vfe.get_output_feature_dim(),
                            grid_size=ds.grid_size, voxel_size=ds.voxel_size, point_cloud_range=ds.point_cloud_range)

    class Wrap(torch.nn.Module):
        def __init__(s):
            super().__init__()
            s.vfe, s.backbone_3d = vfe, bb
    net = Wrap()
    shapes = orc.param_shapes(our_plain, F)
    ref_shapes = {k: tuple(v.shape) for k, v in net.named_parameters()}
    assert ref_shapes == shapes, set(ref_shapes.items()) ^ set(shapes.items())
    sd = orc.seeded_state_dict(shapes, seed=c["seed"])
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(("running_" in m or "num_batches" in m) for m in missing), (missing, unexpected)
    net.train()

    # --- reference forward with injected noise
    bd = {"points": points.clone(), "batch_size": c["B"]}
    bd = vfe(bd)
    M = bd["voxel_coords"].shape[0]
    noise = torch.rand(M, generator=torch.Generator().manual_seed(c["seed"] + 1000))
    # the reference's tie-break (unstable torch.argsort, common_utils.py:57) is implementation defined, so the
    # injected noise must not tie across a keep boundary; ties are pinned by our own oracle-vs-HIP tests instead
    off = 0
    for bs in range(c["B"]):
        L = int((bd["voxel_coords"][:, 0] == bs).sum())
        kk = int(L * (1 - float(model_cfg.BACKBONE_3D.MASK_CONFIG.RATIO)))
        sv = torch.sort(noise[off:off + L]).values
        assert kk == 0 or kk == L or float(sv[kk - 1]) < float(sv[kk])
        off += L
    real_rand = torch.rand
    cursor = [0]

    def fake_rand(n, L, device=None):
        assert n == 1
        r = noise[cursor[0]:cursor[0] + L].view(1, L).clone()
        cursor[0] += L
        return r
    torch.rand = fake_rand
    try:
        bd = bb(bd)
    finally:
        torch.rand = real_rand
    assert cursor[0] == M
    loss, _ = bb.get_loss()
    loss.backward()
    fr = bb.forward_ret_dict

    # --- reference window partition internals (per stage, per shift) re-derived through the reference's own functions
    sst_utils = rh.ref("pcdet.models.model_utils.sst_utils")
    stage_ref = []
    for i, blk in enumerate(bb.sst_blocks):
        sp = bd["multi_scale_3d_features"][f"x_conv{i + 1}"]
        _, vc, grid = blk.decouple_sp_tensor(sp)
        info = blk.sst_input_layer({"voxel_features": sp.features.detach(), "voxel_coords": vc,
                                    "voxel_shuffle_inds": torch.arange(len(vc)), "grid_size": grid})
        assert torch.equal(info["voxel_keep_inds"], torch.arange(len(vc)))
        st = {"indices": sp.indices.clone(), "features": sp.features.detach()}
        for s in range(2):
            fl = info[f"flat2win_inds_shift{s}"]
            lvl = info[f"voxel_drop_level_shift{s}"]
            slot = -torch.ones_like(lvl)
            for dl in (0, 1, 2):
                if dl in fl:
                    slot[fl[dl][1][0]] = fl[dl][0]
            st[f"win_id{s}"] = info[f"batch_win_inds_shift{s}"]
            st[f"level{s}"] = lvl
            st[f"slot{s}"] = slot
            st[f"in_win{s}"] = info[f"coors_in_win_shift{s}"]
        stage_ref.append(st)

    # --- oracle on the same inputs
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    o = orc.forward(points.clone(), c["B"], our_plain, sdg, ds.point_cloud_range, ds.voxel_size, ds.grid_size,
                    noise=noise)
    o["loss"].backward()

    def close(a, b_, tol, what):
        err = float((a - b_).abs().max() / (b_.abs().max() + 1e-12))
        print(f"   {what:34s} rel-max-err {err:.3e}")
        assert err <= tol, (what, err)

    assert torch.equal(o["voxel_coords"], bd["voxel_coords"]), "voxel_coords"
    assert torch.equal(o["point_inverse_indices"], bd["point_inverse_indices"]), "inverse"
    assert torch.equal(o["points"], bd["points"]) and torch.equal(o["point_coords"], bd["point_coords"])
    assert torch.equal(o["voxel_mae_mask"], bd["voxel_mae_mask"]), "mask"
    for i, (st, tr) in enumerate(zip(stage_ref, o["stage_trace"])):
        assert torch.equal(tr["coords"][:, [0, 2, 3]], st["indices"].long()), f"stage {i} active set"
        for s in range(2):
            p = tr["parts"][s]
            assert torch.equal(p["win_id"], st[f"win_id{s}"]) and torch.equal(p["level"], st[f"level{s}"])
            assert torch.equal(p["slot"], st[f"slot{s}"]) and torch.equal(p["in_win"], st[f"in_win{s}"])
        close(tr["features"], st["features"], 2e-4, f"stage {i} features")
    close(o["pillar_features"], bd["pillar_features"].detach() if "pillar_features" in bd else o["pillar_features"], 1e-5, "pillar_features")
    close(o["spatial_features"], bd["spatial_features"].detach(), 2e-4, "spatial_features")
    close(o["pred_points"], fr["pred_points"].detach(), 2e-4, "pred_points")
    assert torch.equal(o["gt_points"], fr["gt_points"]), "gt_points (canonical grouping)"
    lerr = abs(float(o["loss"]) - float(loss)) / abs(float(loss))
    print(f"   loss ref {float(loss):.7f} oracle {float(o['loss']):.7f} rel {lerr:.2e}")
    assert lerr < 1e-5
    gref = dict(net.named_parameters())
    worst = 0.0
    for k in shapes:
        g1, g2 = sdg[k].grad, gref[k].grad
        e = float((g1 - g2).norm() / (g2.norm() + 1e-12))
        if e > (5e-2 if k.endswith('tau') else 5e-3):
            print(f"      grad {k}: rel {e:.3e}  |g| {float(g2.norm()):.3e}")
        worst = max(worst, e if not k.endswith("tau") else e / 10)   # tau grads: heavy cancellation in fp32
    print(f"   worst param-grad rel L2 err {worst:.3e}")
    assert worst < 5e-3

    # --- store
    z = {"seed": np.int64(c["seed"]), "batch_size": np.int64(c["B"]), "mask_ratio": np.float64(model_cfg.BACKBONE_3D.MASK_CONFIG.RATIO),
         "num_point_features": np.int64(F), "points": points.numpy(), "noise": noise.numpy(),
         "point_cloud_range": np.array(ds.point_cloud_range, dtype=np.float32), "voxel_size": np.array(ds.voxel_size, dtype=np.float64),
         "keep_count": np.int64(bd["points"].shape[0]),
         "voxel_coords": bd["voxel_coords"].numpy().astype(np.int32), "inverse": bd["point_inverse_indices"].numpy().astype(np.int32),
         "mask": bd["voxel_mae_mask"].numpy().astype(np.uint8), "loss": np.float64(float(loss)),
         "gt_group_inds": o["gt_group_inds"].numpy().astype(np.int32)}
    for nm, t in (("pillar_features", bd["pillar_features"]), ("spatial_features", bd["spatial_features"]),
                  ("pred_points", fr["pred_points"]), ("gt_points", fr["gt_points"])):
        z[nm + "_s"], z[nm + "_c"] = sample(t)
    for i, st in enumerate(stage_ref):
        z[f"st{i}_indices"] = st["indices"].numpy().astype(np.int32)
        z[f"st{i}_features_s"], z[f"st{i}_features_c"] = sample(st["features"])
        for s in range(2):
            z[f"st{i}_win_id{s}"] = st[f"win_id{s}"].numpy().astype(np.int32)
            z[f"st{i}_level{s}"] = st[f"level{s}"].numpy().astype(np.int8)
            z[f"st{i}_slot{s}"] = st[f"slot{s}"].numpy().astype(np.int32)
    names = sorted(shapes)
    z["grad_norm"] = np.array([float(gref[k].grad.double().norm()) for k in names])
    z["grad_head"] = np.stack([np.pad(gref[k].grad.reshape(-1)[:8].numpy(), (0, max(0, 8 - gref[k].grad.numel()))) for k in names])
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **z)
    print(f"   wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)  N={points.shape[0]} M={M}")


def optimizer_golden():
    """a21 known answers: reference OneCycle(lr, mom) schedule and 3 OptimWrapper steps on a toy model
    (tools/train_utils/optimization/{learning_schedules_fastai,fastai_optim}.py import with torch only)."""
    sys.path.insert(0, os.path.join(rh.REF, "tools"))
    from train_utils.optimization import fastai_optim, learning_schedules_fastai as lsf
    from functools import partial
    import torch.nn as nn
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(4, 6), nn.BatchNorm1d(6), nn.ReLU(), nn.Linear(6, 3))
    rng = np.random.default_rng(5)
    init = [rng.normal(size=tuple(p.shape)).astype(np.float32) for p in model.parameters()]
    grads = [[rng.normal(size=tuple(p.shape)).astype(np.float32) for p in model.parameters()] for _ in range(3)]
    with torch.no_grad():
        for p, v in zip(model.parameters(), init):
            p.copy_(torch.from_numpy(v))
    flatten = lambda m: sum(map(flatten, m.children()), []) if len(list(m.children())) else [m]
    opt = fastai_optim.OptimWrapper.create(partial(torch.optim.Adam, betas=(0.9, 0.99)), 3e-3,
                                           [nn.Sequential(*flatten(model))], wd=0.01, true_wd=True, bn_wd=True)
    sch = lsf.OneCycle(opt, 100, 0.003, [0.95, 0.85], 10, 0.4)
    lrs, moms = [], []
    for t in range(100):
        sch.step(t)
        lrs.append(opt.lr)
        moms.append(opt.mom)
    traj = []
    for t in range(3):
        sch.step(t)
        for p, g in zip(model.parameters(), grads[t]):
            p.grad = torch.from_numpy(g.copy())
        torch.nn.utils.clip_grad_norm_(model.parameters(), 10)
        opt.step()
        traj.append(np.concatenate([p.detach().numpy().ravel() for p in model.parameters()]))
    z = {"lr": np.array(lrs, dtype=np.float64), "mom": np.array(moms, dtype=np.float64),
         "init": np.concatenate([v.ravel() for v in init]), "traj": np.stack(traj),
         "grads": np.stack([np.concatenate([g.ravel() for g in gs]) for gs in grads]),
         "shapes": np.array([list(p.shape) + [0] * (2 - p.dim()) for p in model.parameters()], dtype=np.int64),
         "is_bn": np.array([0, 0, 1, 1, 0, 0], dtype=np.int64)}
    np.savez_compressed(os.path.join(HERE, "optimizer.npz"), **z)
    print("== optimizer golden written; lr[0,20,40,70,99] =", [lrs[i] for i in (0, 20, 40, 70, 99)])


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, c in CASES.items():
        if not only or name in only:
            run_case(name, c)
    if not only or "optimizer" in only:
        optimizer_golden()
