"""Launch-count switches of the encoder / conv-block backward and of the stage forward leave every bit where it was: the reductions that
ride along with the launch that follows them (GDMAE_LAYER_TAIL_RIDES, GDMAE_DW_REDUCE_RIDES: DESIGN section 9) are the same sums in the same
order as the launches of their own, and the in-projection carried by the previous layer's closing launch (GDMAE_QKV_RIDES=1) produces the
q / k / v rows of k_tok_gemm_multi.  The switches are read once per process, so every setting is a subprocess that runs one bench-mode train
step (flat optimizer, bf16 autocast, fused layers) on a golden case's inputs and prints a digest of the loss and of the whole flat gradient."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STEP = r'''
import hashlib, logging, os, sys
REPO = sys.argv[1]
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd"), os.path.join(REPO, "tests")]
import numpy as np, torch
from helpers import load_case
from oracle import gdmae_oracle as orc          # test infrastructure: seeded parameters only
from gdmae_hip import configs, optim
from pcdet.models import build_network
dev = torch.device("cuda:0")
z, ds, cfg, shapes = load_case(sys.argv[2])
torch.manual_seed(0)
net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev)
net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
net.train()
opt = optim.FlatAdamOneCycle(net, configs.optimization_cfg(8), total_steps=10)
opt.zero_grad()
bd = {"points": torch.from_numpy(z["points"]).to(dev), "batch_size": int(z["batch_size"]), "mae_noise": torch.from_numpy(z["noise"]).to(dev)}
with torch.autocast("cuda", dtype=torch.bfloat16):
    ret, _, _ = net(bd)
ret["loss"].backward()
torch.cuda.synchronize()
g = opt.flat_grad.detach().cpu().numpy()
assert np.isfinite(g).all() and float(np.abs(g).sum()) > 0
if len(sys.argv) > 3:
    np.save(sys.argv[3], g)
print("DIGEST", float(ret["loss"]).hex(), hashlib.sha256(g.tobytes()).hexdigest())
'''


def _digest(case, env_extra, save=None):
    env = dict(os.environ)
    for k in ("GDMAE_LAYER_TAIL_RIDES", "GDMAE_DW_REDUCE_RIDES", "GDMAE_QKV_RIDES", "GDMAE_DW_PAIR", "GDMAE_LAYER_V3"):
        env.pop(k, None)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", STEP, REPO, case] + ([save] if save else []), env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")]
    assert len(lines) == 1, out.stdout[-2000:]
    return lines[0]


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["waymo_b1"])
def test_ride_along_switches_are_bit_identical(case):
    base = _digest(case, {})
    assert _digest(case, {}) == base, "the step itself is not bit-repeatable"
    for env in ({"GDMAE_LAYER_TAIL_RIDES": "0"}, {"GDMAE_DW_REDUCE_RIDES": "0"}, {"GDMAE_QKV_RIDES": "1"},
                {"GDMAE_LAYER_TAIL_RIDES": "0", "GDMAE_DW_REDUCE_RIDES": "0", "GDMAE_QKV_RIDES": "1"}):
        assert _digest(case, env) == base, env


@pytest.mark.gpu
def test_pair_tile_weight_gradients_match_the_four_wavefront_kernel(tmp_path):
    """GDMAE_DW_PAIR=1 (dw_grouped.hip k_dw_grouped2, opt-in: 128 x 256 pair tiles, eight wavefronts): the partial TILES are the four-wavefront
    kernel's bit for bit (same fragments, same k order) - every weight gradient equal -, the bias column sums of a layer launch are taken
    by 32 instead of 16 row groups, i.e. in another fixed order: equal to fp32 round-off."""
    import numpy as np
    a, b = str(tmp_path / "base.npy"), str(tmp_path / "pair.npy")
    da, db = _digest("waymo_b1", {}, a), _digest("waymo_b1", {"GDMAE_DW_PAIR": "1"}, b)
    assert da.split()[1] == db.split()[1], "the loss does not depend on the weight-gradient kernel"
    ga, gb = np.load(a), np.load(b)
    diff = np.flatnonzero(ga != gb)
    assert diff.size <= 0.002 * ga.size, (diff.size, ga.size)                 # only bias entries (column sums) may differ
    assert np.allclose(ga, gb, rtol=2e-5, atol=1e-6 * float(np.abs(ga).max()))


@pytest.mark.gpu
def test_plan_issued_under_the_decoder_convolution():
    """SPTBackboneMAE.prefetch_plan_under_decoder: the next batch's plan is issued from inside the current forward (right before the tile
    convolution), finish() returns the plan prefetch_plan() builds, and a forward on it gives the loss of the inline plan, bit for bit."""
    import logging
    sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd"), os.path.join(REPO, "tests")]
    import torch
    from helpers import load_case
    from oracle import gdmae_oracle as orc
    from gdmae_hip import decoder as gdec
    from pcdet.models import build_network
    dev = torch.device("cuda:0")
    z, ds, cfg, shapes = load_case("waymo_b1")
    torch.manual_seed(0)
    net = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev)
    net.load_state_dict(orc.seeded_state_dict(shapes, seed=int(z["seed"])), strict=False)
    net.train()
    pts, B = torch.from_numpy(z["points"]).to(dev), int(z["batch_size"])
    noise = torch.from_numpy(z["noise"]).to(dev)
    noise_cap = torch.cat([noise, torch.rand(pts.shape[0] - noise.numel(), device=dev)])
    bb = net.backbone_3d
    with torch.autocast("cuda", dtype=torch.bfloat16):
        ref, _, _ = net({"points": pts, "batch_size": B, "mae_noise": noise})
        pf = bb.prefetch_plan_under_decoder(pts, B, noise=noise_cap)
        assert bb._pre_conv_hook is not None and not hasattr(gdec, "PRE_CONV_HOOK")      # on the module: no process-global hook
        issued = []
        real = bb.prefetch_plan
        bb.prefetch_plan = lambda *a, **k: (issued.append(1), real(*a, **k))[1]
        net({"points": pts, "batch_size": B, "mae_noise": noise})            # the forward that carries the issue
        assert bb._pre_conv_hook is None and issued == [1], "the forward did not issue the plan"
        bb.prefetch_plan = real
        vox, plan = pf.finish()
        out, _, _ = net({"points": pts, "batch_size": B, "_gdmae_vox": vox, "_gdmae_plan": plan})
    assert float(out["loss"]) == float(ref["loss"])
    # a forward that never reaches the tile convolution: finish() issues the plan itself
    pf2 = bb.prefetch_plan_under_decoder(pts, B, noise=noise_cap)
    vox2, plan2 = pf2.finish()
    assert bb._pre_conv_hook is None
    # a second model's forward does not fire this model's pending hook
    pf3 = bb.prefetch_plan_under_decoder(pts, B, noise=noise_cap)
    net2 = build_network(cfg, len(ds.class_names), ds, logging.getLogger("t")).to(dev).train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        net2({"points": pts, "batch_size": B, "mae_noise": noise})
    assert bb._pre_conv_hook is not None
    pf3.finish()
    assert bb._pre_conv_hook is None
    assert torch.equal(vox2.pillar_cell, vox.pillar_cell) and torch.equal(plan2.mask, plan.mask)


@pytest.mark.gpu
def test_in_register_layer_forward_matches_the_row_tile_kernels(tmp_path):
    """GDMAE_LAYER_V3=1 (csrc/layer_v3.hip, opt-in experiment: one wavefront carries 16 token rows through the whole layer in registers,
    16 x 16 x 32 matrix-core products chained through their accumulators, weights as one LDS-DMA stream): same tensors and rounding points
    as k_layer_fwd, another summation order inside the products - the step's loss and flat gradient agree to bf16 round-off."""
    import numpy as np
    a, b = str(tmp_path / "base.npy"), str(tmp_path / "v3.npy")
    da, db = _digest("waymo_b1", {}, a), _digest("waymo_b1", {"GDMAE_LAYER_V3": "1"}, b)
    la, lb = float.fromhex(da.split()[1]), float.fromhex(db.split()[1])
    assert abs(la - lb) <= 1e-3 * abs(la), (la, lb)
    ga, gb = np.load(a).astype(np.float64), np.load(b).astype(np.float64)
    cos = float(ga @ gb / (np.linalg.norm(ga) * np.linalg.norm(gb)))
    assert cos > 0.995 and abs(np.linalg.norm(gb) / np.linalg.norm(ga) - 1) < 0.02, (cos, np.linalg.norm(ga), np.linalg.norm(gb))
