"""f2 (next row): input pipeline - oracle vs the reference golden (CPU), HIP pipeline vs the golden (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import input_oracle as io

HERE = os.path.dirname(os.path.abspath(__file__))


def _case():
    z = np.load(os.path.join(HERE, "golden", "input_pipeline.npz"))
    nf = int(z["n_frames"])
    frames = [z[f"frame{i}"] for i in range(nf)]
    perms = [z[f"perm{i}"].astype(np.int64) for i in range(nf)]
    params = [{"flip_x": bool(z["flip_x"][i]), "flip_y": bool(z["flip_y"][i]), "angle": float(z["angle"][i]),
               "scale": float(z["scale"][i])} for i in range(nf)]
    return z, frames, params, perms


def test_input_oracle_matches_reference_golden():
    z, frames, params, perms = _case()
    got, kept = io.pipeline(frames, params, z["pc_range"], perms)
    exp = z["expected"]
    assert got.shape == exp.shape and kept == [len(p) for p in perms]
    assert np.array_equal(got[:, 0], exp[:, 0]) and np.array_equal(got[:, 4:], exp[:, 4:])
    assert np.abs(got - exp).max() <= 1e-5                    # BLAS vs numpy matmul rounding of the rotation


def test_world_parameters_are_drawn_like_the_reference():
    """Same np.random calls in the same order as DataAugmentor (the golden stores what the reference drew, seed 123)."""
    from gdmae_hip import input_pipeline as ip
    z, frames, params, perms = _case()
    np.random.seed(123)
    for pr, perm in zip(params, perms):
        mine = ip.draw_world_params(ip.SSL_AUG_CONFIG)
        assert mine == pr
        assert np.array_equal(np.random.permutation(len(perm)), perm)     # the reference's shuffle draw in between


@pytest.mark.gpu
def test_hip_input_pipeline_matches_reference_golden():
    from gdmae_hip import input_pipeline as ip
    z, frames, params, perms = _case()
    pipe = ip.GpuInputPipeline(z["pc_range"])
    got = pipe(frames, params=params, perms=perms).cpu().numpy()
    exp = z["expected"]
    assert got.shape == exp.shape
    assert np.array_equal(got[:, 0], exp[:, 0]) and np.array_equal(got[:, 4:], exp[:, 4:])
    assert np.abs(got - exp).max() <= 2e-5
    # unshuffled output = the oracle without permutation, bit-exact in the untouched columns
    plain = ip.GpuInputPipeline(z["pc_range"], shuffle=False)(frames, params=params).cpu().numpy()
    ref_plain, kept = io.pipeline(frames, params, z["pc_range"], None)
    assert plain.shape == ref_plain.shape and np.abs(plain - ref_plain).max() <= 2e-5
    # default mode: a random permutation within every frame of the same rows
    torch.manual_seed(0)
    shuf = ip.GpuInputPipeline(z["pc_range"])(frames, params=params).cpu().numpy()
    assert shuf.shape == plain.shape and not np.array_equal(shuf, plain)
    assert np.array_equal(shuf[:, 0], plain[:, 0])                          # frames stay contiguous and ordered
    key = lambda a: a[np.lexsort(a.T[::-1])]                                # noqa: E731
    assert np.array_equal(key(shuf), key(plain))


@pytest.mark.gpu
def test_hip_input_pipeline_edge_cases():
    from gdmae_hip import input_pipeline as ip
    rng = np.array([-10, -10, -2, 10, 10, 4], np.float32)
    ident = {"flip_x": False, "flip_y": False, "angle": 0.0, "scale": 1.0}
    empty = np.zeros((0, 5), np.float32)
    allout = np.full((7, 5), 50.0, np.float32)
    edge = np.array([[10.0, -10.0, 0, 1, 2], [10.000001, 0, 0, 3, 4], [0, 0, 0, 5, 6]], np.float32)   # closed range
    pipe = ip.GpuInputPipeline(rng, shuffle=False)
    out = pipe([empty, allout, edge, empty], params=[ident] * 4).cpu().numpy()
    assert out.shape == (2, 6) and np.array_equal(out[:, 0], [2, 2]) and np.array_equal(out[:, 1:], edge[[0, 2]])
    assert pipe([empty], params=[ident]).shape == (0, 6)
