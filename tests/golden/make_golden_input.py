"""Golden vectors for the input pipeline (f2): runs the REFERENCE's own augmentation / masking / shuffle / collate
functions (imported from /root/reference through ref_harness) on seeded synthetic frames and stores inputs, the random
decisions the reference drew, and its collated output.  Run in the build container:  python tests/golden/make_golden_input.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path[:0] = [REPO, os.path.join(REPO, "gd-mae_amd"), HERE]

import types  # noqa: E402

import ref_harness  # noqa: E402
from oracle import input_oracle as io  # noqa: E402

ref_harness.install()
# the three reference modules used here import optional packages / sibling modules that are irrelevant to the functions
# under test (cv2, the GT database sampler, the file client, the voxel generator wrapper): empty stand-ins
for name in ("pcdet.datasets.augmentor", "pcdet.datasets.processor"):
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(ref_harness.REF, *name.split("."))]
    sys.modules[name] = m
sys.modules.setdefault("cv2", types.ModuleType("cv2"))
for name, attrs in (("pcdet.datasets.augmentor.database_sampler", ()), ("pcdet.utils.file_client", ()),
                    ("pcdet.datasets.processor.data_processor", ("DataProcessor",)),
                    ("pcdet.datasets.processor.point_feature_encoder", ("PointFeatureEncoder",))):
    m = types.ModuleType(name)
    for a in attrs:
        setattr(m, a, type(a, (), {}))
    sys.modules[name] = m
    setattr(sys.modules[name.rsplit(".", 1)[0]], name.rsplit(".", 1)[1], m)
DataAugmentor = ref_harness.ref("pcdet.datasets.augmentor.data_augmentor").DataAugmentor
common_utils = ref_harness.ref("pcdet.utils.common_utils")
DatasetTemplate = ref_harness.ref("pcdet.datasets.dataset").DatasetTemplate

PC_RANGE = np.array([-74.88, -74.88, -2, 74.88, 74.88, 4.0], np.float32)
CFG = [
    {"NAME": "random_world_flip", "PROBABILITY": 0.5, "ALONG_AXIS_LIST": ["x", "y"]},
    {"NAME": "random_world_rotation", "PROBABILITY": 1.0, "WORLD_ROT_ANGLE": [-0.78539816, 0.78539816]},
    {"NAME": "random_world_scaling", "PROBABILITY": 1.0, "WORLD_SCALE_RANGE": [0.95, 1.05]},
]


def main():
    g = np.random.default_rng(7)
    frames = []
    for n in (3000, 2500, 1, 3500):
        xyz = g.uniform(-90, 90, (n, 3)).astype(np.float32)      # ~30 % of the points fall outside the xy range
        xyz[:, 2] = g.uniform(-2, 4, n)
        frames.append(np.concatenate([xyz, g.uniform(0, 1, (n, 2)).astype(np.float32)], 1))
    np.random.seed(123)
    state0 = np.random.get_state()
    samples, params, perms = [], [], []
    for f in frames:
        d = {"points": f.copy(), "transformation_3d_list": [], "transformation_3d_params": {}}
        d = DataAugmentor.random_world_flip(None, d, CFG[0])
        d = DataAugmentor.random_world_rotation(None, d, CFG[1])
        d = DataAugmentor.random_world_scaling(None, d, CFG[2])
        t = d["transformation_3d_params"]
        params.append({"flip_x": "x" in t["random_world_flip"], "flip_y": "y" in t["random_world_flip"],
                       "angle": float(t["random_world_rotation"]), "scale": float(t["random_world_scaling"])})
        pts = d["points"]
        pts = pts[common_utils.mask_points_by_range(pts, PC_RANGE)]
        perm = np.random.permutation(pts.shape[0])
        perms.append(perm)
        samples.append({"points": pts[perm]})
    expected = DatasetTemplate.collate_batch(samples)["points"].astype(np.float32)
    # our restatement with the recorded decisions must reproduce the reference
    got, kept = io.pipeline(frames, params, PC_RANGE, perms)
    assert got.shape == expected.shape, (got.shape, expected.shape)
    err = np.abs(got - expected).max()
    assert err <= 1e-5, err
    # and our re-draw of the decisions from the same np.random state must agree with what the reference drew
    from gdmae_hip import input_pipeline as ip
    np.random.set_state(state0)
    for f, pr, perm in zip(frames, params, perms):
        mine = ip.draw_world_params(CFG)
        assert mine["flip_x"] == pr["flip_x"] and mine["flip_y"] == pr["flip_y"]
        assert mine["angle"] == pr["angle"] and mine["scale"] == pr["scale"], (mine, pr)
        assert np.array_equal(np.random.permutation(len(perm)), perm)
    np.savez_compressed(os.path.join(HERE, "input_pipeline.npz"), pc_range=PC_RANGE, n_frames=len(frames),
                        **{f"frame{i}": f for i, f in enumerate(frames)}, **{f"perm{i}": p.astype(np.int32) for i, p in enumerate(perms)},
                        flip_x=np.array([p["flip_x"] for p in params]), flip_y=np.array([p["flip_y"] for p in params]),
                        angle=np.array([p["angle"] for p in params], np.float64), scale=np.array([p["scale"] for p in params], np.float64),
                        expected=expected, oracle_err=err)
    print("wrote input_pipeline.npz: frames", [f.shape[0] for f in frames], "kept", kept, "max |oracle - reference| =", err)


if __name__ == "__main__":
    main()
