"""Rotated BEV IoU / NMS (SURVEY next row f4): the CPU oracle against an independent exact clipping, and the HIP kernels
(through the reference's Python API names) against the oracle; plus the evaluation path of the config-D detector."""
import logging

import numpy as np
import pytest
import torch

from oracle import iou3d_oracle as orc


def _boxes(rng, n, spread=6.0):
    b = np.zeros((n, 7), dtype=np.float32)
    b[:, 0:2] = rng.uniform(-spread, spread, (n, 2))
    b[:, 2] = rng.uniform(-1, 1, n)
    b[:, 3] = rng.uniform(1.0, 5.0, n)
    b[:, 4] = rng.uniform(0.5, 2.5, n)
    b[:, 5] = rng.uniform(1.0, 2.0, n)
    b[:, 6] = rng.uniform(-3.3, 3.3, n)
    return b


def test_oracle_overlap_agrees_with_exact_clipping():
    """The restated reference algorithm (edge crossings + corners inside with a 1e-2 margin) vs exact Sutherland-Hodgman
    clipping in float64: equal up to the margin's slivers (a touching corner counted as inside adds at most a triangle of
    ~margin x edge: <= 1.5e-2 m^2 on boxes of 0.5 .. 12.5 m^2, relative to the exact value <= 2 % where it exceeds 1 m^2);
    identical boxes give
    IoU 1, disjoint boxes 0, and containment the inner area."""
    rng = np.random.default_rng(1)
    A, B = _boxes(rng, 40), _boxes(rng, 40)
    pairs = [(float(orc.overlap(a, b)), orc.exact_overlap(a, b)) for a in A for b in B]
    assert max(abs(r - e) for r, e in pairs) <= 1.5e-2
    assert max(abs(r - e) / e for r, e in pairs if e > 1.0) <= 2e-2 and sum(e > 1.0 for _, e in pairs) > 50
    a = A[0]
    assert abs(float(orc.iou_bev(a, a)) - 1.0) < 1e-4
    far = a.copy()
    far[0] += 100
    assert float(orc.overlap(a, far)) == 0.0
    inner = a.copy()
    inner[3:5] *= 0.5
    assert abs(float(orc.overlap(a, inner)) - float(inner[3] * inner[4])) < 1e-3


# Known answers from plane geometry (boxes [x, y, z, dx, dy, dz, heading]; area in m^2) - they pin BOTH the oracle and the kernels
# independently of either implementation and of the reference's 1e-2 corner margin (no case has a corner within the margin of an
# edge of the other box, except the last two, whose exact value is the limit of the margin rule)
_S2 = float(np.sqrt(2.0))
KNOWN_OVERLAPS = [
    # axis-aligned partial overlap: [-2, 2] x [-1, 1] with the box shifted by (1, 0.5) -> 3 x 1.5
    ([0, 0, 0, 4, 2, 1, 0.0], [1, 0.5, 0, 4, 2, 1, 0.0], 4.5),
    # the same rectangle turned by 90 degrees about the common centre -> the 2 x 2 square
    ([0, 0, 0, 4, 2, 1, 0.0], [0, 0, 0, 4, 2, 1, np.pi / 2], 4.0),
    # a 2 x 2 square and its 45-degree turn: the regular octagon 8 (sqrt 2 - 1) a^2, a = 1
    ([3, -2, 0, 2, 2, 1, 0.0], [3, -2, 0, 2, 2, 1, np.pi / 4], 8.0 * (_S2 - 1.0)),
    # headings of pi and -pi / 2 are the same rectangles
    ([0, 0, 0, 4, 2, 1, np.pi], [1, 0.5, 0, 4, 2, 1, 0.0], 4.5),
    ([0, 0, 0, 4, 2, 1, 0.0], [0, 0, 0, 4, 2, 1, -np.pi / 2], 4.0),
    # a small square inside a large turned one -> the small area
    ([5, 5, 0, 6, 6, 1, 0.3], [5.2, 4.9, 0, 1, 1, 1, 1.1], 1.0),
    # a 45-degree square (half diagonal sqrt 2) whose corner reaches 0.5 into an axis-aligned one: right triangle, legs 0.5 sqrt 2 ... area 0.25
    ([0, 0, 0, 2, 2, 1, 0.0], [1 + _S2 - 0.5, 0, 0, 2, 2, 1, np.pi / 4], 0.25),
    # disjoint by 5 cm, and by far
    ([0, 0, 0, 2, 2, 1, 0.0], [2.05, 0, 0, 2, 2, 1, 0.0], 0.0),
    ([0, 0, 0, 2, 2, 1, 0.7], [40, -30, 0, 2, 2, 1, 0.2], 0.0),
]


def test_oracle_overlap_known_answers():
    """The restated reference algorithm against closed-form intersection areas (fp32: 1e-4 absolute on areas of 0.25 .. 4.5 m^2)."""
    for a, b, area in KNOWN_OVERLAPS:
        a, b = np.asarray(a, dtype=np.float32), np.asarray(b, dtype=np.float32)
        for p, q in ((a, b), (b, a)):
            got = float(orc.overlap(p, q))
            assert abs(got - area) <= 1e-4, (p.tolist(), q.tolist(), got, area)
            assert abs(orc.exact_overlap(p, q) - area) <= 1e-6
        if area > 0:
            union = float(a[3] * a[4] + b[3] * b[4]) - area
            assert abs(float(orc.iou_bev(a, b)) - area / union) <= 1e-5


@pytest.mark.gpu
def test_hip_bev_overlap_known_answers():
    """boxes_iou_bev / boxes_iou3d_gpu (HIP kernels behind the reference's API names) against the same closed forms."""
    from pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    dev = torch.device("cuda:0")
    A = torch.tensor([k[0] for k in KNOWN_OVERLAPS], dtype=torch.float32, device=dev)
    B = torch.tensor([k[1] for k in KNOWN_OVERLAPS], dtype=torch.float32, device=dev)
    area = np.array([k[2] for k in KNOWN_OVERLAPS])
    iou = U.boxes_iou_bev(A, B).cpu().numpy()
    Aa, Ba = (A[:, 3] * A[:, 4]).cpu().numpy(), (B[:, 3] * B[:, 4]).cpu().numpy()
    want = area / np.maximum(Aa + Ba - area, 1e-8)
    assert np.abs(np.diag(iou) - want).max() <= 1e-5, (np.diag(iou), want)
    assert np.abs(np.diag(U.boxes_iou_bev(B, A).cpu().numpy()) - want).max() <= 1e-5
    # all boxes have z = 0, dz = 1: the 3-D IoU is area x 1 over the summed volumes minus it
    iou3 = U.boxes_iou3d_gpu(A, B).cpu().numpy()
    assert np.abs(np.diag(iou3) - want).max() <= 1e-5


@pytest.mark.gpu
def test_hip_bev_iou_and_nms_match_oracle():
    """boxes_iou_bev / boxes_iou3d_gpu / nms_gpu / nms_normal_gpu (reference API names, HIP kernels) vs the CPU oracle."""
    from pcdet.ops.iou3d_nms import iou3d_nms_utils as U
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(2)
    A, B = _boxes(rng, 60), _boxes(rng, 45)
    iou = U.boxes_iou_bev(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)).cpu().numpy()
    ref = orc.pairs(A, B, 1)
    assert np.abs(iou - ref).max() <= 2e-5, np.abs(iou - ref).max()
    assert (ref > 0.05).sum() > 50
    # 3-D IoU: BEV overlap x height overlap
    ov = orc.pairs(A, B, 0)
    ha = np.minimum((A[:, 2] + A[:, 5] / 2)[:, None], (B[:, 2] + B[:, 5] / 2)[None]) - np.maximum((A[:, 2] - A[:, 5] / 2)[:, None], (B[:, 2] - B[:, 5] / 2)[None])
    o3 = ov * np.clip(ha, 0, None)
    ref3 = o3 / np.clip((A[:, 3] * A[:, 4] * A[:, 5])[:, None] + (B[:, 3] * B[:, 4] * B[:, 5])[None] - o3, 1e-6, None)
    iou3 = U.boxes_iou3d_gpu(torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)).cpu().numpy()
    assert np.abs(iou3 - ref3).max() <= 2e-5
    # NMS: crowded boxes (many overlaps), 300 boxes = 5 mask words; thresholds away from ties
    C = _boxes(rng, 300, spread=8.0)
    scores = rng.permutation(300).astype(np.float32)
    order = np.argsort(-scores)
    for thresh, rotated, fn in ((0.1, True, U.nms_gpu), (0.4, True, U.nms_gpu), (0.2, False, U.nms_normal_gpu)):
        keep, _ = fn(torch.from_numpy(C).to(dev), torch.from_numpy(scores).to(dev), thresh)
        ref_keep = order[orc.nms(C[order], thresh, rotated)]
        assert np.array_equal(keep.cpu().numpy(), ref_keep), (thresh, rotated, len(ref_keep), keep.numel())
        assert 10 < len(ref_keep) < 300
    keep, _ = U.nms_gpu(torch.from_numpy(C).to(dev), torch.from_numpy(scores).to(dev), 0.1, pre_maxsize=64)
    assert np.array_equal(keep.cpu().numpy(), order[:64][orc.nms(C[order[:64]], 0.1, True)])
    e, _ = U.nms_gpu(torch.zeros(0, 7, device=dev), torch.zeros(0, device=dev), 0.1)
    assert e.numel() == 0


@pytest.mark.gpu
def test_config_d_eval_post_processing():
    """Evaluation path of the config-D detector (CenterHead.generate_predicted_boxes -> decode + rotated NMS ->
    CenterPoint.post_processing -> recall record): planted heat-map peaks come back as boxes at the planted places with
    the planted sizes / headings, duplicates are suppressed, and ground truth equal to the planted boxes is fully recalled."""
    from gdmae_hip import configs
    from pcdet.models import build_network
    dev = torch.device("cuda:0")
    cfg, ds, _ = configs.named_config("D")
    net = build_network(cfg, 3, ds, logging.getLogger("t")).to(dev).eval()
    head = net.dense_head
    B, H, W = 2, int(ds.grid_size[1]), int(ds.grid_size[0])
    rng = np.random.default_rng(3)
    pd = {"hm": torch.full((B, 3, H, W), -8.0, device=dev), "center": torch.zeros(B, 2, H, W, device=dev),
          "center_z": torch.zeros(B, 1, H, W, device=dev), "dim": torch.zeros(B, 3, H, W, device=dev),
          "rot": torch.zeros(B, 2, H, W, device=dev)}
    planted = []
    for b in range(B):
        rows = []
        for k in range(6):
            y, x, c = int(rng.integers(40, H - 40)), int(rng.integers(40, W - 40)), int(rng.integers(0, 3))
            ang = float(rng.uniform(-3, 3))
            dims = np.array([3.9, 1.6, 1.5]) * rng.uniform(0.9, 1.1, 3)
            off = rng.uniform(0.1, 0.9, 2)
            pd["hm"][b, c, y, x] = 4.0 + 0.1 * k
            pd["hm"][b, c, y, x + 1] = 3.0                           # a weaker neighbour: same box again -> suppressed by NMS
            for (yy, xx) in ((y, x), (y, x + 1)):
                pd["center"][b, :, yy, xx] = torch.tensor(off if xx == x else off - np.array([1.0, 0.0]), dtype=torch.float32)
                pd["center_z"][b, 0, yy, xx] = -0.5
                pd["dim"][b, :, yy, xx] = torch.tensor(np.log(dims), dtype=torch.float32)
                pd["rot"][b, :, yy, xx] = torch.tensor([np.cos(ang), np.sin(ang)], dtype=torch.float32)
            rows.append([(x + off[0]) * 0.16 + ds.point_cloud_range[0], (y + off[1]) * 0.16 + ds.point_cloud_range[1], -0.5, *dims, ang, c + 1])
        planted.append(np.array(rows, dtype=np.float32))
    out = head.generate_predicted_boxes(B, [pd])
    gt = torch.zeros(B, 10, 8, device=dev)
    for b in range(B):
        pb, ps, pl = out[b]["pred_boxes"].cpu().numpy(), out[b]["pred_scores"].cpu().numpy(), out[b]["pred_labels"].cpu().numpy()
        assert pb.shape == (6, 7) and np.all(np.diff(ps) <= 0), (pb.shape, ps)
        ref = planted[b][np.argsort(-np.arange(6))]                 # planted scores increase with k
        assert np.abs(pb[:, :6] - ref[:, :6]).max() < 1e-3 and np.array_equal(pl, ref[:, 7].astype(np.int64))
        assert np.abs(np.angle(np.exp(1j * (pb[:, 6] - ref[:, 6])))).max() < 1e-4
        gt[b, :6] = torch.from_numpy(planted[b]).to(dev)
    final, recall = net.post_processing({"final_box_dicts": out, "batch_size": B, "gt_boxes": gt})
    assert recall["gt_num"] == 12 and recall["recall_rcnn_0.7"] == 12 and recall["recall_roi_0.3"] == 0
