"""Rotated BEV IoU / 3-D IoU / NMS with the reference's Python API (pcdet/ops/iou3d_nms/iou3d_nms_utils.py:31-133), on
libgdmae_hip.so (csrc/iou3d_nms.hip) instead of the ``iou3d_nms_cuda`` extension.  Unlike the reference's ``nms_gpu`` the
suppression masks are scanned on the device: no D2H copy, ``keep`` comes back as a device tensor."""
import torch

from gdmae_hip import lib as L


def _pairs(boxes_a, boxes_b, mode):
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7 and boxes_a.is_cuda and boxes_b.is_cuda
    a, b = boxes_a.float().contiguous(), boxes_b.float().contiguous()
    out = torch.zeros(a.shape[0], b.shape[0], dtype=torch.float32, device=a.device)
    L.call("gdmae_boxes_bev_pairs", L.ptr(a), a.shape[0], L.ptr(b), b.shape[0], mode, L.ptr(out), L.stream())
    return out


def boxes_iou_bev(boxes_a, boxes_b):
    """(N, 7), (M, 7) [x, y, z, dx, dy, dz, heading] -> (N, M) rotated BEV IoU."""
    return _pairs(boxes_a, boxes_b, 1)


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """(N, 7), (M, 7) -> (N, M) 3-D IoU = rotated BEV overlap x height overlap / union volume."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    a_max, a_min = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1), (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max, b_min = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1), (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = _pairs(boxes_a, boxes_b, 0) * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def _nms(boxes, scores, thresh, rotated, pre_maxsize=None):
    assert boxes.shape[1] == 7 and boxes.is_cuda
    order = scores.sort(0, descending=True)[1]
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    b = boxes[order].float().contiguous()
    n = b.shape[0]
    keep = torch.empty(max(n, 1), dtype=torch.int64, device=b.device)
    n_keep = torch.zeros(1, dtype=torch.int32, device=b.device)
    ws = torch.empty(L.load().gdmae_nms_workspace_bytes(n), dtype=torch.uint8, device=b.device)
    L.call("gdmae_nms_bev", L.ptr(b), n, float(thresh), int(rotated), L.ptr(keep), L.ptr(n_keep), L.ptr(ws), L.stream())
    return order[keep[:int(n_keep.item())]].contiguous(), None


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """Rotated NMS: indices (into ``boxes``) of the kept boxes in descending score order."""
    return _nms(boxes, scores, thresh, True, pre_maxsize)


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """NMS on the axis-aligned BEV footprints (heading ignored)."""
    return _nms(boxes, scores, thresh, False)
