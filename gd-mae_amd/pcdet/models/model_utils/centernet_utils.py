"""Box decoding from the CenterHead maps (reference pcdet/models/model_utils/centernet_utils.py:120-260: ``_topk``,
``_transpose_and_gather_feat``, ``decode_bbox_from_heatmap``): per sample the K best heat-map cells over all classes, the
regression maps gathered there, boxes in metric coordinates, filtered by the post-centre range and the score threshold.
Plain tensor ops (a top-k and a handful of gathers per batch); circle-NMS is asserted out by the reference itself."""
import torch


def _gather_feat(feat, ind):
    return feat.gather(1, ind.unsqueeze(2).expand(ind.size(0), ind.size(1), feat.size(2)))


def _transpose_and_gather_feat(feat, ind):
    feat = feat.permute(0, 2, 3, 1).contiguous()
    return _gather_feat(feat.view(feat.size(0), -1, feat.size(3)), ind)


def _topk(scores, K=40):
    batch, num_class, height, width = scores.size()
    topk_scores, topk_inds = torch.topk(scores.flatten(2, 3), K)
    topk_inds = topk_inds % (height * width)
    topk_ys = torch.div(topk_inds, width, rounding_mode='floor').float()
    topk_xs = (topk_inds % width).int().float()
    topk_score, topk_ind = torch.topk(topk_scores.view(batch, -1), K)
    topk_classes = torch.div(topk_ind, K, rounding_mode='floor').int()
    pick = lambda t: _gather_feat(t.view(batch, -1, 1), topk_ind).view(batch, K)   # noqa: E731
    return topk_score, pick(topk_inds), topk_classes, pick(topk_ys), pick(topk_xs)


def decode_bbox_from_heatmap(heatmap, rot_cos, rot_sin, center, center_z, dim, vel=None, iou=None, point_cloud_range=None,
                             voxel_size=None, feature_map_stride=None, K=100, circle_nms=False, score_thresh=None,
                             post_center_limit_range=None):
    assert not circle_nms, "circle_nms is marked 'not checked yet' (assert False) in the reference"
    batch_size = heatmap.size(0)
    scores, inds, class_ids, ys, xs = _topk(heatmap, K=K)
    g = lambda t, c: _transpose_and_gather_feat(t, inds).view(batch_size, K, c)     # noqa: E731
    ious, center, rot_sin, rot_cos, center_z, dim = g(iou, 1), g(center, 2), g(rot_sin, 1), g(rot_cos, 1), g(center_z, 1), g(dim, 3)
    angle = torch.atan2(rot_sin, rot_cos)
    xs = (xs.view(batch_size, K, 1) + center[:, :, 0:1]) * feature_map_stride * voxel_size[0] + point_cloud_range[0]
    ys = (ys.view(batch_size, K, 1) + center[:, :, 1:2]) * feature_map_stride * voxel_size[1] + point_cloud_range[1]
    parts = [xs, ys, center_z, dim, angle]
    if vel is not None:
        parts.append(g(vel, 2))
    boxes = torch.cat(parts, dim=-1)
    scores, ious, class_ids = scores.view(batch_size, K), ious.view(batch_size, K), class_ids.view(batch_size, K)
    assert post_center_limit_range is not None
    mask = (boxes[..., :3] >= post_center_limit_range[:3]).all(2) & (boxes[..., :3] <= post_center_limit_range[3:]).all(2)
    if score_thresh is not None:
        mask &= scores > score_thresh
    return [{'pred_boxes': boxes[k, mask[k]], 'pred_scores': scores[k, mask[k]], 'pred_ious': ious[k, mask[k]],
             'pred_labels': class_ids[k, mask[k]]} for k in range(batch_size)]
