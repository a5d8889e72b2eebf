"""``class_agnostic_nms`` with the reference's signature (pcdet/models/model_utils/model_nms_utils.py:6-27)."""
import torch

from ...ops.iou3d_nms import iou3d_nms_utils


def class_agnostic_nms(box_scores, box_preds, nms_config, score_thresh=None):
    src_box_scores = box_scores
    if score_thresh is not None:
        scores_mask = box_scores >= score_thresh
        box_scores, box_preds = box_scores[scores_mask], box_preds[scores_mask]
    selected = []
    if box_scores.shape[0] > 0:
        top_scores, indices = torch.topk(box_scores, k=min(nms_config.NMS_PRE_MAXSIZE, box_scores.shape[0]))
        keep_idx, _ = getattr(iou3d_nms_utils, nms_config.NMS_TYPE)(box_preds[indices][:, 0:7], top_scores, nms_config.NMS_THRESH,
                                                                   **nms_config)
        selected = indices[keep_idx[:nms_config.NMS_POST_MAXSIZE]]
    if score_thresh is not None:
        selected = scores_mask.nonzero().view(-1)[selected]
    return selected, src_box_scores[selected]
