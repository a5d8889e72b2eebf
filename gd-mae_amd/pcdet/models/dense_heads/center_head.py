"""CenterPoint-style dense head of the GD-MAE fine-tune detector (training path).

Module tree, parameter names, ``forward_ret_dict`` layout, loss definition and loss weights follow the reference
``CenterHead`` / ``SeparateHead`` (pcdet/models/dense_heads/center_head.py:11-392) and its losses
(pcdet/utils/loss_utils.py:273-396: CornerNet focal loss on the clamped sigmoid heat map, masked L1 on the gathered
regression maps).  What differs is where the work happens: the reference assigns targets with a Python loop over boxes on
the CPU (numpy Gaussian patches, a D2H + H2D round trip per sample); here ``assign_targets`` is two launches of
libgdmae_hip.so per head (``gdmae_center_head_targets``), no host transfer.

Evaluation: ``generate_predicted_boxes`` (center_head.py:279-342) decodes the K best heat-map cells
(model_utils/centernet_utils.py) and runs the rotated NMS of csrc/iou3d_nms.hip through ``model_nms_utils``."""
import copy

import torch
import torch.nn as nn
from torch.nn.init import kaiming_normal_

from gdmae_hip import lib as L
from ..model_utils import centernet_utils, model_nms_utils


class SeparateHead(nn.Module):
    def __init__(self, input_channels, sep_head_dict, init_bias=-2.19, use_bias=False):
        super().__init__()
        self.sep_head_dict = sep_head_dict
        for name, spec in sep_head_dict.items():
            layers = []
            for _ in range(spec['num_conv'] - 1):
                layers.append(nn.Sequential(nn.Conv2d(input_channels, input_channels, 3, stride=1, padding=1, bias=use_bias),
                                            nn.BatchNorm2d(input_channels), nn.ReLU()))
            layers.append(nn.Conv2d(input_channels, spec['out_channels'], 3, stride=1, padding=1, bias=True))
            fc = nn.Sequential(*layers)
            if 'hm' in name:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        kaiming_normal_(m.weight.data)
                        if m.bias is not None:
                            nn.init.constant_(m.bias, 0)
            setattr(self, name, fc)

    def forward(self, x):
        return {name: getattr(self, name)(x) for name in self.sep_head_dict}


def focal_loss_centernet(pred, gt):
    """CornerNet focal loss (loss_utils.py:273-312, no mask): pred = clamped sigmoid, gt = Gaussian heat map."""
    pos = gt.eq(1).float()
    neg = gt.lt(1).float()
    neg_w = torch.pow(1 - gt, 4)
    pos_loss = (torch.log(pred) * torch.pow(1 - pred, 2) * pos).sum()
    neg_loss = (torch.log(1 - pred) * torch.pow(pred, 2) * neg_w * neg).sum()
    num_pos = pos.sum()
    return torch.where(num_pos == 0, -neg_loss, -(pos_loss + neg_loss) / num_pos.clamp(min=1))


def reg_loss_centernet(output, mask, ind, target):
    """Masked L1 on the regression maps gathered at the object cells (loss_utils.py:325-396): (dim,) per-code losses."""
    B, C = output.shape[0], output.shape[1]
    feat = output.permute(0, 2, 3, 1).reshape(B, -1, C)
    pred = feat.gather(1, ind.unsqueeze(2).expand(B, ind.shape[1], C))
    num = mask.float().sum()
    m = mask.unsqueeze(2).expand_as(target).float() * (~torch.isnan(target)).float()
    loss = torch.abs(pred * m - target * m).sum(dim=(0, 1))
    return loss / torch.clamp_min(num, 1.0)


class CenterHead(nn.Module):
    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, voxel_size,
                 predict_boxes_when_training=True, **kwargs):
        super().__init__()
        self.model_cfg, self.num_class, self.grid_size = model_cfg, num_class, grid_size
        self.point_cloud_range, self.voxel_size = point_cloud_range, voxel_size
        self.feature_map_stride = model_cfg.TARGET_ASSIGNER_CONFIG.get('FEATURE_MAP_STRIDE', None)
        self.class_names = list(class_names)
        self.class_names_each_head = [[x for x in names if x in class_names] for names in model_cfg.CLASS_NAMES_EACH_HEAD]
        assert sum(len(x) for x in self.class_names_each_head) == len(self.class_names), self.class_names_each_head
        # global class id (1-based, 0 = padding) -> 1-based id inside each head (0: the class belongs to another head)
        self._class_maps = [[0] + [names.index(c) + 1 if c in names else 0 for c in self.class_names]
                            for names in self.class_names_each_head]
        self._class_map_dev = {}
        use_bias = model_cfg.get('USE_BIAS_BEFORE_NORM', False)
        c = model_cfg.SHARED_CONV_CHANNEL
        self.shared_conv = nn.Sequential(nn.Conv2d(input_channels, c, 3, stride=1, padding=1, bias=use_bias), nn.BatchNorm2d(c), nn.ReLU())
        self.separate_head_cfg = model_cfg.SEPARATE_HEAD_CFG
        self.heads_list = nn.ModuleList()
        for names in self.class_names_each_head:
            hd = copy.deepcopy(dict(self.separate_head_cfg.HEAD_DICT))
            hd['hm'] = dict(out_channels=len(names), num_conv=model_cfg.NUM_HM_CONV)
            self.heads_list.append(SeparateHead(c, hd, init_bias=-2.19, use_bias=use_bias))
        self.with_iou = 'iou' in self.separate_head_cfg.HEAD_DICT
        if self.with_iou:
            raise NotImplementedError("IoU-aware head (IoULossCenterNet) is not part of the shipped GD-MAE configs")
        self.predict_boxes_when_training = predict_boxes_when_training
        self.forward_ret_dict = {}

    # ---- targets (HIP)
    def assign_targets(self, gt_boxes, feature_map_size=None, **kwargs):
        """gt_boxes (B, n_max, 8+) on the device, class in the last column; feature_map_size [H, W].
        -> {'heatmaps', 'target_boxes', 'inds', 'masks'}: one tensor per head, as the reference returns."""
        cfg = self.model_cfg.TARGET_ASSIGNER_CONFIG
        fh, fw = int(feature_map_size[0]), int(feature_map_size[1])
        gt = gt_boxes.float().contiguous()
        assert gt.is_cuda, "CenterHead.assign_targets runs in libgdmae_hip.so (no CPU fallback)"
        B, n_max, box_dim = gt.shape
        dev = gt.device
        K = int(cfg.NUM_MAX_OBJS)
        ret = {'heatmaps': [], 'target_boxes': [], 'inds': [], 'masks': []}
        ws = torch.empty(L.load().gdmae_center_head_targets_workspace_bytes(B, K), dtype=torch.uint8, device=dev)
        for h, names in enumerate(self.class_names_each_head):
            key = (h, dev.index)
            if key not in self._class_map_dev:
                self._class_map_dev[key] = torch.tensor(self._class_maps[h], dtype=torch.int32, device=dev)
            hm = torch.empty(B, len(names), fh, fw, dtype=torch.float32, device=dev)
            tb = torch.empty(B, K, box_dim, dtype=torch.float32, device=dev)
            inds = torch.empty(B, K, dtype=torch.int64, device=dev)
            mask = torch.empty(B, K, dtype=torch.int64, device=dev)
            L.call("gdmae_center_head_targets", L.ptr(gt), B, n_max, box_dim, L.ptr(self._class_map_dev[key]), len(self.class_names),
                   len(names), L.host_f32([self.point_cloud_range[0], self.point_cloud_range[1]]),
                   L.host_f32([self.voxel_size[0], self.voxel_size[1]]), float(cfg.FEATURE_MAP_STRIDE), fw, fh, K,
                   float(cfg.GAUSSIAN_OVERLAP), int(cfg.MIN_RADIUS), L.ptr(hm), L.ptr(tb), L.ptr(inds), L.ptr(mask), L.ptr(ws), L.stream())
            ret['heatmaps'].append(hm), ret['target_boxes'].append(tb), ret['inds'].append(inds), ret['masks'].append(mask)
        return ret

    @staticmethod
    def sigmoid(x):
        return torch.clamp(x.sigmoid(), min=1e-4, max=1 - 1e-4)

    def get_loss(self):
        pred_dicts, targets = self.forward_ret_dict['pred_dicts'], self.forward_ret_dict['target_dicts']
        w = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        tb_dict, loss = {}, 0
        for idx, pd in enumerate(pred_dicts):
            pd['hm'] = self.sigmoid(pd['hm'].float())
            hm_loss = focal_loss_centernet(pd['hm'], targets['heatmaps'][idx]) * w['cls_weight']
            pred_boxes = torch.cat([pd[name].float() for name in self.separate_head_cfg.HEAD_ORDER], dim=1)
            reg = reg_loss_centernet(pred_boxes, targets['masks'][idx], targets['inds'][idx], targets['target_boxes'][idx])
            loc_loss = (reg * reg.new_tensor(w['code_weights'])).sum() * w['loc_weight']
            loss = loss + hm_loss + loc_loss
            tb_dict['hm_loss_head_%d' % idx] = hm_loss.detach()
            tb_dict['loc_loss_head_%d' % idx] = loc_loss.detach()
        return loss, tb_dict

    def generate_predicted_boxes(self, batch_size, pred_dicts):
        cfg = self.model_cfg.POST_PROCESSING
        dev = pred_dicts[0]['hm'].device
        limit = torch.tensor(cfg.POST_CENTER_LIMIT_RANGE, dtype=torch.float32, device=dev)
        ret = [{'pred_boxes': [], 'pred_scores': [], 'pred_labels': []} for _ in range(batch_size)]
        for idx, pd in enumerate(pred_dicts):
            hm = pd['hm'].float().sigmoid()
            vel = pd['vel'].float() if 'vel' in self.separate_head_cfg.HEAD_ORDER else None
            finals = centernet_utils.decode_bbox_from_heatmap(
                heatmap=hm, rot_cos=pd['rot'][:, 0:1].float(), rot_sin=pd['rot'][:, 1:2].float(), center=pd['center'].float(),
                center_z=pd['center_z'].float(), dim=pd['dim'].float().exp(), vel=vel, iou=torch.ones_like(hm[:, 0:1]),
                point_cloud_range=self.point_cloud_range, voxel_size=self.voxel_size, feature_map_stride=self.feature_map_stride,
                K=cfg.MAX_OBJ_PER_SAMPLE, circle_nms=(cfg.NMS_CONFIG.NMS_TYPE == 'circle_nms'), score_thresh=cfg.SCORE_THRESH,
                post_center_limit_range=limit)
            cmap = torch.tensor([self.class_names.index(c) for c in self.class_names_each_head[idx]], dtype=torch.int64, device=dev)
            for k, fd in enumerate(finals):
                labels = cmap[fd['pred_labels'].long()]
                if cfg.NMS_CONFIG.NMS_TYPE != 'nms_gpu':
                    raise NotImplementedError(cfg.NMS_CONFIG.NMS_TYPE)
                selected, selected_scores = model_nms_utils.class_agnostic_nms(box_scores=fd['pred_scores'], box_preds=fd['pred_boxes'],
                                                                               nms_config=cfg.NMS_CONFIG, score_thresh=None)
                ret[k]['pred_boxes'].append(fd['pred_boxes'][selected])
                ret[k]['pred_scores'].append(selected_scores)
                ret[k]['pred_labels'].append(labels[selected])
        for k in range(batch_size):
            ret[k]['pred_boxes'] = torch.cat(ret[k]['pred_boxes'], dim=0)
            ret[k]['pred_scores'] = torch.cat(ret[k]['pred_scores'], dim=0)
            ret[k]['pred_labels'] = torch.cat(ret[k]['pred_labels'], dim=0) + 1
        return ret

    def forward(self, data_dict):
        x = self.shared_conv(data_dict['spatial_features_2d'])
        pred_dicts = [head(x) for head in self.heads_list]
        if self.training:
            self.forward_ret_dict['target_dicts'] = self.assign_targets(data_dict['gt_boxes'], feature_map_size=x.shape[2:])
        self.forward_ret_dict['pred_dicts'] = pred_dicts
        if not self.training or self.predict_boxes_when_training:
            data_dict['final_box_dicts'] = self.generate_predicted_boxes(data_dict['batch_size'], pred_dicts)
        return data_dict
