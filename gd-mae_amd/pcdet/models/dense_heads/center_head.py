"""CenterPoint-style dense head of the GD-MAE fine-tune detector (training path).

Module tree, parameter names, ``forward_ret_dict`` layout, loss definition and loss weights follow the reference
``CenterHead`` / ``SeparateHead`` (pcdet/models/dense_heads/center_head.py:11-392) and its losses
(pcdet/utils/loss_utils.py:273-396: CornerNet focal loss on the clamped sigmoid heat map, masked L1 on the gathered
regression maps).  What differs is where the work happens: the reference assigns targets with a Python loop over boxes on
the CPU (numpy Gaussian patches, a D2H + H2D round trip per sample); here ``assign_targets`` is two launches of
libgdmae_hip.so per head (``gdmae_center_head_targets``), no host transfer.

IoU-aware variant (``iou`` entry in HEAD_DICT, tools/cfgs/waymo_models/gd_mae_iou.yaml:228-254): an extra regression map
trained with L1 against 2 * IoU3D(decoded box, ground truth) - 1 at the object cells (loss_utils.py:398-419,
center_head.py:258-275); at evaluation the scores are rectified as score^(1-a) * iou^a per class before a per-class rotated
NMS (``multi_class_nms``; model_nms_utils.py:28-46).

Evaluation / RoI path: ``generate_predicted_boxes`` (center_head.py:279-342) takes the K best cells of the flattened
(class, y, x) heat map, decodes them in ONE launch (``gdmae_center_head_decode``: the reference's
``decode_bbox_from_heatmap`` gathers every map through transposed copies) and runs the on-device rotated NMS of
csrc/iou3d_nms.hip."""
import copy

import torch
import torch.nn as nn
from torch.nn.init import kaiming_normal_

from gdmae_hip import dense as gdense
from gdmae_hip import ops as gops
from gdmae_hip import lib as L
from ...ops.iou3d_nms import iou3d_nms_utils


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
        out = {}
        # one map, one branch per regression / heat-map head: the branches' input gradients meet in one pass (ops.FanOut)
        xs = gops.FanOut.apply(x, len(self.sep_head_dict)) if (gdense.FUSE_SHORTCUT and x.is_cuda and x.requires_grad and torch.is_grad_enabled()) else None
        for h, name in enumerate(self.sep_head_dict):
            y = x if xs is None else xs[h]
            for layer in getattr(self, name):
                y = gdense.conv_bn_relu(layer, y) if isinstance(layer, nn.Sequential) else gdense.conv3x3(layer, y)
            out[name] = y
        return out


def focal_loss_centernet(pred, gt):
    """CornerNet focal loss (loss_utils.py:273-312, no mask): pred = clamped sigmoid, gt = Gaussian heat map."""
    pos = gt.eq(1).float()
    neg = gt.lt(1).float()
    neg_w = torch.pow(1 - gt, 4)
    pos_loss = (torch.log(pred) * torch.pow(1 - pred, 2) * pos).sum()
    neg_loss = (torch.log(1 - pred) * torch.pow(pred, 2) * neg_w * neg).sum()
    num_pos = pos.sum()
    return torch.where(num_pos == 0, -neg_loss, -(pos_loss + neg_loss) / num_pos.clamp(min=1))


class FocalLossCenterNetFn(torch.autograd.Function):
    """focal_loss_centernet(clamp(sigmoid(logits)), gt) as one launch per direction (csrc/center_head.hip gdmae_focal_loss_fwd / _bwd):
    the logits are read where the head's convolution left them (any strides, bf16 or fp32), the backward writes a channels-last
    gradient in the logits' dtype.  -> (loss, clamped sigmoid (B, C, H, W) fp32 - what the reference keeps in pred_dict['hm'])."""

    @staticmethod
    def forward(ctx, logits, gt):
        B, C, H, W = logits.shape
        assert logits.is_cuda and logits.dtype in (torch.bfloat16, torch.float32) and gt.shape == logits.shape
        gt = gt.float().contiguous()
        dev = logits.device
        st = logits.stride()
        strides = L.host_i64([st[0], st[2], st[3], st[1]])
        base = logits                                   # the view's own data pointer (storage offset included)
        prob = torch.empty(B, C, H, W, dtype=torch.float32, device=dev)
        part = torch.empty(L.load().gdmae_focal_loss_rows() * 3, dtype=torch.float32, device=dev)
        out4 = torch.empty(4, dtype=torch.float32, device=dev)
        L.call("gdmae_focal_loss_fwd", base.data_ptr(), int(logits.dtype == torch.bfloat16), strides, L.ptr(gt), B, C, H, W, L.ptr(prob),
               L.ptr(part), L.ptr(out4), L.stream())
        ctx.save_for_backward(logits, gt, out4)
        ctx.mark_non_differentiable(prob)
        return out4[0], prob

    @staticmethod
    def backward(ctx, g, _gp):
        logits, gt, out4 = ctx.saved_tensors
        B, C, H, W = logits.shape
        st = logits.stride()
        strides = L.host_i64([st[0], st[2], st[3], st[1]])
        dx = torch.empty(B, H, W, C, dtype=logits.dtype, device=logits.device)
        gs = g.float().reshape(1).contiguous()
        L.call("gdmae_focal_loss_bwd", logits.data_ptr(), int(logits.dtype == torch.bfloat16), strides, L.ptr(gt), B, C, H, W, L.ptr(out4),
               L.ptr(gs), L.ptr(dx), int(dx.dtype == torch.bfloat16), L.stream())
        return dx.permute(0, 3, 1, 2), None


def reg_loss_centernet_maps(maps, mask, ind, target):
    """reg_loss_centernet on the regression maps of a head WITHOUT concatenating them: every map is gathered at the object cells first
    ((B, K, c_i) each, K = NUM_MAX_OBJS), the gathered columns are concatenated - the reference concatenates the (B, 8, H, W) maps and
    gathers then (center_head.py:240-245, loss_utils.py:325-396): same values, 55 MB of fp32 copies less per direction."""
    B = maps[0].shape[0]
    cols = []
    for m in maps:
        c = m.shape[1]
        rows = m.permute(0, 2, 3, 1).reshape(B, -1, c)                # a view for a column slice of a channels-last map
        cols.append(rows.gather(1, ind.unsqueeze(2).expand(B, ind.shape[1], c)).float())
    pred = torch.cat(cols, dim=2)
    num = mask.float().sum()
    m = mask.unsqueeze(2).expand_as(target).float() * (~torch.isnan(target)).float()
    loss = torch.abs(pred * m - target * m).sum(dim=(0, 1))
    return loss / torch.clamp_min(num, 1.0)


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
        ret = {'heatmaps': [], 'target_boxes': [], 'iou_boxes': [], 'inds': [], 'masks': []}
        ws = torch.empty(L.load().gdmae_center_head_targets_workspace_bytes(B, K), dtype=torch.uint8, device=dev)
        for h, names in enumerate(self.class_names_each_head):
            key = (h, dev.index)
            if key not in self._class_map_dev:
                self._class_map_dev[key] = torch.tensor(self._class_maps[h], dtype=torch.int32, device=dev)
            hm = torch.empty(B, len(names), fh, fw, dtype=torch.float32, device=dev)
            tb = torch.empty(B, K, box_dim, dtype=torch.float32, device=dev)
            inds = torch.empty(B, K, dtype=torch.int64, device=dev)
            mask = torch.empty(B, K, dtype=torch.int64, device=dev)
            ib = torch.empty(B, K, 7, dtype=torch.float32, device=dev)
            L.call("gdmae_center_head_targets_iou", L.ptr(gt), B, n_max, box_dim, L.ptr(self._class_map_dev[key]), len(self.class_names),
                   len(names), L.host_f32([self.point_cloud_range[0], self.point_cloud_range[1]]),
                   L.host_f32([self.voxel_size[0], self.voxel_size[1]]), float(cfg.FEATURE_MAP_STRIDE), fw, fh, K,
                   float(cfg.GAUSSIAN_OVERLAP), int(cfg.MIN_RADIUS), L.ptr(hm), L.ptr(tb), L.ptr(ib), L.ptr(inds), L.ptr(mask), L.ptr(ws),
                   L.stream())
            ret['heatmaps'].append(hm), ret['target_boxes'].append(tb), ret['inds'].append(inds), ret['masks'].append(mask)
            ret['iou_boxes'].append(ib)
        return ret

    @staticmethod
    def sigmoid(x):
        return torch.clamp(x.sigmoid(), min=1e-4, max=1 - 1e-4)

    def get_loss(self):
        pred_dicts, targets = self.forward_ret_dict['pred_dicts'], self.forward_ret_dict['target_dicts']
        w = self.model_cfg.LOSS_CONFIG.LOSS_WEIGHTS
        tb_dict, loss = {}, 0
        for idx, pd in enumerate(pred_dicts):
            if gdense.FUSE_SHORTCUT and pd['hm'].is_cuda and pd['hm'].dtype in (torch.bfloat16, torch.float32):
                # one launch per direction instead of the ~20 elementwise passes of sigmoid / clamp / focal loss; the regression maps
                # gathered at the object cells before they are concatenated
                hm_loss, pd['hm'] = FocalLossCenterNetFn.apply(pd['hm'], targets['heatmaps'][idx])
                hm_loss = hm_loss * w['cls_weight']
                reg = reg_loss_centernet_maps([pd[name] for name in self.separate_head_cfg.HEAD_ORDER], targets['masks'][idx],
                                              targets['inds'][idx], targets['target_boxes'][idx])
            else:
                pd['hm'] = self.sigmoid(pd['hm'].float())
                hm_loss = focal_loss_centernet(pd['hm'], targets['heatmaps'][idx]) * w['cls_weight']
                pred_boxes = torch.cat([pd[name].float() for name in self.separate_head_cfg.HEAD_ORDER], dim=1)
                reg = reg_loss_centernet(pred_boxes, targets['masks'][idx], targets['inds'][idx], targets['target_boxes'][idx])
            loc_loss = (reg * reg.new_tensor(w['code_weights'])).sum() * w['loc_weight']
            loss = loss + hm_loss + loc_loss
            tb_dict['hm_loss_head_%d' % idx] = hm_loss.detach()
            tb_dict['loc_loss_head_%d' % idx] = loc_loss.detach()
            if self.with_iou:
                iou_loss = self.iou_loss(pd, targets['masks'][idx], targets['inds'][idx], targets['iou_boxes'][idx]) * w['iou_weight']
                loss = loss + iou_loss
                tb_dict['iou_loss_head_%d' % idx] = iou_loss.detach()
        return loss, tb_dict

    def iou_loss(self, pd, mask, ind, box_gt):
        """L1 between the ``iou`` map at the object cells and 2 * IoU3D(decoded box there, its ground truth) - 1, summed and
        divided by (number of objects + 1e-4) (loss_utils.py:398-419 on the boxes of center_head.py:258-272).  Only the
        object cells are decoded (the reference decodes the full (B, 7, H, W) map first); the IoU target carries no gradient."""
        B, _, H, W = pd['dim'].shape
        sel = mask.bool()
        b_idx = torch.arange(B, device=ind.device).view(B, 1).expand_as(ind)[sel]
        cell = ind[sel]

        def at(name):
            m = pd[name].float()
            return m.permute(0, 2, 3, 1).reshape(B, H * W, m.shape[1])[b_idx, cell]
        pred = at('iou')
        with torch.no_grad():
            ctr, cz, dim, rot = at('center'), at('center_z'), at('dim').exp(), at('rot')
            xs = ((cell % W).to(ctr.dtype) + ctr[:, 0]) * self.feature_map_stride * self.voxel_size[0] + self.point_cloud_range[0]
            ys = (torch.div(cell, W, rounding_mode='floor').to(ctr.dtype) + ctr[:, 1]) * self.feature_map_stride * self.voxel_size[1] \
                + self.point_cloud_range[1]
            boxes = torch.cat([xs[:, None], ys[:, None], cz, dim, torch.atan2(rot[:, 1:2], rot[:, 0:1])], dim=1)
            gt = box_gt[sel]
            # IoU of the i-th decoded box with the i-th ground-truth box: the diagonal of the pair matrix
            target = torch.diagonal(iou3d_nms_utils.boxes_iou3d_gpu(boxes, gt)).unsqueeze(-1) if boxes.shape[0] else pred.detach()
            target = 2 * target - 1
        return torch.abs(pred - target).sum() / (sel.sum() + 1e-4)

    def generate_predicted_boxes(self, batch_size, pred_dicts):
        cfg = self.model_cfg.POST_PROCESSING
        nms = cfg.NMS_CONFIG
        if nms.NMS_TYPE not in ('nms_gpu', 'multi_class_nms'):
            raise NotImplementedError(f"NMS_TYPE {nms.NMS_TYPE} (the reference asserts circle_nms out as 'not checked yet')")
        K = int(cfg.MAX_OBJ_PER_SAMPLE)
        per_sample = [([], [], []) for _ in range(batch_size)]
        for idx, pd in enumerate(pred_dicts):
            hm = pd['hm'].float().sigmoid()
            B, C, H, W = hm.shape
            dev = hm.device
            f = lambda name: pd[name].float().contiguous()   # noqa: E731
            vel = f('vel') if 'vel' in self.separate_head_cfg.HEAD_ORDER else None
            iou = f('iou') if 'iou' in pd else None
            k = min(K, C * H * W)
            score, cell = torch.topk(hm.reshape(B, -1), k)             # best cells over all classes, descending
            boxes = torch.empty(B, k, 9 if vel is not None else 7, dtype=torch.float32, device=dev)
            labels = torch.empty(B, k, dtype=torch.int32, device=dev)
            ious = torch.empty(B, k, dtype=torch.float32, device=dev)
            valid = torch.empty(B, k, dtype=torch.uint8, device=dev)
            L.call("gdmae_center_head_decode", L.ptr(cell.contiguous()), L.ptr(score.contiguous()), L.ptr(f('center')), L.ptr(f('center_z')),
                   L.ptr(f('dim')), L.ptr(f('rot')), L.ptr(vel), L.ptr(iou), B, k, H, W,
                   L.host_f32([self.point_cloud_range[0], self.point_cloud_range[1]]), L.host_f32([self.voxel_size[0], self.voxel_size[1]]),
                   float(self.feature_map_stride), L.host_f32(list(cfg.POST_CENTER_LIMIT_RANGE)),
                   float(cfg.SCORE_THRESH if cfg.SCORE_THRESH is not None else 0.0), int(cfg.SCORE_THRESH is not None),
                   L.ptr(boxes), L.ptr(labels), L.ptr(ious), L.ptr(valid), L.stream())
            to_global = torch.tensor([self.class_names.index(c) for c in self.class_names_each_head[idx]], dtype=torch.int64, device=dev)
            for b in range(batch_size):
                ok = valid[b].bool()
                bx, sc, lab, q = boxes[b][ok], score[b][ok], to_global[labels[b][ok].long()], ious[b][ok]
                if nms.NMS_TYPE == 'nms_gpu':
                    keep = self._nms_one(bx, sc, nms.NMS_THRESH, nms.NMS_PRE_MAXSIZE, nms.NMS_POST_MAXSIZE)
                else:
                    # IoU-rectified scores, then one rotated NMS per class (model_nms_utils.py:28-46)
                    a = sc.new_tensor(list(nms.IOU_RECTIFIER))[lab]
                    sc = torch.pow(sc, 1 - a) * torch.pow(q, a)
                    keeps = []
                    for c in range(len(nms.NMS_THRESH)):
                        own = (lab == c).nonzero(as_tuple=True)[0]
                        if own.numel():
                            keeps.append(own[self._nms_one(bx[own], sc[own], nms.NMS_THRESH[c], nms.NMS_PRE_MAXSIZE[c],
                                                           nms.NMS_POST_MAXSIZE[c])])
                    keep = torch.cat(keeps) if keeps else torch.zeros(0, dtype=torch.int64, device=dev)
                per_sample[b][0].append(bx[keep]), per_sample[b][1].append(sc[keep]), per_sample[b][2].append(lab[keep] + 1)
        return [{'pred_boxes': torch.cat(p[0]), 'pred_scores': torch.cat(p[1]), 'pred_labels': torch.cat(p[2])} for p in per_sample]

    @staticmethod
    def _nms_one(boxes, scores, thresh, pre_max, post_max):
        """Indices (descending score) surviving the rotated NMS of the ``pre_max`` best boxes, at most ``post_max`` of them."""
        if boxes.shape[0] == 0:
            return torch.zeros(0, dtype=torch.int64, device=boxes.device)
        keep, _ = iou3d_nms_utils.nms_gpu(boxes[:, :7], scores, thresh, pre_maxsize=int(pre_max))
        return keep[:int(post_max)]

    @staticmethod
    def reorder_rois_for_refining(batch_size, pred_dicts):
        """Per-sample predictions -> zero-padded (B, n_max, .) RoI tensors for a second stage (center_head.py:344-360; at least
        one padded row so an empty batch keeps its shape)."""
        n = max(1, max(int(d['pred_boxes'].shape[0]) for d in pred_dicts))
        ref = pred_dicts[0]['pred_boxes']
        rois = ref.new_zeros(batch_size, n, ref.shape[-1])
        scores = ref.new_zeros(batch_size, n)
        labels = torch.zeros(batch_size, n, dtype=torch.int64, device=ref.device)
        for b, d in enumerate(pred_dicts[:batch_size]):
            m = d['pred_boxes'].shape[0]
            rois[b, :m], scores[b, :m], labels[b, :m] = d['pred_boxes'], d['pred_scores'], d['pred_labels']
        return rois, scores, labels

    def forward(self, data_dict):
        x = gdense.conv_bn_relu(self.shared_conv, data_dict['spatial_features_2d'])
        pred_dicts = [head(x) for head in self.heads_list]
        if self.training:
            self.forward_ret_dict['target_dicts'] = self.assign_targets(data_dict['gt_boxes'], feature_map_size=x.shape[2:])
        self.forward_ret_dict['pred_dicts'] = pred_dicts
        if not self.training or self.predict_boxes_when_training:
            boxes = self.generate_predicted_boxes(data_dict['batch_size'], pred_dicts)
            data_dict['cls_preds_normalized'] = True
            if self.predict_boxes_when_training:      # a RoI head follows (two-stage configs): padded RoI tensors
                data_dict['rois'], data_dict['roi_scores'], data_dict['roi_labels'] = self.reorder_rois_for_refining(
                    data_dict['batch_size'], boxes)
                data_dict['has_class_labels'] = True
            else:
                data_dict['final_box_dicts'] = boxes
        return data_dict
