"""Model-builder subset of the reference ``Detector3DTemplate``
(pcdet/models/detectors/detector3d_template.py:14-100, 361-442): the ``module_topology`` walk with the
same keyword calls into the name->class registries, ``global_step``, and checkpoint loading by
key+shape.  Builders for modules outside the GD-MAE pre-training path return ``None`` unless their yaml
section is present, in which case they raise (nothing is silently skipped)."""
import os

import torch
import torch.nn as nn

from .. import backbones_2d, backbones_3d, dense_heads
from ..backbones_3d import vfe


class Detector3DTemplate(nn.Module):
    def __init__(self, model_cfg, num_class, dataset, logger):
        super().__init__()
        self.model_cfg, self.num_class, self.dataset, self.logger = model_cfg, num_class, dataset, logger
        self.class_names = dataset.class_names
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        self.module_topology = ['img_backbone', 'vfe', 'backbone_3d', 'map_to_bev_module', 'pfe', 'backbone_2d',
                                'dense_head', 'point_head', 'roi_head']
        self._cfg_key = {'img_backbone': 'IMG_BACKBONE', 'vfe': 'VFE', 'backbone_3d': 'BACKBONE_3D',
                         'map_to_bev_module': 'MAP_TO_BEV', 'pfe': 'PFE', 'backbone_2d': 'BACKBONE_2D',
                         'dense_head': 'DENSE_HEAD', 'point_head': 'POINT_HEAD', 'roi_head': 'ROI_HEAD'}

    @property
    def mode(self):
        return 'TRAIN' if self.training else 'TEST'

    def update_global_step(self):
        self.global_step += 1

    def build_networks(self):
        info = {'module_list': [], 'num_rawpoint_features': self.dataset.point_feature_encoder.num_point_features,
                'num_point_features': self.dataset.point_feature_encoder.num_point_features,
                'grid_size': self.dataset.grid_size, 'point_cloud_range': self.dataset.point_cloud_range,
                'voxel_size': self.dataset.voxel_size}
        for name in self.module_topology:
            builder = getattr(self, 'build_%s' % name, None)
            if builder is None:
                if self.model_cfg.get(self._cfg_key[name], None) is not None:
                    raise NotImplementedError(f"{self._cfg_key[name]} is outside the GD-MAE pre-training hot path")
                module = None
            else:
                module, info = builder(model_info_dict=info)
            self.add_module(name, module)
        return info['module_list']

    def build_vfe(self, model_info_dict):
        if self.model_cfg.get('VFE', None) is None:
            return None, model_info_dict
        m = vfe.__all__[self.model_cfg.VFE.NAME](
            model_cfg=self.model_cfg.VFE, num_point_features=model_info_dict['num_rawpoint_features'],
            point_cloud_range=model_info_dict['point_cloud_range'], voxel_size=model_info_dict['voxel_size'],
            grid_size=model_info_dict['grid_size'])
        model_info_dict['num_point_features'] = m.get_output_feature_dim()
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    def build_backbone_3d(self, model_info_dict):
        if self.model_cfg.get('BACKBONE_3D', None) is None:
            return None, model_info_dict
        m = backbones_3d.__all__[self.model_cfg.BACKBONE_3D.NAME](
            model_cfg=self.model_cfg.BACKBONE_3D, input_channels=model_info_dict['num_point_features'],
            grid_size=model_info_dict['grid_size'], voxel_size=model_info_dict['voxel_size'],
            point_cloud_range=model_info_dict['point_cloud_range'])
        model_info_dict['module_list'].append(m)
        model_info_dict['num_point_features'] = m.num_point_features
        model_info_dict['backbone_channels'] = getattr(m, 'backbone_channels', None)
        return m, model_info_dict

    def build_backbone_2d(self, model_info_dict):
        if self.model_cfg.get('BACKBONE_2D', None) is None:
            return None, model_info_dict
        m = backbones_2d.__all__[self.model_cfg.BACKBONE_2D.NAME](
            model_cfg=self.model_cfg.BACKBONE_2D, input_channels=model_info_dict.get('num_bev_features', None))
        model_info_dict['module_list'].append(m)
        model_info_dict['num_bev_features'] = m.num_bev_features
        return m, model_info_dict

    def build_dense_head(self, model_info_dict):
        if self.model_cfg.get('DENSE_HEAD', None) is None:
            return None, model_info_dict
        m = dense_heads.__all__[self.model_cfg.DENSE_HEAD.NAME](
            model_cfg=self.model_cfg.DENSE_HEAD, input_channels=model_info_dict['num_bev_features'],
            num_class=self.num_class if not self.model_cfg.DENSE_HEAD.CLASS_AGNOSTIC else 1, class_names=self.class_names,
            grid_size=model_info_dict['grid_size'], point_cloud_range=model_info_dict['point_cloud_range'],
            predict_boxes_when_training=self.model_cfg.get('ROI_HEAD', False), voxel_size=model_info_dict.get('voxel_size', False),
            backbone_channels=model_info_dict.get('backbone_channels', None))
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    @staticmethod
    def generate_recall_record(box_preds, recall_dict, batch_index, data_dict=None, thresh_list=None):
        """Recall bookkeeping of the evaluation loop (reference detector3d_template.py:318-358): ground-truth boxes matched by
        a predicted box with 3-D IoU above each threshold (``recall_rcnn_*``; ``recall_roi_*`` when RoIs are present)."""
        from ...ops.iou3d_nms import iou3d_nms_utils
        if 'gt_boxes' not in data_dict:
            return recall_dict
        rois = data_dict['rois'][batch_index] if 'rois' in data_dict else None
        gt = data_dict['gt_boxes'][batch_index]
        if len(recall_dict) == 0:
            recall_dict = {'gt_num': 0}
            for t in thresh_list:
                recall_dict['recall_roi_%s' % str(t)] = 0
                recall_dict['recall_rcnn_%s' % str(t)] = 0
        nz = (gt.abs().sum(dim=1) != 0).nonzero()
        gt = gt[:int(nz.max()) + 1] if nz.numel() else gt[:0]       # trailing all-zero rows are padding
        if gt.shape[0] > 0:
            iou_rcnn = iou3d_nms_utils.boxes_iou3d_gpu(box_preds[:, 0:7], gt[:, 0:7]) if box_preds.shape[0] > 0 else None
            iou_roi = iou3d_nms_utils.boxes_iou3d_gpu(rois[:, 0:7], gt[:, 0:7]) if rois is not None else None
            for t in thresh_list:
                if iou_rcnn is not None:
                    recall_dict['recall_rcnn_%s' % str(t)] += int((iou_rcnn.max(dim=0)[0] > t).sum().item())
                if iou_roi is not None:
                    recall_dict['recall_roi_%s' % str(t)] += int((iou_roi.max(dim=0)[0] > t).sum().item())
            recall_dict['gt_num'] += gt.shape[0]
        return recall_dict

    def forward(self, **kwargs):
        raise NotImplementedError

    # ---- checkpoints (reference-compatible dict: model_state / optimizer_state / epoch / it / version)
    def _load_state_dict(self, model_state_disk, *, strict=True):
        """Load by key + shape (reference detector3d_template.py:360-388).  Sparse-conv weights saved by another spconv
        generation are re-laid first: a tensor whose last two axes are swapped w.r.t. ours (spconv 1.x keeps (..., Cin, Cout))
        is transposed; failing that, the kernel-first layout (k.., Cin, Cout) is rotated to ours (Cout, k.., Cin) - the
        reference does this for 5-D (3-D conv) weights only and asserts otherwise, the 2-D convolutions of this path get the
        same treatment.  With ``strict`` the FILTERED dict is what ``load_state_dict`` sees, so keys of ours that the file
        does not cover raise, while foreign / mis-shaped keys of the file are ignored - exactly the reference's behaviour."""
        from ...utils.spconv_utils import find_all_spconv_keys
        state_dict = self.state_dict()
        sparse_keys = find_all_spconv_keys(self)
        update = {}
        for key, val in model_state_disk.items():
            want = state_dict.get(key, None)
            if want is None:
                continue
            if key in sparse_keys and want.shape != val.shape and val.dim() == want.dim():
                swapped = val.transpose(-1, -2)
                rotated = val.permute(val.dim() - 1, *range(val.dim() - 1))
                if swapped.shape == want.shape:
                    val = swapped.contiguous()
                elif rotated.shape == want.shape:
                    val = rotated.contiguous()
            if want.shape == val.shape:
                update[key] = val
        if strict:
            self.load_state_dict(update)
        else:
            state_dict.update(update)
            self.load_state_dict(state_dict)
        return state_dict, update

    def load_params_from_file(self, filename, logger, to_cpu=False):
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        ckpt = torch.load(filename, map_location=torch.device('cpu') if to_cpu else None)
        disk = ckpt['model_state']
        logger.info('==> Loading parameters from checkpoint %s (version %s)' % (filename, ckpt.get('version', None)))
        state_dict, update = self._load_state_dict(disk, strict=False)
        for k in state_dict:
            if k not in update:
                logger.info('Not updated weight %s: %s' % (k, str(state_dict[k].shape)))
        logger.info('==> Done (loaded %d/%d)' % (len(update), len(state_dict)))

    def load_params_with_optimizer(self, filename, to_cpu=False, optimizer=None, logger=None):
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        ckpt = torch.load(filename, map_location=torch.device('cpu') if to_cpu else None)
        self._load_state_dict(ckpt['model_state'], strict=True)
        if optimizer is not None and ckpt.get('optimizer_state', None) is not None:
            optimizer.load_state_dict(ckpt['optimizer_state'])
        return ckpt.get('it', 0.0), ckpt.get('epoch', -1)
