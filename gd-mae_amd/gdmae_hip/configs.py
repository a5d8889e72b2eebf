"""Programmatic builders for the model/data configs this path is measured on.

The MODEL section produced by ``gdmae_ssl_model_cfg()`` equals the ``MODEL`` section of the
reference's ``tools/cfgs/{waymo,kitti,once}_models/gd_mae_ssl.yaml`` (lines 46-175; the three
files share it) - asserted by ``tests/golden/make_golden.py`` in the build container.  The named
configs A/B/E are the BASELINE.json workloads as made concrete in SURVEY.md §8(d).
"""
from __future__ import annotations

import numpy as np

from pcdet.config import AttrDict


def _drop_info():
    lv = {'0': {'max_tokens': 16, 'drop_range': [0, 16]},
          '1': {'max_tokens': 32, 'drop_range': [16, 32]},
          '2': {'max_tokens': 64, 'drop_range': [32, 100000]}}
    return {'train': lv, 'test': {k: dict(v) for k, v in lv.items()}}


def _sst_block(name, stride, d_model, ff, num_blocks=2, nhead=8):
    return {
        'NAME': name,
        'PREPROCESS': {'WINDOW_SHAPE': [8, 8, 1], 'DROP_INFO': _drop_info(), 'SHUFFLE_VOXELS': False,
                       'POS_TEMPERATURE': 1000, 'NORMALIZE_POS': False},
        'ENCODER': {'NUM_BLOCKS': num_blocks, 'STRIDE': stride, 'D_MODEL': d_model, 'NHEAD': nhead,
                    'DIM_FEEDFORWARD': ff, 'DROPOUT': 0.0, 'ACTIVATION': 'gelu',
                    'LAYER_CFG': {'cosine': True, 'tau_min': 0.01}},
    }


def gdmae_ssl_model_cfg(mask_ratio=0.85, d_models=(128, 256, 256), ffs=(256, 512, 512), num_blocks=2,
                        vfe_mlps=(64, 128), eval_metric='waymo_custom'):
    strides = [1, 2, 2][:len(d_models)]
    names = ['sst_block_x1', 'sst_block_x2', 'sst_block_x4'][:len(d_models)]
    blocks = [_sst_block(n, s, d, f, num_blocks) for n, s, d, f in zip(names, strides, d_models, ffs)]
    srcs = ['x_conv1', 'x_conv2', 'x_conv3'][:len(d_models)]
    fuse = {src: {'UPSAMPLE_STRIDE': int(np.prod(strides[:i + 1])), 'NUM_FILTER': d_models[i], 'NUM_UPSAMPLE_FILTER': 128}
            for i, src in enumerate(srcs)}
    return AttrDict({
        'NAME': 'GDMAE',
        'VFE': {'NAME': 'DynVFE', 'TYPE': 'mean', 'WITH_DISTANCE': False, 'USE_ABSLOTE_XYZ': True,
                'USE_CLUSTER_XYZ': True, 'MLPS': [list(vfe_mlps)]},
        'BACKBONE_3D': {'NAME': 'SPTBackboneMAE', 'SST_BLOCK_LIST': blocks,
                        'MASK_CONFIG': {'RATIO': mask_ratio, 'NUM_PRD_POINTS': 16, 'NUM_GT_POINTS': 64},
                        'FEATURES_SOURCE': srcs, 'FUSE_LAYER': fuse},
        'POST_PROCESSING': {'RECALL_THRESH_LIST': [0.3, 0.5, 0.7], 'EVAL_METRIC': eval_metric},
    })


def gdmae_finetune_backbone_cfg(**kw):
    """BACKBONE_3D section of the fine-tune configs (tools/cfgs/{kitti,waymo}_models/gd_mae.yaml:81-133): SPTBackbone with
    the same SST block list / decoder sources as the pre-training backbone, no masking."""
    b = gdmae_ssl_model_cfg(**kw).BACKBONE_3D
    return AttrDict({'NAME': 'SPTBackbone', 'SST_BLOCK_LIST': b.SST_BLOCK_LIST, 'FEATURES_SOURCE': b.FEATURES_SOURCE,
                     'FUSE_LAYER': b.FUSE_LAYER})


def sst_bev_backbone_cfg():
    """BACKBONE_2D section of tools/cfgs/waymo_models/gd_mae.yaml:205-213."""
    k = lambda dil: {'out_channels': 128, 'kernel_size': 3, 'dilation': dil, 'padding': dil, 'stride': 1}   # noqa: E731
    return AttrDict({'NAME': 'SSTBEVBackbone', 'NUM_FILTER': 128, 'CONV_KWARGS': [k(1), k(1), k(2), k(1)], 'CONV_SHORTCUT': [0, 1, 2]})


def center_head_cfg(class_names=('Vehicle', 'Pedestrian', 'Cyclist'), post_range=(-75.2, -75.2, -2, 75.2, 75.2, 4)):
    """DENSE_HEAD section of tools/cfgs/waymo_models/gd_mae.yaml:215-258 (CenterHead)."""
    return AttrDict({
        'NAME': 'CenterHead', 'CLASS_AGNOSTIC': False, 'CLASS_NAMES_EACH_HEAD': [list(class_names)],
        'SHARED_CONV_CHANNEL': 64, 'USE_BIAS_BEFORE_NORM': True, 'NUM_HM_CONV': 2,
        'SEPARATE_HEAD_CFG': {'HEAD_ORDER': ['center', 'center_z', 'dim', 'rot'],
                              'HEAD_DICT': {'center': {'out_channels': 2, 'num_conv': 2}, 'center_z': {'out_channels': 1, 'num_conv': 2},
                                            'dim': {'out_channels': 3, 'num_conv': 2}, 'rot': {'out_channels': 2, 'num_conv': 2}}},
        'TARGET_ASSIGNER_CONFIG': {'FEATURE_MAP_STRIDE': 1, 'NUM_MAX_OBJS': 500, 'GAUSSIAN_OVERLAP': 0.1, 'MIN_RADIUS': 2},
        'LOSS_CONFIG': {'LOSS_WEIGHTS': {'cls_weight': 1.0, 'loc_weight': 2.0, 'code_weights': [1.0] * 8}},
        'POST_PROCESSING': {'SCORE_THRESH': 0.1, 'POST_CENTER_LIMIT_RANGE': list(post_range), 'MAX_OBJ_PER_SAMPLE': 500,
                            'NMS_CONFIG': {'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.7, 'NMS_PRE_MAXSIZE': 4096, 'NMS_POST_MAXSIZE': 500}},
    })


def center_head_iou_cfg(class_names=('Vehicle', 'Pedestrian', 'Cyclist'), post_range=(-75.2, -75.2, -2, 75.2, 75.2, 4)):
    """DENSE_HEAD section of tools/cfgs/waymo_models/gd_mae_iou.yaml:215-254: CenterHead with the extra ``iou`` regression map,
    ``iou_weight`` and the IoU-rectified per-class NMS."""
    c = center_head_cfg(class_names, post_range)
    c.SEPARATE_HEAD_CFG.HEAD_DICT['iou'] = AttrDict({'out_channels': 1, 'num_conv': 2})
    c.LOSS_CONFIG.LOSS_WEIGHTS['iou_weight'] = 1.0
    c.POST_PROCESSING.NMS_CONFIG = AttrDict({'NMS_TYPE': 'multi_class_nms', 'NMS_THRESH': [0.8, 0.55, 0.55],
                                             'NMS_PRE_MAXSIZE': [2048, 1024, 1024], 'NMS_POST_MAXSIZE': [200, 150, 150],
                                             'IOU_RECTIFIER': [0.5, 0.71, 0.65]})
    return c


def optimization_cfg(batch_size_per_gpu=8, num_epochs=30):
    """OPTIMIZATION section of the ssl yamls (gd_mae_ssl.yaml:183-203)."""
    return AttrDict({'BATCH_SIZE_PER_GPU': batch_size_per_gpu, 'NUM_EPOCHS': num_epochs, 'OPTIMIZER': 'adam_onecycle',
                     'LR': 0.003, 'WEIGHT_DECAY': 0.01, 'MOMENTUM': 0.9, 'MOMS': [0.95, 0.85], 'PCT_START': 0.4,
                     'DIV_FACTOR': 10, 'DECAY_STEP_LIST': [35, 45], 'LR_DECAY': 0.1, 'LR_CLIP': 0.0000001,
                     'LR_WARMUP': False, 'WARMUP_EPOCH': 1, 'GRAD_NORM_CLIP': 10})


class SyntheticDatasetInfo:
    """The attributes ``Detector3DTemplate.build_networks`` reads from the dataset object
    (detector3d_template.py:45-53; dataset.py:27-41; data_processor.py:166-171)."""

    def __init__(self, point_cloud_range, voxel_size, num_point_features, class_names):
        self.class_names = list(class_names)
        self.point_cloud_range = np.array(point_cloud_range, dtype=np.float32)
        self.voxel_size = list(voxel_size)
        grid = (self.point_cloud_range[3:6] - self.point_cloud_range[0:3]) / np.array(voxel_size)
        self.grid_size = np.round(grid).astype(np.int64)
        self.point_feature_encoder = AttrDict({'num_point_features': num_point_features})


WAYMO = dict(point_cloud_range=[-74.88, -74.88, -2, 74.88, 74.88, 4.0], voxel_size=[0.32, 0.32, 6.0],
             num_point_features=5, class_names=['Vehicle', 'Pedestrian', 'Cyclist'])
KITTI = dict(point_cloud_range=[0, -39.68, -3, 69.12, 39.68, 1], voxel_size=[0.32, 0.32, 4],
             num_point_features=4, class_names=['Car', 'Pedestrian', 'Cyclist'])
ONCE = dict(point_cloud_range=[-74.88, -74.88, -5.0, 74.88, 74.88, 3.0], voxel_size=[0.32, 0.32, 8.0],
            num_point_features=4, class_names=['Car', 'Bus', 'Truck', 'Pedestrian', 'Cyclist'])


def named_config(name: str, mask_ratio=None):
    """BASELINE.json configs: 'A' CPU plumbing, 'B' Waymo-shape, 'E' ONCE-shape stress.
    Returns (model_cfg, dataset_info, synth kwargs)."""
    if name == 'A':
        m = gdmae_ssl_model_cfg(0.5 if mask_ratio is None else mask_ratio, d_models=(128,), ffs=(256,), num_blocks=1,
                                eval_metric='kitti')
        return m, SyntheticDatasetInfo(**KITTI), dict(beams=32, azimuths=600, extra=800, features=4)
    if name == 'B':
        m = gdmae_ssl_model_cfg(0.75 if mask_ratio is None else mask_ratio)
        return m, SyntheticDatasetInfo(**WAYMO), dict(beams=64, azimuths=2650, extra=10400, features=5)
    if name == 'D':
        # KITTI-shape fine-tune config (BASELINE-defined): 0.16 m pillars (432 x 496), SPTBackbone ->
        # SSTBEVBackbone -> CenterHead behind the CenterPoint detector; head / BEV sections = waymo_models/gd_mae.yaml
        ds = SyntheticDatasetInfo(point_cloud_range=KITTI['point_cloud_range'], voxel_size=[0.16, 0.16, 4], num_point_features=4,
                                  class_names=KITTI['class_names'])
        ssl = gdmae_ssl_model_cfg(0.0, eval_metric='kitti')
        m = AttrDict({'NAME': 'CenterPoint', 'VFE': ssl.VFE, 'BACKBONE_3D': gdmae_finetune_backbone_cfg(),
                      'BACKBONE_2D': sst_bev_backbone_cfg(),
                      'DENSE_HEAD': center_head_cfg(KITTI['class_names'], post_range=(0, -40, -3, 70.4, 40, 1)),
                      'POST_PROCESSING': ssl.POST_PROCESSING})
        return m, ds, dict(beams=32, azimuths=600, extra=800, features=4)
    if name == 'E':
        m = gdmae_ssl_model_cfg(0.75 if mask_ratio is None else mask_ratio, d_models=(256, 256, 256), ffs=(512, 512, 512),
                                num_blocks=1, vfe_mlps=(64, 256), eval_metric='once')
        return m, SyntheticDatasetInfo(**ONCE), dict(beams=40, azimuths=1400, extra=4000, features=4)
    raise KeyError(name)
