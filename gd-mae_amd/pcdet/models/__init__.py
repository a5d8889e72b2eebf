"""``build_network`` / ``load_data_to_gpu`` / ``model_fn_decorator`` with the reference's signatures
(pcdet/models/__init__.py:9-41)."""
from collections import namedtuple

import numpy as np
import torch

from .detectors import build_detector


def build_network(model_cfg, num_class, dataset, logger):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset, logger=logger)


_SKIP = ('frame_id', 'metadata', 'calib', 'image_shape', 'image_pad_shape', 'image_rescale_shape')


def load_data_to_gpu(batch_dict):
    """ndarray -> float32 device tensor (the batch index stays in column 0 of ``points`` as a float)."""
    for key, val in batch_dict.items():
        if isinstance(val, np.ndarray) and key not in _SKIP:
            t = torch.from_numpy(val).float()
            batch_dict[key] = t.pin_memory().cuda(non_blocking=True) if torch.cuda.is_available() else t


def model_fn_decorator():
    ModelReturn = namedtuple('ModelReturn', ['loss', 'tb_dict', 'disp_dict'])

    def model_func(model, batch_dict, **kwargs):
        load_data_to_gpu(batch_dict)
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        loss = ret_dict['loss'].mean()
        (model if hasattr(model, 'update_global_step') else model.module).update_global_step()
        return ModelReturn(loss, tb_dict, disp_dict)

    return model_func
