from .detector3d_template import Detector3DTemplate
from .centerpoint import CenterPoint
from .gd_mae import GDMAE

__all__ = {
    'Detector3DTemplate': Detector3DTemplate,
    'GDMAE': GDMAE,
    'CenterPoint': CenterPoint,
}


def build_detector(model_cfg, num_class, dataset, logger):
    return __all__[model_cfg.NAME](model_cfg=model_cfg, num_class=num_class, dataset=dataset, logger=logger)
