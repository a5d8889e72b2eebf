from .sst_bev_backbone import SSTBEVBackbone

# name -> class, as the reference registry (pcdet/models/backbones_2d/__init__.py)
__all__ = {
    'SSTBEVBackbone': SSTBEVBackbone,
}
