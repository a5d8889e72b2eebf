from .center_head import CenterHead

# name -> class, as the reference registry (pcdet/models/dense_heads/__init__.py)
__all__ = {
    'CenterHead': CenterHead,
}
