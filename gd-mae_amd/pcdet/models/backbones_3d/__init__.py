from .spt_backbone import SPTBackbone
from .spt_backbone_mae import SPTBackboneMAE

__all__ = {
    'SPTBackboneMAE': SPTBackboneMAE,
    'SPTBackbone': SPTBackbone,
}
