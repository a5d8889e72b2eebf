from .vfe_template import VFETemplate
from .dyn_vfe import DynVFE

__all__ = {
    'VFETemplate': VFETemplate,
    'DynVFE': DynVFE,
}
