"""OpenPCDet-compatible package surface for the GD-MAE pre-training hot path, MI355X native.

Only the modules on the hot path exist here: the registries expose ``DynVFE``,
``SPTBackboneMAE``, ``SPTBackbone`` and ``GDMAE`` under the reference's names so a
``tools/train.py``-style driver resolves them from an unchanged yaml.  Unlike the reference's
``pcdet/__init__.py`` nothing is imported eagerly and no CUDA extension is required at import time.
"""
__version__ = "0.5.1+gdmae.mi355x"
