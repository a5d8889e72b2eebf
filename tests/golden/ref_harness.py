"""Import harness for the *unmodified* reference Python modules (build container only).

Used ONLY by ``tests/golden/make_golden.py`` to generate golden vectors.  ``/root/reference`` does
not exist on the GPU box, so nothing in the test-suite imports this module at test time.

What it does: registers ``pcdet`` and its sub-packages in ``sys.modules`` as bare
namespace modules whose ``__path__`` points into ``/root/reference/pcdet`` (bypassing the package
``__init__``s that import every CUDA extension), and provides stand-ins for the pieces that are
absent from this image:

* ``pcdet.ops.sst_ops.sst_ops_cuda``  - CPU restatement of the 3 kernels of
  ``pcdet/ops/sst_ops/src/sst_ops_gpu.cu:14-39`` in canonical (ascending index) order;
* ``spconv`` / ``spconv.pytorch``, ``torch_scatter``, ``pytorch3d.loss`` - thin adapters over
  ``oracle/thirdparty.py`` (our restatement of those libraries' published semantics; parity for
  these three is *unpinned*, see that file's header);
* ``SharedArray``, ``easydict`` - trivial.

No reference source is copied: the reference files are imported from where they lie.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from oracle import thirdparty as tp  # noqa: E402


class EasyDict(dict):
    """Minimal attribute dict (easydict is not installed)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            v = EasyDict(v)
        elif isinstance(v, (list, tuple)):
            v = type(v)(EasyDict(x) if isinstance(x, dict) and not isinstance(x, EasyDict) else x for x in v)
        super().__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def update(self, d=None, **kw):
        for k, v in dict(d or {}, **kw).items():
            self[k] = v


# ----------------------------------------------------------------------------------------------
# sst_ops_cuda stand-in (canonical order = sequential loop over the kernels' bodies)
# ----------------------------------------------------------------------------------------------
def _ingroup_inds_wrapper(group_inds: torch.Tensor, out_inds: torch.Tensor) -> int:
    g = group_inds.numpy()
    order = np.argsort(g, kind="stable")
    gs = g[order]
    start = np.r_[0, np.flatnonzero(gs[1:] != gs[:-1]) + 1]
    seg = np.repeat(start, np.diff(np.r_[start, len(gs)]))
    rank = np.arange(len(gs)) - seg
    out = np.empty(len(g), dtype=np.int64)
    out[order] = rank
    out_inds.copy_(torch.from_numpy(out))
    return 1


def _group_inner_inds_wrapper(inverse_inds: torch.Tensor, group_inds: torch.Tensor) -> int:
    M, K = group_inds.shape
    inv = inverse_inds.numpy()
    rank = torch.empty_like(inverse_inds)
    _ingroup_inds_wrapper(inverse_inds, rank)
    rank = rank.numpy()
    g = group_inds.numpy()
    sel = rank < K
    g[inv[sel], rank[sel]] = np.flatnonzero(sel)
    cnt = np.bincount(inv, minlength=M)
    for m in np.flatnonzero((cnt > 0) & (cnt < K)):
        c = cnt[m]
        g[m, c:] = g[m, np.arange(c, K) % c]
    return 1


# ----------------------------------------------------------------------------------------------
# spconv stand-in
# ----------------------------------------------------------------------------------------------
class SparseConvTensor:
    def __init__(self, features, indices, spatial_shape, batch_size):
        self.features = features
        self.indices = indices
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)

    def replace_feature(self, f):
        return SparseConvTensor(f, self.indices, self.spatial_shape, self.batch_size)

    def dense(self):
        return tp.densify(self.features, self.indices, self.spatial_shape, self.batch_size)


class SparseModule(nn.Module):
    pass


class SparseConvolution(SparseModule):
    def __init__(self, cin, cout, k, stride=1, padding=0, bias=False, indice_key=None, subm=False):
        super().__init__()
        assert not bias
        self.subm, self.stride, self.padding, self.k = subm, stride, padding, k
        # spconv 2.x layout (Cout, kH, kW, Cin)
        self.weight = nn.Parameter(torch.empty(cout, k, k, cin))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)

    def forward(self, x: SparseConvTensor):
        if self.subm:
            f = tp.subm_conv2d(x.features, x.indices, x.spatial_shape, x.batch_size, self.weight)
            return x.replace_feature(f)
        f, idx, shp = tp.sparse_conv2d(x.features, x.indices, x.spatial_shape, x.batch_size, self.weight,
                                       self.stride, self.padding)
        return SparseConvTensor(f, idx, shp, x.batch_size)


class SubMConv2d(SparseConvolution):
    def __init__(self, cin, cout, k, bias=False, indice_key=None, **kw):
        super().__init__(cin, cout, k, 1, k // 2, bias, indice_key, subm=True)


class SparseConv2d(SparseConvolution):
    def __init__(self, cin, cout, k, stride=1, padding=0, bias=False, indice_key=None, **kw):
        super().__init__(cin, cout, k, stride, padding, bias, indice_key, subm=False)


class SparseSequential(SparseModule):
    def __init__(self, *mods):
        super().__init__()
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def forward(self, x):
        for m in self._modules.values():
            if isinstance(m, SparseModule):
                x = m(x)
            else:
                x = x.replace_feature(m(x.features))
        return x


def _make_spconv():
    sp = types.ModuleType("spconv")
    spt = types.ModuleType("spconv.pytorch")
    conv = types.ModuleType("spconv.pytorch.conv")
    conv.SparseConvolution = SparseConvolution
    for m in (spt,):
        m.SparseConvTensor = SparseConvTensor
        m.SparseModule = SparseModule
        m.SubMConv2d = SubMConv2d
        m.SparseConv2d = SparseConv2d
        m.SparseSequential = SparseSequential
        m.conv = conv
    sp.pytorch = spt
    return sp, spt, conv


def install():
    """Install namespace packages + stand-ins; idempotent."""
    if "pcdet" in sys.modules and getattr(sys.modules["pcdet"], "_ref_harness", False):
        return
    for name in ["pcdet", "pcdet.models", "pcdet.models.backbones_3d", "pcdet.models.backbones_3d.vfe",
                 "pcdet.models.model_utils", "pcdet.utils", "pcdet.ops", "pcdet.ops.sst_ops",
                 "pcdet.datasets"]:
        m = types.ModuleType(name)
        m.__path__ = [os.path.join(REF, *name.split("."))]
        m._ref_harness = True
        sys.modules[name] = m
    cuda = types.ModuleType("pcdet.ops.sst_ops.sst_ops_cuda")
    cuda.ingroup_inds_wrapper = _ingroup_inds_wrapper
    cuda.group_inner_inds_wrapper = _group_inner_inds_wrapper
    sys.modules["pcdet.ops.sst_ops.sst_ops_cuda"] = cuda
    sys.modules["pcdet.ops.sst_ops"].sst_ops_cuda = cuda

    sp, spt, conv = _make_spconv()
    sys.modules["spconv"] = sp
    sys.modules["spconv.pytorch"] = spt
    sys.modules["spconv.pytorch.conv"] = conv

    ts = types.ModuleType("torch_scatter")

    def scatter(src, index, dim=0, reduce="mean"):
        assert dim == 0 and reduce == "mean"
        return tp.scatter_mean(src, index)

    def scatter_max(src, index, dim=0):
        assert dim == 0
        return tp.scatter_max(src, index)

    ts.scatter, ts.scatter_max = scatter, scatter_max
    sys.modules["torch_scatter"] = ts

    p3 = types.ModuleType("pytorch3d")
    p3l = types.ModuleType("pytorch3d.loss")
    p3l.chamfer_distance = tp.chamfer_distance
    p3.loss = p3l
    sys.modules["pytorch3d"] = p3
    sys.modules["pytorch3d.loss"] = p3l

    sys.modules["SharedArray"] = types.ModuleType("SharedArray")
    ed = types.ModuleType("easydict")
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed


def ref(name: str):
    install()
    return importlib.import_module(name)


def load_yaml_cfg(rel_path: str) -> EasyDict:
    import yaml
    with open(os.path.join(REF, "tools", rel_path)) as f:
        return EasyDict(yaml.safe_load(f))
