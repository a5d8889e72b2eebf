"""CPU restatement of the THIRD-PARTY arithmetic the GD-MAE hot path reaches (test infrastructure).

TEST INFRASTRUCTURE ONLY: nothing under ``oracle/`` may be imported by the product path
(``gd-mae_amd/``); only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker.

**parity unpinned** for everything in this file: the three libraries below are un-vendored
dependencies of the reference (no version pin, absent from this image, and the reference ships no
test that pins their results), so the functions here restate their *published* semantics and are
anchored on the reference's call sites only:

* ``torch_scatter`` (README.md:23)   call sites ``pcdet/models/backbones_3d/vfe/dyn_vfe.py:81,109``
* ``spconv`` 2.x (README.md:22)      call sites ``pcdet/utils/spconv_utils.py:37-56``,
  ``pcdet/models/backbones_3d/spt_backbone.py:206,217``, ``spt_backbone_mae.py:102-107,128``
* ``pytorch3d`` (README.md:21)       call site ``pcdet/models/backbones_3d/spt_backbone_mae.py:88``

They are cross-checked in ``tests/test_oracle_thirdparty.py`` against independent formulations
(per-site neighbour loops over a coordinate dictionary for the sparse convolutions, per-segment python
loops for the scatters, brute-force ``cdist`` for the Chamfer distance).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# torch_scatter
# --------------------------------------------------------------------------------------------
def scatter_mean(src: torch.Tensor, index: torch.Tensor, dim_size: int | None = None) -> torch.Tensor:
    """``torch_scatter.scatter(src, index, dim=0, reduce='mean')`` (dyn_vfe.py:81).

    mean = sum / count (count clamped to >= 1).  The sum is accumulated in ascending row order
    (``index_add_`` on CPU is a sequential loop), which is the canonical order of SURVEY §9.0.
    """
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    out = out.index_add(0, index, src)
    cnt = torch.zeros(dim_size, dtype=src.dtype).index_add(0, index, torch.ones_like(index, dtype=src.dtype))
    cnt = cnt.clamp(min=1)
    return out / cnt.view(-1, *([1] * (src.dim() - 1)))


def scatter_max(src: torch.Tensor, index: torch.Tensor, dim_size: int | None = None):
    """``torch_scatter.scatter_max(src, index, dim=0)`` (dyn_vfe.py:109) -> (max, argmax).

    Gradient flows to the arg-max row only (implemented through ``gather`` on the arg-max).
    Ties: the lowest row index wins (canonical; the library's CUDA tie-break is unspecified).
    """
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    n, c = src.shape
    with torch.no_grad():
        idx = index.view(-1, 1).expand(n, c)
        mx = torch.full((dim_size, c), float("-inf"), dtype=src.dtype).scatter_reduce(
            0, idx, src.detach(), reduce="amax", include_self=True)
        is_max = src.detach() == mx[index]
        rows = torch.arange(n).view(-1, 1).expand(n, c)
        cand = torch.where(is_max, rows, torch.full_like(rows, n))
        arg = torch.full((dim_size, c), n, dtype=torch.long).scatter_reduce(
            0, idx, cand, reduce="amin", include_self=True)
    safe = arg.clamp(max=max(n - 1, 0))
    out = torch.gather(src, 0, safe)
    return out, arg


# --------------------------------------------------------------------------------------------
# spconv (2-D, as used by post_act_block(dim=2))
# --------------------------------------------------------------------------------------------
def linear_key(indices: torch.Tensor, spatial_shape) -> torch.Tensor:
    """(b, y, x) int -> b*Y*X + y*X + x  (canonical site order)."""
    Y, X = int(spatial_shape[0]), int(spatial_shape[1])
    ind = indices.long()
    return (ind[:, 0] * Y + ind[:, 1]) * X + ind[:, 2]


def densify(features: torch.Tensor, indices: torch.Tensor, spatial_shape, batch_size: int) -> torch.Tensor:
    """``SparseConvTensor.dense()`` -> (B, C, Y, X), zeros at inactive sites (spt_backbone_mae.py:128)."""
    Y, X = int(spatial_shape[0]), int(spatial_shape[1])
    C = features.shape[1]
    flat = torch.zeros(batch_size * Y * X, C, dtype=features.dtype)
    flat = flat.index_copy(0, linear_key(indices, spatial_shape), features)
    return flat.view(batch_size, Y, X, C).permute(0, 3, 1, 2).contiguous()


def subm_conv2d(features, indices, spatial_shape, batch_size, weight):
    """``spconv.SubMConv2d(k=3, bias=False)`` (spconv_utils.py:41; spt_backbone.py:217).

    weight layout is spconv-2.x ``(Cout, kH, kW, Cin)`` (detector3d_template.py:361-380).
    out[p] = sum_k W[:, ky, kx, :] . in[p + k - 1] over ACTIVE neighbours; active set unchanged.
    = dense cross-correlation on the zero-filled map restricted to the active set.
    """
    dense = densify(features, indices, spatial_shape, batch_size)
    out = F.conv2d(dense, weight.permute(0, 3, 1, 2), padding=1)
    ind = indices.long()
    return out[ind[:, 0], :, ind[:, 1], ind[:, 2]]


def strided_out_shape(spatial_shape, k=3, s=2, p=1):
    return [(int(n) + 2 * p - k) // s + 1 for n in spatial_shape]


def sparse_conv2d(features, indices, spatial_shape, batch_size, weight, stride=2, padding=1):
    """``spconv.SparseConv2d(k=3, stride=2, padding=1, bias=False)`` (spconv_utils.py:43;
    spt_backbone.py:206).  Output site o is active iff some active input i = s*o - p + k exists;
    out[o] = sum_{k: i active} W[:, ky, kx, :] . in[i].  Output sites are returned ordered by the
    linear key at the output resolution (canonical; spconv's own order is hash dependent).
    Returns (out_features, out_indices[int32 (b,y,x)], out_spatial_shape).
    """
    k = weight.shape[1]
    dense = densify(features, indices, spatial_shape, batch_size)
    out = F.conv2d(dense, weight.permute(0, 3, 1, 2), stride=stride, padding=padding)
    occ = densify(torch.ones(features.shape[0], 1, dtype=features.dtype), indices, spatial_shape, batch_size)
    act = F.conv2d(occ, torch.ones(1, 1, k, k, dtype=features.dtype), stride=stride, padding=padding) > 0.5
    out_idx = act[:, 0].nonzero()  # (b, y, x) ascending lexicographic = ascending linear key
    of = out[out_idx[:, 0], :, out_idx[:, 1], out_idx[:, 2]]
    return of, out_idx.int(), list(out.shape[-2:])


# --------------------------------------------------------------------------------------------
# pytorch3d.loss.chamfer_distance
# --------------------------------------------------------------------------------------------
def chamfer_distance(x: torch.Tensor, y: torch.Tensor, weights: torch.Tensor | None = None):
    """``pytorch3d.loss.chamfer_distance(x, y, weights=w)`` with library defaults: squared L2,
    point_reduction='mean', batch_reduction='mean' (spt_backbone_mae.py:88).

    loss = [sum_n w_n (1/P1) sum_i min_j |x_ni - y_nj|^2 + sum_n w_n (1/P2) sum_j min_i |..|^2] / sum_n w_n
    and exactly 0 (with a graph) when sum w == 0.  Returns (loss, None) like the library.
    """
    N, P1, _ = x.shape
    P2 = y.shape[1]
    if weights is not None and float(weights.sum()) == 0.0:
        return (x.sum((1, 2)) * weights).sum() * 0.0, None
    d = ((x.unsqueeze(2) - y.unsqueeze(1)) ** 2).sum(-1)  # (N, P1, P2)
    cham_x = d.min(dim=2).values  # (N, P1)
    cham_y = d.min(dim=1).values  # (N, P2)
    if weights is not None:
        cham_x = cham_x * weights.view(N, 1)
        cham_y = cham_y * weights.view(N, 1)
    cham_x = cham_x.sum(1) / P1
    cham_y = cham_y.sum(1) / P2
    div = weights.sum() if weights is not None else max(N, 1)
    return cham_x.sum() / div + cham_y.sum() / div, None
