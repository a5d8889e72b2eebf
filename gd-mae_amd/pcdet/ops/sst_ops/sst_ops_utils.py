"""Python op API of the reference's ``sst_ops`` extension (pcdet/ops/sst_ops/sst_ops_utils.py:5-27) on the
HIP library: same names, arguments and results (canonical ascending-index order instead of the CUDA
kernels' atomic arrival order).  Like the reference wrappers these need device tensors; there is no CPU path.
"""
import torch

from gdmae_hip import lib as L


def _ws(n, n_groups, dev):
    nbytes = L.load().gdmae_group_workspace_bytes(n, n_groups)
    return torch.empty(nbytes, dtype=torch.uint8, device=dev), nbytes


def get_inner_win_inds(group_inds):
    """group_inds (N,) int64 -> (N,) int64 rank of every element inside its group."""
    group_inds = group_inds.contiguous()
    out = torch.zeros_like(group_inds) - 1
    if group_inds.numel() == 0:
        return out
    n_groups = int(group_inds.max().item()) + 1      # same host sync as the reference wrapper (sst_ops.cpp:26)
    ws, nb = _ws(group_inds.numel(), n_groups, group_inds.device)
    L.call("gdmae_ingroup_inds", L.ptr(group_inds), group_inds.numel(), n_groups, L.ptr(out), L.ptr(ws), nb, L.stream())
    return out


def group_inner_inds(points, inverse_inds, K):
    """points (N, C), inverse_inds (N,) -> (valid_voxel_num + 1, K, C) grouped points."""
    inverse_inds = inverse_inds.contiguous()
    M = int(inverse_inds.max().item()) + 1
    group_inds = torch.full((M, K), -1, dtype=torch.long, device=points.device)
    ws, nb = _ws(inverse_inds.numel(), M, points.device)
    L.call("gdmae_group_inner_inds", L.ptr(inverse_inds), inverse_inds.numel(), M, K, L.ptr(group_inds), L.ptr(ws), nb,
           L.stream())
    return points[group_inds]
