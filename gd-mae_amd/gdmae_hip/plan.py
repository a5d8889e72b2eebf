"""Geometry plan: every index structure of a pre-training step, built on the GPU by the HIP library.

``voxelize`` covers SURVEY §8 rows a1-a3 (+ the canonical in-pillar order used by a17);
``encoder_plan`` covers a5 (masking), the index side of a6 (token sets + sparse-conv rulebooks) and
a7-a10 (shifted-window partition) for all stages and both shifts.  Each of the two functions enqueues
all of its kernels back-to-back and performs exactly ONE device->host copy of a few int32 counts at
the end (the reference performs dozens of ``.item()`` / ``unique`` / boolean-index syncs on this path:
SURVEY §3.2).  Nothing here falls back to PyTorch/CPU arithmetic: sizes and pointers only.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import lib as L

I32 = torch.int32


def _empty(n, dtype, dev):
    return torch.empty(int(n), dtype=dtype, device=dev)


@dataclass
class VoxelPlan:
    batch_size: int
    grid: tuple            # (X, Y, Z)
    lo: tuple
    vs: tuple
    n_cols: int
    N: int
    M: int
    points: torch.Tensor          # (N, 1+F) kept points, original order
    point_coords: torch.Tensor    # (N, 4) int64 [b, z, y, x]
    inverse: torch.Tensor         # (N,) int64
    inverse32: torch.Tensor       # (N,) int32
    voxel_coords: torch.Tensor    # (M, 4) int64
    pillar_cell: torch.Tensor     # (M,) int32
    pt_off: torch.Tensor          # (M+1,) int32
    pillar_pts: torch.Tensor      # (N,) int32
    point_rank: torch.Tensor      # (N,) int32
    sample_off: torch.Tensor      # (B+1,) int32
    pillar_mean: torch.Tensor     # (M, F)
    cell2pillar: torch.Tensor     # (B*Z*Y*X,) int32
    counts: torch.Tensor          # device int32[2]
    points_pm: torch.Tensor = None    # (N, 1+F) the kept points in pillar-major order: points[pillar_pts]
    row_pillar: torch.Tensor = None   # (N,) int32 pillar of each pillar-major row: inverse32[pillar_pts]


def _voxelize_launch(points: torch.Tensor, point_cloud_range, voxel_size, grid_size, batch_size: int) -> dict:
    """Enqueue gdmae_voxelize on the current stream with capacity-sized outputs; no host sync."""
    assert points.is_cuda and points.dtype == torch.float32 and points.dim() == 2
    points = points.contiguous()
    dev = points.device
    n0, ncols = points.shape
    gx, gy, gz = (int(g) for g in grid_size)
    cells = batch_size * gx * gy * gz
    lo = tuple(float(v) for v in point_cloud_range[:3])
    vs = tuple(float(v) for v in voxel_size)
    F = ncols - 1
    cap = max(n0, 1)
    r = dict(batch_size=batch_size, grid=(gx, gy, gz), lo=lo, vs=vs, ncols=ncols, n0=n0, src=points,
             pts_out=_empty(cap * ncols, torch.float32, dev), pcoords=_empty(cap * 4, torch.int64, dev),
             inverse=_empty(cap, torch.int64, dev), inverse32=_empty(cap, I32, dev), vcoords=_empty(cap * 4, torch.int64, dev),
             pillar_cell=_empty(cap, I32, dev), pt_off=_empty(cap + 1, I32, dev), pillar_pts=_empty(cap, I32, dev),
             prank=_empty(cap, I32, dev), sample_off=_empty(batch_size + 1, I32, dev), mean=_empty(cap * F, torch.float32, dev),
             cell2pillar=_empty(cells, I32, dev), counts=_empty(2, I32, dev))
    wsb = L.load().gdmae_voxelize_workspace_bytes(n0, batch_size, gx, gy, gz)
    r["ws"] = _empty(wsb, torch.uint8, dev)
    L.call("gdmae_voxelize", L.ptr(points), n0, ncols, L.host_f32(lo), L.host_f32(vs), L.host_i32((gx, gy, gz)),
           batch_size, L.ptr(r["pts_out"]), L.ptr(r["pcoords"]), L.ptr(r["inverse"]), L.ptr(r["inverse32"]), L.ptr(r["vcoords"]),
           L.ptr(r["pillar_cell"]), L.ptr(r["pt_off"]), L.ptr(r["pillar_pts"]), L.ptr(r["prank"]), L.ptr(r["sample_off"]),
           L.ptr(r["mean"]), L.ptr(r["cell2pillar"]), L.ptr(r["counts"]), L.ptr(r["ws"]), wsb, L.stream())
    # pillar-major copy of the kept points (rows of a pillar contiguous) for the DynVFE point layers
    r["pts_pm"] = _empty(cap * ncols, torch.float32, dev)
    r["row_pillar"] = _empty(cap, I32, dev)
    L.call("gdmae_pillar_major_rows", L.ptr(r["pts_out"]), ncols, L.ptr(r["pillar_pts"]), L.ptr(r["inverse32"]), L.ptr(r["counts"]),
           cap, L.ptr(r["pts_pm"]), L.ptr(r["row_pillar"]), L.stream())
    return r


def _voxelize_finalize(r: dict, N: int, M: int) -> VoxelPlan:
    ncols = r["ncols"]
    F = ncols - 1
    return VoxelPlan(r["batch_size"], r["grid"], r["lo"], r["vs"], ncols, N, M,
                     r["pts_out"][:N * ncols].view(N, ncols), r["pcoords"][:N * 4].view(N, 4), r["inverse"][:N],
                     r["inverse32"][:N], r["vcoords"][:M * 4].view(M, 4), r["pillar_cell"][:M], r["pt_off"][:M + 1],
                     r["pillar_pts"][:N], r["prank"][:N], r["sample_off"], r["mean"][:M * F].view(M, F), r["cell2pillar"],
                     r["counts"], r["pts_pm"][:N * ncols].view(N, ncols), r["row_pillar"][:N])


def voxelize(points: torch.Tensor, point_cloud_range, voxel_size, grid_size, batch_size: int) -> VoxelPlan:
    """points (N0, 1+F) fp32 on the GPU -> VoxelPlan (one host sync for N, M)."""
    r = _voxelize_launch(points, point_cloud_range, voxel_size, grid_size, batch_size)
    N, M = (int(v) for v in r["counts"].tolist())      # the one host sync of this phase
    return _voxelize_finalize(r, N, M)


@dataclass
class WindowPlan:
    """One shift of one stage (tokens grouped into windows, windows grouped into occupancy levels)."""
    tok_win: torch.Tensor
    tok_level: torch.Tensor
    tok_slot: torch.Tensor
    tok_pos: torch.Tensor
    csr_tok: torch.Tensor
    win_start: torch.Tensor
    win_len: torch.Tensor
    n_win: List[int] = field(default_factory=list)    # per level
    n_tok: List[int] = field(default_factory=list)    # per level
    max_tokens: List[int] = field(default_factory=list)


@dataclass
class StagePlan:
    B: int
    Y: int
    X: int
    n_tok: int
    tok_cell: torch.Tensor            # (n_tok,) int32 linear key (b*Y+y)*X+x, ascending
    map: torch.Tensor                 # (B*Y*X,) int32 cell -> token / -1
    nbr_subm: torch.Tensor            # (n_tok, 9) int32
    nbr_down: Optional[torch.Tensor]  # (n_tok, 9) ids into the previous stage (strided conv forward)
    nbr_down_t: Optional[torch.Tensor]  # (n_prev, 9) ids into THIS stage (strided conv backward)
    windows: List[WindowPlan] = field(default_factory=list)

    def indices_byx(self) -> torch.Tensor:
        c = self.tok_cell.long()
        x = c % self.X
        r = c // self.X
        return torch.stack([r // self.Y, r % self.Y, x], dim=-1).int()


@dataclass
class DecoderTiles:
    """Active 8x8 tiles of the full-resolution BEV map for the decoder's 3x3 conv_out (csrc/conv_tiles.hip): the tiles
    whose one-site halo touches a site covered by a token of one of the decoder's source stages."""
    sources: tuple                     # stage indices the tile set was built for
    B: int
    H: int
    W: int
    n_act: int
    tile_slot: torch.Tensor            # (B * ceil(H/8) * ceil(W/8),) int32 tile -> slot / -1
    tile_list: torch.Tensor            # (n_act,) int32 ascending tile ids
    nbr: Optional[list] = None         # per source stage (n_sites, 9) int32: tile-compact row of (site, tap) / -1 (conv_out backward)


@dataclass
class EncoderPlan:
    mask: Optional[torch.Tensor]       # (M,) fp32 0 visible / 1 masked (None when nothing is masked)
    tok_pillar: torch.Tensor           # (M1,) pillar id of each stage-1 token
    stages: List[StagePlan]
    dec_tiles: Optional[DecoderTiles] = None


def upsample_cells(c: torch.Tensor, Ys: int, Xs: int, s: int) -> torch.Tensor:
    """(len(c), s*s) int32 full-resolution cells ((b*Y + y)*X + x, Y = Ys*s, X = Xs*s) covered by the stride-s cells c."""
    x = c % Xs
    r = torch.div(c, Xs, rounding_mode='floor')
    y = r % Ys
    b = torch.div(r, Ys, rounding_mode='floor')
    d = torch.arange(s, device=c.device, dtype=c.dtype)
    yy = (y * s).view(-1, 1, 1) + d.view(1, s, 1)
    xx = (x * s).view(-1, 1, 1) + d.view(1, 1, s)
    return ((b.view(-1, 1, 1) * (Ys * s) + yy) * (Xs * s) + xx).reshape(c.numel(), s * s)


def _drop_arrays(drop_info):
    d = {int(k): v for k, v in drop_info.items()}
    keys = sorted(d)
    assert keys == list(range(len(keys))) and len(keys) <= 3
    lo = [int(d[k]["drop_range"][0]) for k in keys]
    hi = [int(d[k]["drop_range"][1]) for k in keys]
    T = [int(d[k]["max_tokens"]) for k in keys]
    return lo, hi, T


def _decoder_tiles_launch(maps, ups, B, H, W, dev) -> dict:
    """Enqueue gdmae_decoder_tiles on the current stream (capacity-sized outputs, count left on the device)."""
    nt = B * ((H + 7) // 8) * ((W + 7) // 8)
    r = dict(tile_slot=_empty(nt, I32, dev), tile_list=_empty(nt, I32, dev), n_act=_empty(1, I32, dev), B=B, H=H, W=W)
    ws = _empty(L.load().gdmae_decoder_tiles_workspace_bytes(B, H, W), torch.uint8, dev)
    L.call("gdmae_decoder_tiles", L.host_ptrs(maps), L.host_i32(ups), len(maps), B, H, W, L.ptr(r["tile_slot"]),
           L.ptr(r["tile_list"]), L.ptr(r["n_act"]), L.ptr(ws), L.stream())
    r["_ws"] = ws
    return r


def decoder_tiles(ep: "EncoderPlan", sources, H: int, W: int) -> DecoderTiles:
    """Tile set for the given decoder source stages (cached on the plan; built inline with ONE host sync when the plan was
    made without it)."""
    sources = tuple(int(v) for v in sources)
    dt = ep.dec_tiles
    if dt is not None and dt.sources == sources and dt.H == H and dt.W == W:
        return dt
    st = [ep.stages[i] for i in sources]
    B = st[0].B
    ups = [H // sp.Y for sp in st]
    assert all(sp.Y * u == H and sp.X * u == W for sp, u in zip(st, ups))
    r = _decoder_tiles_launch([sp.map for sp in st], ups, B, H, W, st[0].map.device)
    n = int(r["n_act"].item())
    ep.dec_tiles = DecoderTiles(sources, B, H, W, n, r["tile_slot"], r["tile_list"][:n], _site_rulebooks(st, ups, r["tile_slot"], H, W))
    return ep.dec_tiles


def decoder_site_rulebooks(dt: "DecoderTiles", sites, ups):
    """Rulebooks of the conv_out backward for a tile set that came without them (inline plans): ``sites[g]`` = full-resolution
    cells of source stage g's active sites, ``ups[g]`` its upsampling stride."""
    out = []
    for st, u in zip(sites, ups):
        n_dev = torch.tensor([st.numel() // (u * u)], dtype=I32, device=st.device)
        nbr = _empty(st.numel() * 9, I32, st.device)
        L.call("gdmae_decoder_site_rulebook", L.ptr(st.contiguous()), L.ptr(n_dev), u * u, st.numel(), L.ptr(dt.tile_slot), dt.H, dt.W,
               L.ptr(nbr), L.stream())
        out.append(nbr.view(-1, 9))
    return out


def _site_rulebooks(stages, ups, tile_slot, H, W):
    """Per source stage the (n_sites, 9) rulebook of the conv_out backward (gdmae_decoder_site_rulebook)."""
    out = []
    for sp, u in zip(stages, ups):
        sites = sp.tok_cell if u == 1 else (sp._up_sites[1] if getattr(sp, "_up_sites", None) is not None and sp._up_sites[0] == u
                                            else upsample_cells(sp.tok_cell, sp.Y, sp.X, u).reshape(-1).contiguous())
        n_dev = torch.tensor([sp.n_tok], dtype=I32, device=tile_slot.device)
        nbr = _empty(sites.numel() * 9, I32, tile_slot.device)
        L.call("gdmae_decoder_site_rulebook", L.ptr(sites.contiguous()), L.ptr(n_dev), u * u, sites.numel(), L.ptr(tile_slot), H, W, L.ptr(nbr),
               L.stream())
        out.append(nbr.view(-1, 9))
    return out


def _encoder_launch(vox, m_cap: int, strides, window_shapes, drop_infos, keep_frac, noise, dec_sources=None) -> dict:
    """Enqueue masking + token sets + rulebooks + window partitions on the current stream; no host sync.
    ``vox``: VoxelPlan or the raw dict of _voxelize_launch; m_cap >= number of pillars (capacity)."""
    if isinstance(vox, dict):
        dev, grid, B = vox["src"].device, vox["grid"], vox["batch_size"]
        sample_off, pillar_cell, vcounts = vox["sample_off"], vox["pillar_cell"], vox["counts"]
    else:
        dev, grid, B = vox.points.device, vox.grid, vox.batch_size
        sample_off, pillar_cell, vcounts = vox.sample_off, vox.pillar_cell, vox.counts
    gx, gy, gz = grid
    assert gz == 1, "the SST backbone works on single-layer pillar grids (spt_backbone_mae.py:94)"
    lib = L.load()
    st = L.stream()
    M = m_cap
    scan_ws = _empty(8 * (max(B * gx * gy, M) // 4096 + 4), torch.int64, dev)

    # ---- a5 masking
    if keep_frac is not None:
        if noise is None:
            noise = torch.rand(max(M, 1), device=dev, dtype=torch.float32)
        assert noise.numel() >= 1 and noise.dtype == torch.float32
        mask = _empty(max(M, 1), torch.float32, dev)
        len_keep = _empty(B, I32, dev)
        L.call("gdmae_random_mask", L.ptr(noise.contiguous()), L.ptr(sample_off), B, float(keep_frac), L.ptr(mask),
               L.ptr(len_keep), st)
    else:
        mask = torch.zeros(max(M, 1), dtype=torch.float32, device=dev)

    # ---- stage-1 tokens = visible pillars
    cap = max(M, 1)
    Y, X = gy, gx
    tok_pillar = _empty(cap, I32, dev)
    tok_cell = _empty(cap, I32, dev)
    smap = _empty(B * Y * X, I32, dev)
    n_tok = _empty(1, I32, dev)
    L.call("gdmae_visible_tokens", L.ptr(mask), L.ptr(pillar_cell), L.ptr(vcounts), M, B * Y * X,
           L.ptr(tok_pillar), L.ptr(tok_cell), L.ptr(smap), L.ptr(n_tok), L.ptr(scan_ws), st)

    raw = []   # per stage dict of capacity-sized tensors
    for si, stride in enumerate(strides):
        if stride > 1:
            assert stride == 2, "only the k3 s2 p1 strided sparse conv of the shipped configs is implemented"
            Yi, Xi = Y, X
            Y, X = (Yi - 1) // 2 + 1, (Xi - 1) // 2 + 1
            cap_in = cap
            cap = min(4 * cap_in, B * Y * X)
            tok_cell_n = _empty(cap, I32, dev)
            smap_n = _empty(B * Y * X, I32, dev)
            n_tok_n = _empty(1, I32, dev)
            flag = _empty(B * Y * X, I32, dev)
            L.call("gdmae_downsample_tokens", L.ptr(n_tok), L.ptr(tok_cell), cap_in, B, Yi, Xi, L.ptr(tok_cell_n),
                   L.ptr(smap_n), L.ptr(n_tok_n), L.ptr(flag), L.ptr(scan_ws), st)
            nbr_down = _empty(cap * 9, I32, dev)
            L.call("gdmae_rulebook", L.ptr(n_tok_n), L.ptr(tok_cell_n), cap, B, Y, X, Yi, Xi, L.ptr(smap), 1,
                   L.ptr(nbr_down), st)
            nbr_down_t = _empty(cap_in * 9, I32, dev)
            L.call("gdmae_rulebook", L.ptr(n_tok), L.ptr(tok_cell), cap_in, B, Yi, Xi, Y, X, L.ptr(smap_n), 2,
                   L.ptr(nbr_down_t), st)
            tok_cell, smap, n_tok = tok_cell_n, smap_n, n_tok_n
        else:
            nbr_down = nbr_down_t = None
        nbr_subm = _empty(cap * 9, I32, dev)
        L.call("gdmae_rulebook", L.ptr(n_tok), L.ptr(tok_cell), cap, B, Y, X, Y, X, L.ptr(smap), 0, L.ptr(nbr_subm), st)
        wx, wy, wz = (int(v) for v in window_shapes[si])
        assert wz == 1
        dlo, dhi, dT = _drop_arrays(drop_infos[si])
        assert wx * wy <= max(dT), "token drop would not be the identity: unsupported (SURVEY header item 5)"
        wsb = lib.gdmae_window_workspace_bytes(B, Y, X, wx, wy)
        wins = []
        for shifted in (0, 1):
            nwin_cap = B * ((X + wx - 1) // wx + 1) * ((Y + wy - 1) // wy + 1)
            w = dict(tok_win=_empty(cap, I32, dev), tok_level=_empty(cap, I32, dev), tok_slot=_empty(cap, I32, dev),
                     tok_pos=_empty(cap, I32, dev), csr_tok=_empty(cap, I32, dev),
                     win_start=_empty(min(nwin_cap, cap), I32, dev), win_len=_empty(min(nwin_cap, cap), I32, dev),
                     counts=_empty(8, I32, dev), T=dT)
            ws = _empty(wsb, torch.uint8, dev)
            L.call("gdmae_window_partition", L.ptr(smap), B, Y, X, wx, wy, shifted, len(dT), L.host_i32(dlo),
                   L.host_i32(dhi), L.host_i32(dT), L.ptr(w["tok_win"]), L.ptr(w["tok_level"]), L.ptr(w["tok_slot"]),
                   L.ptr(w["tok_pos"]), L.ptr(w["csr_tok"]), L.ptr(w["win_start"]), L.ptr(w["win_len"]),
                   L.ptr(w["counts"]), L.ptr(ws), wsb, st)
            w["_ws"] = ws
            wins.append(w)
        # transposed rulebook of the submanifold conv = tap-reversed rulebook (built here, with the plan, so that it is
        # off the training stream's critical path)
        nbr_subm_t = torch.flip(nbr_subm.view(cap, 9), dims=[1]).contiguous().view(-1)
        # full-resolution sites under every token (the decoder's ConvTranspose2d(k = s, stride = s) scatters there)
        up_s = gy // Y
        up_sites = upsample_cells(tok_cell, Y, X, up_s) if (up_s > 1 and up_s * Y == gy and up_s * X == gx) else None
        raw.append(dict(B=B, Y=Y, X=X, cap=cap, tok_cell=tok_cell, map=smap, n_tok=n_tok, nbr_subm=nbr_subm,
                        nbr_subm_t=nbr_subm_t, up_s=up_s, up_sites=up_sites, nbr_down=nbr_down, nbr_down_t=nbr_down_t, wins=wins))
    dec = None
    if dec_sources is not None:
        # active tiles of the decoder's 3x3 conv (geometry only: built here, off the training stream)
        srcs = [raw[int(i)] for i in dec_sources]
        if all(r["up_s"] * r["Y"] == gy and r["up_s"] * r["X"] == gx and r["up_s"] in (1, 2, 4, 8) for r in srcs):
            dec = _decoder_tiles_launch([r["map"] for r in srcs], [r["up_s"] for r in srcs], B, gy, gx, dev)
            dec["sources"] = tuple(int(i) for i in dec_sources)
    counts = torch.cat([r["n_tok"] for r in raw] + [w["counts"] for r in raw for w in r["wins"]]
                       + ([dec["n_act"]] if dec is not None else []))
    return dict(stages=raw, mask=mask, tok_pillar=tok_pillar, masked=keep_frac is not None, counts=counts,
                keep=(noise, scan_ws), dec=dec)


def _encoder_finalize(e: dict, allc, M: int) -> EncoderPlan:
    raw = e["stages"]
    ns = len(raw)
    stages = []
    for si, r in enumerate(raw):
        n = int(allc[si])
        wps = []
        for k, w in enumerate(r["wins"]):
            c = allc[ns + 8 * (2 * si + k): ns + 8 * (2 * si + k) + 8]
            assert c[7] == n, (c, n)
            nw = int(c[6])
            wps.append(WindowPlan(w["tok_win"][:n], w["tok_level"][:n], w["tok_slot"][:n], w["tok_pos"][:n],
                                  w["csr_tok"][:n], w["win_start"][:nw], w["win_len"][:nw],
                                  [int(v) for v in c[0:3]], [int(v) for v in c[3:6]], list(w["T"])))
        n_prev = int(allc[si - 1]) if r["nbr_down"] is not None else 0
        stages.append(StagePlan(r["B"], r["Y"], r["X"], n, r["tok_cell"][:n], r["map"],
                                r["nbr_subm"][:n * 9].view(n, 9),
                                None if r["nbr_down"] is None else r["nbr_down"][:n * 9].view(n, 9),
                                None if r["nbr_down_t"] is None else r["nbr_down_t"][:n_prev * 9].view(n_prev, 9),
                                wps))
        stages[-1]._nbr_subm_t = r["nbr_subm_t"][:n * 9].view(n, 9)
        stages[-1]._up_sites = None if r["up_sites"] is None else (r["up_s"], r["up_sites"][:n].reshape(-1))
    dec = e.get("dec", None)
    dt = None
    if dec is not None:
        n_act = int(allc[ns + 16 * ns])
        dt = DecoderTiles(dec["sources"], dec["B"], dec["H"], dec["W"], n_act, dec["tile_slot"], dec["tile_list"][:n_act])
    return EncoderPlan(e["mask"][:M] if e["masked"] else None, e["tok_pillar"][:stages[0].n_tok], stages, dt)


def encoder_plan(vox: VoxelPlan, strides, window_shapes, drop_infos, keep_frac: Optional[float] = None,
                 noise: Optional[torch.Tensor] = None, dec_sources=None) -> EncoderPlan:
    """Masking + token sets + rulebooks + window partitions for all stages; ONE host sync at the end.

    strides: per stage conv_down stride (1 or 2); window_shapes: per stage [wx, wy, wz];
    drop_infos: per stage DROP_INFO['train'] dict; keep_frac = 1 - MASK RATIO (python double) or None
    for no masking (fine-tune backbone); noise: (M,) fp32 masking noise (drawn if None).
    """
    if noise is not None:
        assert noise.shape == (vox.M,)
    e = _encoder_launch(vox, vox.M, strides, window_shapes, drop_infos, keep_frac, noise, dec_sources)
    return _encoder_finalize(e, e["counts"].tolist(), vox.M)   # the one host sync of this phase


# ------------------------------------------------------------------------------------------------
# the whole plan as ONE call of the library (csrc/plan.hip: gdmae_geometry_plan)
# ------------------------------------------------------------------------------------------------
import collections
import time as _time

EVENT_WAIT_S = 0.0     # seconds the host has spent blocked in PlanPrefetch.finish() waiting for a plan's counts (bench.py reads it)

_PLAN_SHAPES = collections.OrderedDict()   # shape key -> (PlanParams, {name: (offset, bytes)}, arena bytes, per-stage geometry): LRU
_PLAN_SHAPES_MAX = 32
_PLAN_GRANULE = 16384  # point capacity of a layout: the batch's point count rounded up to this (real batches differ in n0 every step)
_PINNED = {}           # device index -> free pinned int32 buffers for the count read-back (a prefetch owns one until finish())


def _plan_shape(n0, ncols, B, pcr, voxel_size, grid_size, strides, window_shapes, drop_infos, keep_frac, dec_sources):
    """Layout of the plan arena for the CAPACITY bucket of ``n0`` points (the layout is a pure function of capacities; the exact
    point count travels with each call).  Bounded LRU: a layout is ~30 KB of host memory and real data produces a new point
    count with almost every batch."""
    gx, gy, gz = (int(g) for g in grid_size)
    lo = tuple(float(v) for v in pcr[:3])
    vs = tuple(float(v) for v in voxel_size)
    drops = tuple(tuple(map(tuple, _drop_arrays(d))) for d in drop_infos)
    n_exact = n0
    n0 = max(1, (n0 + _PLAN_GRANULE - 1) // _PLAN_GRANULE) * _PLAN_GRANULE
    key = (n0, ncols, B, lo, vs, (gx, gy, gz), tuple(int(s) for s in strides), tuple(tuple(int(v) for v in w) for w in window_shapes),
           drops, None if keep_frac is None else float(keep_frac), None if dec_sources is None else tuple(int(i) for i in dec_sources))
    hit = _PLAN_SHAPES.get(key)
    if hit is not None:
        _PLAN_SHAPES.move_to_end(key)
        return hit
    del n_exact
    import ctypes as C
    ns = len(strides)
    assert gz == 1, "the SST backbone works on single-layer pillar grids (spt_backbone_mae.py:94)"
    assert 1 <= ns <= 4
    P = L.PlanParams()
    P.n_points, P.cap_points, P.n_cols, P.batch_size = n0, n0, ncols, B      # n_points is set per call (PlanPrefetch)
    for i in range(3):
        P.lo[i], P.vs[i] = lo[i], vs[i]
        P.grid[i] = (gx, gy, gz)[i]
    P.n_stages = ns
    geo = []
    Y, X = gy, gx
    cap = max(1, min(n0, B * gx * gy))
    m_cap = cap
    for i in range(ns):
        wx, wy, wz = (int(v) for v in window_shapes[i])
        assert wz == 1
        dlo, dhi, dT = drops[i]
        assert wx * wy <= max(dT), "token drop would not be the identity: unsupported (SURVEY header item 5)"
        P.stride[i], P.win_x[i], P.win_y[i], P.n_levels[i] = int(strides[i]), wx, wy, len(dT)
        for l in range(len(dT)):
            P.drop_lo[i][l], P.drop_hi[i][l], P.max_tokens[i][l] = dlo[l], dhi[l], dT[l]
        cap_in = cap
        if int(strides[i]) > 1:
            assert int(strides[i]) == 2, "only the k3 s2 p1 strided sparse conv of the shipped configs is implemented"
            Y, X = (Y - 1) // 2 + 1, (X - 1) // 2 + 1
            cap = min(4 * cap, B * Y * X)
        us = gy // Y
        up_s = us if (us >= 1 and us * Y == gy and us * X == gx) else 0
        geo.append(dict(Y=Y, X=X, cap=cap, cap_in=cap_in, up_s=up_s, strided=int(strides[i]) > 1, T=list(dT)))
    P.masked = int(keep_frac is not None)
    P.keep_frac = float(keep_frac) if keep_frac is not None else 1.0
    P.n_dec = 0 if dec_sources is None else len(dec_sources)
    for g in range(P.n_dec):
        P.dec_sources[g] = int(dec_sources[g])
    P.want_pm = 1
    table = (L.PlanBuffer * 160)()
    n_ent, total = C.c_int(0), C.c_size_t(0)
    L.call("gdmae_geometry_plan_layout", C.byref(P), table, 160, C.byref(n_ent), C.byref(total))
    names = {table[i].name.decode(): (int(table[i].offset), int(table[i].bytes)) for i in range(n_ent.value)}
    hit = _PLAN_SHAPES[key] = (P, names, int(total.value), geo, m_cap, (lo, vs, (gx, gy, gz)))
    while len(_PLAN_SHAPES) > _PLAN_SHAPES_MAX:
        _PLAN_SHAPES.popitem(last=False)
    return hit


class _Arena:
    """Typed views into the plan's single device allocation (one ``as_strided`` per buffer)."""

    def __init__(self, raw, names):
        self.raw, self.names = raw, names
        self.base = {torch.int32: raw.view(torch.int32), torch.float32: raw.view(torch.float32), torch.int64: raw.view(torch.int64)}

    def has(self, name):
        return name in self.names

    def view(self, name, dtype, *shape):
        off, _ = self.names[name]
        es = 8 if dtype == torch.int64 else 4
        strides, acc = [], 1
        for d in reversed(shape):
            strides.append(acc)
            acc *= int(d)
        return self.base[dtype].as_strided(tuple(int(d) for d in shape), tuple(reversed(strides)), off // es)


class PlanPrefetch:
    """Geometry plan of a batch built ahead of time on a side stream (the analogue of a data-loader prefetch: the
    plan depends only on the input points, never on the weights).  The whole plan is ONE call of the library
    (``gdmae_geometry_plan``, csrc/plan.hip: ~32 launches into one arena allocation), the data-dependent counts are copied
    to pinned host memory asynchronously, and ``finish()`` only waits for THAT stream - so the training step of batch t+1
    never stalls the host behind the backward of batch t, and the host can run a full step ahead of the GPU."""

    _side = {}
    _arena_bytes = {}        # device index -> bytes of every plan arena allocated there (monotone)

    def __init__(self, points, point_cloud_range, voxel_size, grid_size, batch_size, strides, window_shapes, drop_infos,
                 keep_frac=None, noise=None, ready=None, dec_sources=None):
        """The plan stream is always ordered after everything queued on the calling stream so far (see the comment below for
        why that ordering is what keeps the plan tensors safe without ``record_stream``).  ``ready``: an ADDITIONAL event the
        plan stream waits for - the H2D copy of ``points`` when it was issued on a copy stream the calling stream has not
        waited for yet."""
        import ctypes as C
        assert points.is_cuda and points.dtype == torch.float32 and points.dim() == 2
        points = points.contiguous()
        dev = points.device
        main = torch.cuda.current_stream(dev)
        side = PlanPrefetch._side.setdefault(dev.index, torch.cuda.Stream(device=dev))
        # The plan stream is ordered after everything queued on the main stream so far.  That makes `points` / `noise` visible
        # and - the reason it is unconditional - lets the plan arena live without `record_stream`: it is allocated from the plan
        # stream's pool and read by the main stream; once its last reference dies the allocator hands the block to LATER
        # plan-stream allocations only, i.e. to a prefetch whose kernels wait here for all main-stream work that could still
        # read it.  (With `record_stream` every freed block cost an event record on the training stream.)  `ready` adds a
        # wait, it never replaces this one.
        side.wait_stream(main)
        if ready is not None:
            side.wait_event(ready)
        points.record_stream(side)
        if noise is not None:
            noise.record_stream(side)
        n0, ncols = points.shape
        P, names, total, geo, m_cap, _ = self.shape = _plan_shape(n0, ncols, int(batch_size), point_cloud_range, voxel_size, grid_size,
                                                                    strides, window_shapes, drop_infos, keep_frac, dec_sources)
        self.masked, self.dec_sources, self.batch_size, self.ncols = keep_frac is not None, dec_sources, int(batch_size), ncols
        with torch.cuda.stream(side):
            # every arena of a device has the SAME size - the largest layout seen so far, in 64 MB steps: the caching allocator then
            # hands a freed arena to the next prefetch whole.  With the exact `total` of each capacity bucket it split a cached block for
            # a slightly smaller request and had nothing left for the next larger one: 21 hipMalloc calls (device-synchronising) in 50
            # timed steps of round 5's bench
            cap = PlanPrefetch._arena_bytes.get(dev.index, 0)
            if total > cap:
                cap = PlanPrefetch._arena_bytes[dev.index] = -(-total // (64 << 20)) * (64 << 20)
            raw = torch.empty(cap, dtype=torch.uint8, device=dev)
            if self.masked and noise is None:
                noise = torch.rand(m_cap, device=dev, dtype=torch.float32)
            if noise is not None:
                assert noise.dtype == torch.float32 and noise.numel() >= 1
                noise = noise.contiguous()
            Pc = L.PlanParams.from_buffer_copy(P)                  # the cached struct describes the capacity bucket
            Pc.n_points = n0
            L.call("gdmae_geometry_plan", C.byref(Pc), L.ptr(points), None if noise is None else L.ptr(noise), L.ptr(raw), total, L.stream())
            self.arena = _Arena(raw, names)
            nc = names["counts"][1] // 4
            assert nc <= 256
            free = _PINNED.setdefault(dev.index, [])
            # this prefetch OWNS its pinned buffer until finish() hands it back: any number of prefetches may be in flight
            self._pinned = free.pop() if free else torch.empty(256, dtype=torch.int32).pin_memory()
            self.host = self._pinned[:nc]
            self.host.copy_(self.arena.view("counts", torch.int32, nc), non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(side)
        self.side, self.keep, self.points = side, noise, points
        self.serial = side == main          # plan issued on the calling stream itself (A/B switch of tools/phase_times.py)

    def finish(self):
        """-> (VoxelPlan, EncoderPlan); the current (main) stream is ordered after the plan stream."""
        # invariant behind the missing record_stream calls (see __init__): the plan arena is only ever ALLOCATED on the plan
        # stream inside __init__ and only ever CONSUMED on the stream that issues the prefetches
        assert self.serial or torch.cuda.current_stream() != self.side, "plan tensors must not be consumed on the plan stream"
        global EVENT_WAIT_S
        t_w = _time.perf_counter()
        self.event.synchronize()
        EVENT_WAIT_S += _time.perf_counter() - t_w          # host blocked on the plan stream (it runs ahead of the GPU otherwise)
        torch.cuda.current_stream().wait_event(self.event)
        c = self.host.tolist()
        # from here on the plan lives in the returned objects only (vox._arena keeps the allocation): this handle lets go of the arena,
        # the input points and the noise, so that whoever still holds the handle does not hold 0.8 GB with it
        self.points_device = self.points.device
        self.host = None
        if self._pinned is not None:                               # read: the buffer may serve the next prefetch
            _PINNED.setdefault(self.points_device.index, []).append(self._pinned)
            self._pinned = None
        P, names, total, geo, m_cap, (lo, vs, grid) = self.shape
        A, V = self.arena, self.arena.view
        i32, f32, i64 = torch.int32, torch.float32, torch.int64
        B, ncols = self.batch_size, self.ncols
        F = ncols - 1
        N, M = int(c[0]), int(c[1])
        gx, gy, gz = grid
        vox = VoxelPlan(B, grid, lo, vs, ncols, N, M, V("points", f32, N, ncols), V("point_coords", i64, N, 4), V("inverse", i64, N),
                        V("inverse32", i32, N), V("voxel_coords", i64, M, 4), V("pillar_cell", i32, M), V("pt_off", i32, M + 1),
                        V("pillar_pts", i32, N), V("point_rank", i32, N), V("sample_off", i32, B + 1), V("pillar_mean", f32, M, F),
                        V("cell2pillar", i32, B * gx * gy * gz), V("counts", i32, 2), V("points_pm", f32, N, ncols), V("row_pillar", i32, N))
        vox._arena = A                                           # keeps the allocation alive with the plan
        ns = len(geo)
        n_vis = int(c[2 + ns + 16 * ns + 1])
        stages = []
        for i, g in enumerate(geo):
            n = int(c[2 + i])
            pre = f"s{i}."
            wps = []
            for k in range(2):
                wc = c[2 + ns + 16 * i + 8 * k: 2 + ns + 16 * i + 8 * k + 8]
                assert wc[7] == n, (wc, n)
                nw = int(wc[6])
                wpre = f"{pre}w{k}."
                wps.append(WindowPlan(V(wpre + "tok_win", i32, n), V(wpre + "tok_level", i32, n), V(wpre + "tok_slot", i32, n),
                                      V(wpre + "tok_pos", i32, n), V(wpre + "csr_tok", i32, n), V(wpre + "win_start", i32, nw),
                                      V(wpre + "win_len", i32, nw), [int(v) for v in wc[0:3]], [int(v) for v in wc[3:6]], list(g["T"])))
            n_prev = (int(c[2 + i - 1]) if i > 0 else n_vis) if g["strided"] else 0
            sp = StagePlan(B, g["Y"], g["X"], n, V(pre + "tok_cell", i32, n), V(pre + "map", i32, B * g["Y"] * g["X"]),
                           V(pre + "nbr_subm", i32, n, 9), V(pre + "nbr_down", i32, n, 9) if g["strided"] else None,
                           V(pre + "nbr_down_t", i32, n_prev, 9) if g["strided"] else None, wps)
            sp._nbr_subm_t = V(pre + "nbr_subm_t", i32, n, 9)
            sp._up_sites = (g["up_s"], V(pre + "up_sites", i32, n * g["up_s"] * g["up_s"])) if g["up_s"] > 1 else None
            stages.append(sp)
        dt = None
        if A.has("dec.tile_slot"):
            n_act = int(c[2 + ns + 16 * ns])
            nt = B * ((gy + 7) // 8) * ((gx + 7) // 8)
            nbrs = [V(f"dec.nbr{g}", i32, stages[int(si)].n_tok * geo[int(si)]["up_s"] ** 2, 9) for g, si in enumerate(self.dec_sources)]
            dt = DecoderTiles(tuple(int(i) for i in self.dec_sources), B, gy, gx, n_act, V("dec.tile_slot", i32, nt), V("dec.tile_list", i32, n_act),
                              nbrs)
        ep = EncoderPlan(V("mask", f32, M) if self.masked else None, V("tok_pillar", i32, stages[0].n_tok if not geo[0]["strided"] else n_vis),
                         stages, dt)
        self.arena = self.points = self.keep = None
        return vox, ep


def _tensors_of(obj):
    out = []
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            out.append(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            out += _tensors_of(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            out += _tensors_of(v)
    return out
