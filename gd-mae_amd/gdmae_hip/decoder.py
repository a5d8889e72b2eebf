"""Sparse-aware generative decoder: the same arithmetic as the reference's dense decoder
(pcdet/models/backbones_3d/spt_backbone_mae.py:120-143: densify -> ConvTranspose2d(k = s) + BN2d + ReLU per
source stage -> cat -> Conv2d 3x3 + BN2d + ReLU -> gather at all pillar sites), restructured around what
is actually dense.

Exact identities used (no approximation):
* ConvTranspose2d with kernel = stride maps each input site to its own s x s output block, so on a map
  that is zero outside the active set it is one token GEMM  P = X @ W.view(Cin, s*s*Cout)  and zeros
  elsewhere.
* BatchNorm2d statistics over ALL B*Y*X sites of such a map follow from the token values alone
  (sum and sum of squares over tokens, divided by the dense count); after BN + ReLU every empty site
  holds the same per-channel constant relu(beta - gamma*mean/sigma).
* The concatenated 384-channel map is therefore "constant background + token rows": it is written once
  (fill + three row scatters into column slices) instead of 3 dense deconvs, 3 dense BN passes, 3 ReLUs
  and a cat copy.
* The forward 3x3 convolution is the one genuinely dense contraction.  In throughput mode (bf16) it is the library's
  own bf16-MFMA implicit GEMM over the ACTIVE 8x8 TILES of the map (csrc/conv_tiles.hip): the 384-channel input map
  is never written (the kernel gathers the deconvolution rows through the stages' cell -> token maps and applies
  BN + ReLU on the way into LDS), sites outside the active tiles hold one of 9 border-class constants, and the
  BatchNorm statistics of the output come out of the same kernel.  In fp32 parity mode the map is materialised
  (fill + three row scatters) and convolved by F.conv2d.  Either way the output is only consumed at the M pillar
  sites, so BN + ReLU are applied to the gathered rows only.
* In the backward the gradient of the conv output is an affine function of the conv output plus M sparse
  rows, and the conv input gradient is only needed where the input is not background: per source stage the
  9 shifted output-gradient rows of its active sites are gathered (gdmae_conv3x3_grad_taps) and two GEMMs give
  the input-gradient rows and the weight gradient; the background's share (input gradient summed over all
  other sites, weight gradient against the constant) is a closed form of 9 border-region sums.

Autograd: one hand-derived Function (DecoderHead) from the token-side deconvolution rows to the pillar rows.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

import os

from . import bn as gbn
from . import lib as L
from . import ops, packing
from . import plan as gplan
from . import timing

# conv_out backward of the tile path: "implicit" = tile-compact output gradient + implicit-GEMM input gradient + gathered grouped
# weight gradient (no tap matrix, no library GEMM); "taps" = the round-2 schedule (gdmae_conv3x3_grad_taps + two library GEMMs)
CONV_BWD = os.environ.get("GDMAE_DEC_BWD", "implicit")

BN_EPS, BN_MOM = 1e-3, 0.01


# sparse_decoder(..., pre_conv_hook=callable): invoked once right before the decoder's tile convolution is launched (the one
# matrix-core-bound launch of the step) - the training loop issues the NEXT batch's geometry plan there, so that the plan's
# atomics-and-scatter kernels on the side stream run under it instead of under the memory-bound start of the forward (bench.py --plan-at
# conv).  The hook travels with the call (SPTBackboneMAE.prefetch_plan_under_decoder keeps it on the module until its forward), there is
# no process-global state.

def colstats(x2d: torch.Tensor):
    """(sum, sumsq) per column of a contiguous (R, C) fp32/bf16 device matrix, as float64 (C,) tensors."""
    R, C = x2d.shape
    assert x2d.is_contiguous() and x2d.dtype in (torch.float32, torch.bfloat16)
    out = torch.empty(2 * C, dtype=torch.float64, device=x2d.device)
    ws = torch.empty(L.load().gdmae_colstats_workspace_bytes(C), dtype=torch.uint8, device=x2d.device)
    L.call("gdmae_colstats", L.ptr(x2d), R, C, int(x2d.dtype == torch.bfloat16), L.ptr(out), L.ptr(ws), L.stream())
    return out[:C], out[C:]


def _gather_slice(table2d, idx, col0, ncol):
    es = table2d.element_size()
    out = torch.empty(idx.numel(), ncol, dtype=table2d.dtype, device=table2d.device)
    L.call("gdmae_gather_rows_strided", L.ptr(table2d), L.ptr(idx), idx.numel(), ncol * es, table2d.shape[1] * es,
           col0 * es, L.ptr(out), L.stream())
    return out


def _scatter_slice(src2d, idx, table2d, col0):
    es = table2d.element_size()
    src2d = src2d.contiguous()
    assert src2d.dtype == table2d.dtype
    L.call("gdmae_scatter_rows_strided", L.ptr(src2d), L.ptr(idx), idx.numel(), src2d.shape[1] * es,
           table2d.shape[1] * es, col0 * es, L.ptr(table2d), L.stream())


def _bf(t):
    return int(t.dtype == torch.bfloat16)


# S_k = sum of the output gradient over the sites s whose tap s+k lies inside the map, from the 9 region sums
# [all, row 0, row H-1, col 0, col W-1, corner(0,0), corner(0,W-1), corner(H-1,0), corner(H-1,W-1)]
_TAP_REGION = torch.zeros(9, 9, dtype=torch.float64)
for _ky in (-1, 0, 1):
    for _kx in (-1, 0, 1):
        _k = (_ky + 1) * 3 + (_kx + 1)
        _TAP_REGION[_k, 0] = 1
        if _ky:
            _TAP_REGION[_k, 1 if _ky < 0 else 2] = -1
        if _kx:
            _TAP_REGION[_k, 3 if _kx < 0 else 4] = -1
        if _ky and _kx:
            _TAP_REGION[_k, 5 + (0 if _ky < 0 else 2) + (0 if _kx < 0 else 1)] = 1


_REGION_CONSTS = {}


def _region_consts(dev, B, H, W):
    """Device copies of the tap/region matrix and the region site counts (cached: a host->device copy inside the
    backward would stall the host until the GPU has caught up)."""
    key = (dev, B, H, W)
    if key not in _REGION_CONSTS:
        R = B * H * W
        cnt = torch.tensor([R, B * W, B * W, B * H, B * H, B, B, B, B], dtype=torch.float64)
        _REGION_CONSTS[key] = (_TAP_REGION.to(dev), cnt.to(dev))
    return _REGION_CONSTS[key]


class DecoderHead(torch.autograd.Function):
    """The whole generative-decoder head on the token rows (reference spt_backbone_mae.py:30-52,125-135):
    ConvTranspose2d(k=s) outputs P_i of the active sites -> BatchNorm2d(train)+ReLU of the (implicit) dense maps ->
    channel concat Z (R, sum C_i) channels-last -> dense 3x3 conv_out (MIOpen) -> BatchNorm2d(train)+ReLU evaluated
    only at the pillar sites.

    forward(geom, conv_w, gamma2, beta2, pillar_cell, cell2pillar, sites_0, P_0, gamma_0, beta_0, sites_1, ...)
      geom = (B, H, W, eps1, eps2, compute dtype).  Every non-active site of map i is zero before its BatchNorm, so
      the batch statistics over all R sites are column sums of P_i and every other site holds the constant
      background relu(beta - gamma*mean*rstd).  Returns (out (M, C2) fp32, Y (R, C2) [non-differentiable, for the
      dense spatial_features], mean2, var2, mean_0, var_0, ...).
    backward: hand-derived.  The conv output gradient is dY = k0 + k1*Y + rows-at-pillar-sites (BatchNorm2d chain
      rule), which is dense - but the input gradient is only needed at the active sites of each stage (everything
      else is the shared background whose gradient is a border-corrected closed form), and the weight gradient
      splits into background x region-sums plus active-site rows.  So instead of MIOpen's dense backward-data and
      backward-weights over all R sites, gdmae_conv3x3_grad_taps gathers the 9 shifted dY rows of the active sites
      and two GEMMs per stage finish the job; dY / dZ are never materialised."""

    last_ybg = None      # class constants of the latest tile-path forward (for the optional dense expansion)

    @staticmethod
    def forward(ctx, geom, conv_w, gamma2, beta2, pillar_cell, cell2pillar, bn_mods, *args):
        B, H, W, eps1, eps2, cdt = geom[:6]
        R = B * H * W
        k = len(args) // 4
        sites, Ps, gammas, betas = args[0::4], args[1::4], args[2::4], args[3::4]
        widths = [int(P.shape[1]) for P in Ps]
        dev = Ps[0].device
        bns = list(bn_mods) if bn_mods is not None else [None] * (k + 1)
        ab_l, stats_l, mv_out = [], [], []
        for i in range(k):
            stats, ab, mv = gbn.fold(Ps[i].contiguous(), R, gammas[i], betas[i], eps1, bns[i])
            ab_l.append(ab), stats_l.append(stats)
            mv_out += [mv[:widths[i]], mv[widths[i]:]]
        tiles = geom[6] if len(geom) > 6 else None
        use_tiles = (tiles is not None and cdt == torch.bfloat16 and all(w == 128 for w in widths) and conv_w.shape[0] == 128
                     and conv_w.dtype == torch.float32 and conv_w.is_contiguous() and all(P.dtype == torch.bfloat16 for P in Ps))
        if use_tiles:
            # own bf16-MFMA 3x3 convolution over the active tiles; the dense 384-channel map is never built
            dt, maps, ups = tiles
            lib = L.load()
            Cin = sum(widths)
            C2 = 128
            Wp = torch.empty(lib.gdmae_conv3x3_tiles_packed_bytes(k), dtype=torch.uint8, device=dev)
            bgz = torch.empty(Cin, dtype=cdt, device=dev)
            ybg = torch.empty(9, C2, dtype=cdt, device=dev)
            a_l, b_l = [ab[:w] for ab, w in zip(ab_l, widths)], [ab[w:] for ab, w in zip(ab_l, widths)]
            L.call("gdmae_conv3x3_tiles_pack", L.ptr(conv_w), C2, Cin, L.host_ptrs(b_l), k, L.ptr(Wp), L.ptr(bgz), L.ptr(ybg),
                   L.stream())
            y2 = torch.empty(max(dt.n_act, 1) * 64, C2, dtype=cdt, device=dev)
            stats2 = torch.empty(2 * C2, dtype=torch.float64, device=dev)
            ab2 = torch.empty(2 * C2, dtype=torch.float32, device=dev)
            mv2 = torch.empty(2 * C2, dtype=torch.float32, device=dev)
            ws = torch.empty(lib.gdmae_conv3x3_tiles_workspace_bytes(dt.n_act), dtype=torch.uint8, device=dev)
            bn2 = bns[k]
            rm = rv = nb = None
            mom = 0.0
            if bn2 is not None and bn2.training and bn2.running_mean is not None:
                rm, rv, nb, mom = bn2.running_mean, bn2.running_var, bn2.num_batches_tracked, float(bn2.momentum)
            flops = 2.0 * dt.n_act * 64 * C2 * 9 * Cin
            hook = geom[7] if len(geom) > 7 else None
            if hook is not None:                 # work to issue (on another stream) right before the tile convolution
                hook()
            with timing.kernel("k_conv3x3_tiles", dt.n_act * 64 * (Cin + C2) * 2, flops,
                               {"dense_equivalent_TFLOPs_per_launch": round(2.0 * R * C2 * 9 * Cin / 1e12, 4), "active_tiles": dt.n_act}):
                L.call("gdmae_conv3x3_tiles_fwd", L.host_ptrs([P.contiguous() for P in Ps]), L.host_ptrs(maps), L.host_ptrs(a_l),
                       L.host_ptrs(b_l), L.host_i32(ups), k, L.ptr(Wp), L.ptr(ybg), L.ptr(dt.tile_list), dt.n_act, B, H, W,
                       L.ptr(y2), L.ptr(gamma2), L.ptr(beta2), float(eps2), mom, L.ptr(rm), L.ptr(rv), L.ptr(nb), L.ptr(stats2),
                       L.ptr(ab2), L.ptr(mv2), L.ptr(ws), L.stream())
            M = pillar_cell.numel()
            yrows = torch.empty(M, C2, dtype=cdt, device=dev)
            L.call("gdmae_tiles_gather_rows", L.ptr(y2), L.ptr(dt.tile_slot), L.ptr(ybg), L.ptr(pillar_cell), M, H, W, C2, 2,
                   L.ptr(yrows), L.stream())
            Z, tile_slot = None, dt.tile_slot
            DecoderHead.last_ybg = ybg
        else:
            bgz = torch.relu(torch.cat([ab[w:] for ab, w in zip(ab_l, widths)])).to(cdt)   # value of every non-active site
            Z = torch.empty(R, sum(widths), dtype=cdt, device=dev)
            L.call("gdmae_fill_rows", L.ptr(bgz), R, Z.shape[1], Z.element_size(), L.ptr(Z), L.stream())
            col = 0
            for i in range(k):
                w = widths[i]
                L.call("gdmae_rows_affine_relu_scatter", L.ptr(Ps[i]), _bf(Ps[i]), L.ptr(sites[i]), Ps[i].shape[0], w,
                       L.ptr(ab_l[i]), L.ptr(ab_l[i][w:]), L.ptr(Z), _bf(Z), Z.shape[1], col, L.stream())
                col += w
            # the dense convolution of the materialised map through csrc/conv_dense.hip: bf16 operands in the throughput mode, the fp32-grade
            # three-term form in the parity mode (F.conv2d -> MIOpen before round 5)
            from . import dense as gdense
            C2o = conv_w.shape[0]
            if cdt == torch.bfloat16 and Z.shape[1] % 64 == 0 and conv_w.dtype == torch.float32:
                y2 = gdense.Conv3x3Dense.apply(Z.view(B, H, W, -1).permute(0, 3, 1, 2), conv_w.detach(), None, 1, None).permute(0, 2, 3, 1)
            elif cdt == torch.float32 and Z.shape[1] % 64 == 0:
                y2 = gdense.conv3x3_f32_rows(Z.view(B, H, W, -1), conv_w.detach().float().contiguous(), None, 1)[..., :C2o]
            else:
                wc = ops.shadow(conv_w, cdt)
                with torch.autocast("cuda", enabled=False):
                    y2 = F.conv2d(Z.view(B, H, W, -1).permute(0, 3, 1, 2), wc, None, 1, 1).permute(0, 2, 3, 1)
            if not y2.is_contiguous():
                y2 = y2.contiguous()
            y2 = y2.view(R, -1)
            C2 = y2.shape[1]
            stats2, ab2, mv2 = gbn.fold(y2, R, gamma2, beta2, eps2, bns[k])
            yrows = ops.gather_rows_raw(y2, pillar_cell)                       # (M, C2) conv outputs at the pillar sites
            M = yrows.shape[0]
            tile_slot = ybg = None
        out = torch.empty(M, C2, dtype=torch.float32, device=dev)
        L.call("gdmae_rows_affine_relu_scatter", L.ptr(yrows), _bf(yrows), None, M, C2, L.ptr(ab2), L.ptr(ab2[C2:]), L.ptr(out), 0,
               C2, 0, L.stream())
        ctx.save_for_backward(*sites, *Ps, *ab_l, *stats_l, *[g.detach() for g in gammas], Z, y2, bgz, yrows, ab2, stats2,
                              pillar_cell, cell2pillar, gamma2.detach(), conv_w.detach(), tile_slot, ybg)
        ctx.k, ctx.widths, ctx.geom = k, widths, geom[:7]      # (without the one-shot hook: the graph must not keep a plan handle alive)
        ctx.direct = [gbn.direct_pair(g, be) for g, be in zip(gammas, betas)] + [gbn.direct_pair(gamma2, beta2)]
        ctx.direct_w = ops.direct_grad(conv_w)
        ctx.conv_param = conv_w        # the parameter object itself: the packed images of the backward are registered on its identity
        mean2, var2 = mv2[:C2], mv2[C2:]
        ctx.mark_non_differentiable(y2, mean2, var2, *mv_out)
        ctx.set_materialize_grads(False)      # no zero tensors for the unused gradients of those outputs
        return (out, y2, mean2, var2, *mv_out)

    @staticmethod
    def backward(ctx, dout, *_):
        k = ctx.k
        B, H, W, eps1, eps2, cdt = ctx.geom[:6]
        R = B * H * W
        sv = ctx.saved_tensors
        sites, Ps, ab_l, stats_l, gammas = (sv[i * k:(i + 1) * k] for i in range(5))
        Z, y2, bgz, yrows, ab2, stats2, pillar_cell, cell2pillar, gamma2, conv_w, tile_slot, ybg = sv[5 * k:]
        C2, Cin = y2.shape[1], bgz.numel()
        M = yrows.shape[0]
        dev = y2.device
        f64 = torch.float64
        # ---- BatchNorm2d #2 (+ReLU) at the pillar rows: dY = c0 + c1*Y + rows[pillar sites], rows = a*dh
        dout = dout.float().contiguous()
        st2 = torch.empty(3 * C2, dtype=f64, device=dev)
        ws = torch.empty(L.load().gdmae_rows_bwd_stats_workspace_bytes(C2), dtype=torch.uint8, device=dev)
        L.call("gdmae_rows_bwd_stats", L.ptr(yrows), _bf(yrows), None, M, C2, L.ptr(ab2), L.ptr(ab2[C2:]), L.ptr(dout), 0, C2, 0,
               L.ptr(st2), L.ptr(ws), L.stream())
        dgamma2, dbeta2, k01 = gbn.bwd_coeffs(st2, 3, stats2, ab2, gamma2, R, None, ctx.direct[k])
        zero = torch.zeros(2 * C2, dtype=torch.float32, device=dev)
        rows = torch.empty(M, C2, dtype=torch.float32, device=dev)
        L.call("gdmae_rows_bwd", L.ptr(yrows), _bf(yrows), None, M, C2, L.ptr(ab2), L.ptr(ab2[C2:]), L.ptr(zero), L.ptr(zero[C2:]),
               L.ptr(dout), 0, C2, 0, L.ptr(rows), 0, L.stream())
        # ---- region sums of dY (all / border rows / border columns / corners) -> S_k per tap
        reg = torch.empty(16, C2, dtype=f64, device=dev)
        ws = torch.empty(L.load().gdmae_border_sums_workspace_bytes(B, C2), dtype=torch.uint8, device=dev)
        L.call("gdmae_border_sums", L.ptr(y2), _bf(y2), L.ptr(tile_slot), L.ptr(ybg), L.ptr(rows), L.ptr(pillar_cell), M, B, H, W,
               C2, L.ptr(reg), L.ptr(ws), L.stream())
        # region 0 = all sites: sum Y = mean * R, sum rows = a * sum dh; S = tap_region @ (cnt k0 + regY k1 + regR), the
        # column sums of dZ over ALL sites, the background part of the weight gradient and the per-tap weight in the
        # compute dtype: one call (three launches) instead of ~20 small torch ops
        tap_region, cnt = _region_consts(dev, B, H, W)
        use_native = conv_w.dtype == torch.float32 and conv_w.is_contiguous() and C2 % 8 == 0
        if use_native:
            S = torch.empty(9, C2, dtype=f64, device=dev)
            tot = torch.empty(Cin, dtype=f64, device=dev)
            dWk = torch.empty(9, C2, Cin, dtype=torch.float32, device=dev)
            Wd = torch.empty(9, C2, Cin, dtype=cdt, device=dev)
            ws = torch.empty(L.load().gdmae_decoder_region_workspace_bytes(C2, Cin), dtype=torch.uint8, device=dev)
            L.call("gdmae_decoder_region_algebra", L.ptr(stats2), L.ptr(ab2), L.ptr(st2), L.ptr(k01), L.ptr(reg),
                   L.ptr(tap_region), L.ptr(cnt), float(R), C2, Cin, L.ptr(conv_w), L.ptr(bgz), _bf(Wd), L.ptr(S), L.ptr(tot),
                   L.ptr(dWk), L.ptr(Wd), L.ptr(ws), L.stream())
        else:
            k0, k1 = k01[:C2].double(), k01[C2:].double()
            regY = torch.cat([(stats2[:C2] * R)[None], reg[:8]])
            regR = torch.cat([(ab2[:C2].double() * st2[:C2])[None], reg[8:]])
            regD = cnt[:, None] * k0[None, :] + regY * k1[None, :] + regR
            S = tap_region @ regD                                     # (9, C2)
            Wk = conv_w.permute(2, 3, 0, 1).reshape(9, C2, Cin)              # W_k[o, i], k = (ky+1)*3 + (kx+1)
            tot = torch.einsum('ko,koi->i', S, Wk.double()).contiguous()      # column sums of dZ over ALL sites
            dWk = (S[:, :, None] * bgz.double()[None, None, :]).float()       # background part of the weight gradient
            Wd = Wk.to(cdt)
        grads = [None] * 7
        dW_rows = []
        col = 0
        # ---- tile path: the output gradient ONCE in Yc's tile-compact layout, rulebooks of the (active site, tap) rows
        implicit = None
        if (CONV_BWD == "implicit" and tile_slot is not None and y2.dtype == torch.bfloat16 and use_native and C2 == 128
                and all(w == 128 for w in ctx.widths) and len(ctx.geom) > 6 and ctx.geom[6] is not None):
            dt = ctx.geom[6][0]
            packed = packing.decoder_conv_packed(ctx.conv_param, ctx.widths, ctx.direct_w is not None)
            if packed is not None:
                if dt.nbr is None:
                    dt.nbr = gplan.decoder_site_rulebooks(dt, sites, ctx.geom[6][2])
                dYc = torch.empty_like(y2)
                L.call("gdmae_decoder_dy", L.ptr(y2), L.ptr(dt.tile_list), dt.n_act, L.ptr(k01), L.ptr(k01[C2:]), L.ptr(rows),
                       L.ptr(cell2pillar), H, W, C2, L.ptr(dYc), L.stream())
                implicit = (dt, packed, dYc)
        for i, w in enumerate(ctx.widths):
            P = Ps[i]
            n = P.shape[0]
            ab = ab_l[i]
            if implicit is not None:
                dt, packed, dYc = implicit
                nbr = dt.nbr[i]
                assert nbr.shape[0] == n, (nbr.shape, n)
                # dZ rows of the stage's active sites: implicit GEMM over the rulebook (9 gathered dY rows per site)
                dX = torch.empty(n, w, dtype=cdt, device=dev)
                L.call("gdmae_spconv", L.ptr(dYc), 0, L.ptr(nbr), packed.data_ptr() + i * 9 * w * C2 * 2, n, C2, w, L.ptr(dX), 8, L.stream())
                n_pad = int(L.load().gdmae_tap_dw_rows(n, w, C2))
                Zd = torch.empty(n_pad, w, dtype=cdt, device=dev)
                bg = bgz[col:col + w]
                L.call("gdmae_rows_affine_relu_sub", L.ptr(P), _bf(P), None, n, w, L.ptr(ab), L.ptr(ab[w:]), L.ptr(bg), L.ptr(Zd),
                       _bf(Zd), w, 0, L.stream())
                # weight gradient: dWk[k][co][col + ci] += sum_t dY[site_t - k][co] * Zd[t][ci], nine taps in one grouped launch
                ws = torch.empty(L.load().gdmae_tap_dw_workspace_bytes(n, w, C2), dtype=torch.uint8, device=dev)
                L.call("gdmae_tap_dw", L.ptr(Zd), n, n_pad, w, L.ptr(dYc), L.ptr(nbr), C2, L.ptr(dWk), Cin, col, L.ptr(ws), L.stream())
            else:
                G = torch.empty(n, 9 * C2, dtype=cdt, device=dev)
                with timing.kernel("k_conv_grad_taps", 2.0 * n * 9 * C2 * G.element_size()):
                    L.call("gdmae_conv3x3_grad_taps", L.ptr(y2), _bf(y2), L.ptr(tile_slot), L.ptr(ybg), L.ptr(k01), L.ptr(k01[C2:]),
                           L.ptr(rows), L.ptr(cell2pillar), L.ptr(sites[i]), n, H, W, C2, L.ptr(G), L.stream())
                dX = ops.mm(G, Wd[:, :, col:col + w].reshape(9 * C2, w))      # dZ rows of this stage's active sites
                if Z is not None:
                    Zrows = _gather_slice(Z, sites[i], col, w)
                    Zd = Zrows - bgz[col:col + w]
                else:                                                         # the map was never built: redo BN + ReLU of the rows
                    Zd = torch.empty(n, w, dtype=cdt, device=dev)
                    bg = bgz[col:col + w]
                    if w % 8 == 0 and bg.dtype == cdt:
                        L.call("gdmae_rows_affine_relu_sub", L.ptr(P), _bf(P), None, n, w, L.ptr(ab), L.ptr(ab[w:]), L.ptr(bg), L.ptr(Zd),
                               _bf(Zd), w, 0, L.stream())
                    else:
                        L.call("gdmae_rows_affine_relu_scatter", L.ptr(P), _bf(P), None, n, w, L.ptr(ab), L.ptr(ab[w:]), L.ptr(Zd),
                               _bf(Zd), w, 0, L.stream())
                        Zd = Zd - bg
                dW_rows.append(ops.splitk_tn(G, Zd))                          # (9*C2, w) fp32
                del G
            ws = torch.empty(L.load().gdmae_rows_bwd_stats_workspace_bytes(w), dtype=torch.uint8, device=dev)
            L.call("gdmae_rows_bwd_stats", L.ptr(P), _bf(P), None, n, w, L.ptr(ab), L.ptr(ab[w:]), L.ptr(dX), _bf(dX), w, 0,
                   None, L.ptr(ws), L.stream())
            dgamma, dbeta, c01 = gbn.bwd_coeffs((ws, L.load().gdmae_rows_bwd_stats_rows(n)), 3, stats_l[i], ab, gammas[i], R,
                                                tot[col:col + w], ctx.direct[i])
            dP = torch.empty_like(P)
            L.call("gdmae_rows_bwd", L.ptr(P), _bf(P), None, n, w, L.ptr(ab), L.ptr(ab[w:]), L.ptr(c01), L.ptr(c01[w:]), L.ptr(dX),
                   _bf(dX), w, 0, L.ptr(dP), _bf(dP), L.stream())
            grads += [None, dP, dgamma, dbeta]
            col += w
        if dW_rows:
            dWk = dWk + torch.cat(dW_rows, dim=1).view(9, C2, Cin)
        dW = dWk.permute(1, 2, 0).reshape(C2, Cin, 3, 3)
        if ctx.direct_w is not None:
            ctx.direct_w.add_(dW)
        else:
            grads[1] = dW.to(conv_w.dtype)
        grads[2], grads[3] = dgamma2, dbeta2
        return tuple(grads)


class DeconvRowsFn(torch.autograd.Function):
    """P (n * s*s, cout) = rows of ConvTranspose2d(k = s, stride s, no bias) at the active tokens (csrc/rows_gemm.hip): x (n, cin)
    bf16 token rows, weight (cin, cout, s, s) fp32 owned by a flat optimizer (its gradient is accumulated there directly).  Replaces
    the weight permute + cast and the three library GEMMs (forward, input gradient, split-K weight gradient) of a deblock."""

    @staticmethod
    def forward(ctx, x, weight, direct):
        cin, cout, s, _ = weight.shape
        n = x.shape[0]
        dev = x.device
        x = x.contiguous()
        nb = L.load().gdmae_deconv_rows_packed_bytes(cin, cout, s)
        pf = torch.empty(nb, dtype=torch.uint8, device=dev)
        pb = torch.empty(nb, dtype=torch.uint8, device=dev)
        L.call("gdmae_deconv_rows_pack", L.ptr(weight), cin, cout, s, L.ptr(pf), L.ptr(pb), L.stream())
        P = torch.empty(n * s * s, cout, dtype=torch.bfloat16, device=dev)
        L.call("gdmae_deconv_rows_fwd", L.ptr(x), n, cin, cout, s, L.ptr(pf), L.ptr(P), L.stream())
        ctx.save_for_backward(x, pb)
        ctx.meta = (cin, cout, s, direct)
        return P

    @staticmethod
    def backward(ctx, dP):
        x, pb = ctx.saved_tensors
        cin, cout, s, direct = ctx.meta
        n = x.shape[0]
        dP = dP.contiguous()
        if dP.dtype != torch.bfloat16:
            dP = dP.to(torch.bfloat16)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(n, cin, dtype=torch.bfloat16, device=x.device)
            L.call("gdmae_deconv_rows_bwd_input", L.ptr(dP), n, cin, cout, s, L.ptr(pb), L.ptr(dx), L.stream())
        ws = torch.empty(L.load().gdmae_deconv_rows_dw_workspace_bytes(n, cin, cout, s), dtype=torch.uint8, device=x.device)
        L.call("gdmae_deconv_rows_bwd_weight", L.ptr(x), L.ptr(dP), n, cin, cout, s, L.ptr(direct), L.ptr(ws), L.stream())
        return dx, None, None


DECONV_ROWS = os.environ.get("GDMAE_DECONV_ROWS", "1") != "0"     # False: weight permute + library GEMMs (A/B reference)


def deconv_rows(x, deconv):
    """Token rows of a ConvTranspose2d(k = s, stride s, no bias) block: -> (n * s*s, cout), row (token, dy * s + dx)."""
    w = deconv.weight
    s = int(deconv.stride[0])
    cin, cout = w.shape[0], w.shape[1]
    direct = ops.direct_grad(w)
    if (DECONV_ROWS and x.is_cuda and torch.is_autocast_enabled() and x.dtype == torch.bfloat16 and direct is not None and w.dtype == torch.float32
            and w.is_contiguous() and cin in (128, 256) and cout == 128 and s in (1, 2, 4) and x.shape[0] > 0):
        return DeconvRowsFn.apply(x, w, direct)
    wmat = w.permute(0, 2, 3, 1).reshape(cin, s * s * cout)                  # columns ordered (dy, dx, c)
    return ops.linear(x, wmat.t()).view(-1, cout)


class PredHeadFn(torch.autograd.Function):
    """x W^T + b (fp32-accurate forward, bf16-operand backward) for the prediction head nn.Linear(128 -> 48) on the fp32 decoder rows of all pillars (csrc/rows_gemm.hip
    k_pred_*; reference spt_backbone_mae.py:52,74): no cast passes, no unaligned library GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias, direct):
        n, n_out = x.shape[0], weight.shape[0]
        x = x.contiguous()
        packed = torch.empty(L.load().gdmae_pred_head_packed_bytes(), dtype=torch.uint8, device=x.device)
        L.call("gdmae_pred_head_pack", L.ptr(weight), None if bias is None else L.ptr(bias), weight.shape[1], n_out, L.ptr(packed), L.stream())
        xb = torch.empty(n, x.shape[1], dtype=torch.bfloat16, device=x.device)      # the rounded operand rows: operand of the weight gradient
        # fp32-accurate result (three-term split-bf16 product, fp32 bias): the Chamfer kernel squares these offsets, and a bf16-rounded
        # output biases the loss by 1 - 4e-4 relative; an fp32 output also receives the loss's fp32 gradient without a cast in between
        y = torch.empty(n, n_out, dtype=torch.float32, device=x.device)
        L.call("gdmae_pred_head_fwd", L.ptr(x), n, n_out, L.ptr(packed), None, L.ptr(xb), L.ptr(y), L.stream())
        ctx.save_for_backward(xb, packed)
        ctx.meta = (n_out, direct, weight.shape, None if bias is None else bias.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, packed = ctx.saved_tensors
        n_out, direct, wshape, bshape = ctx.meta
        n = x.shape[0]
        dy = dy.contiguous()
        if dy.dtype not in (torch.bfloat16, torch.float32):
            dy = dy.float()
        f32 = dy.dtype == torch.float32                   # rounded inside the input-gradient launch (no cast pass)
        dyb = torch.empty(dy.shape, dtype=torch.bfloat16, device=x.device) if f32 else None
        dx = torch.empty(x.shape, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[0] else None
        if direct is not None:
            dW, db = direct
            ret = (None, None)
        else:
            dW = torch.zeros(wshape, dtype=torch.float32, device=x.device)
            db = None if bshape is None else torch.zeros(bshape, dtype=torch.float32, device=x.device)
            ret = (dW, db)
        ws = torch.empty(L.load().gdmae_pred_head_bwd_workspace_bytes(n), dtype=torch.uint8, device=x.device)
        scale = ops.LAZY_SCALE.pop(dy.data_ptr(), None) if f32 else None         # (upstream gradient, 1 / sum of weights) of the Chamfer mean
        if scale is None and ops.LAZY_SCALE:
            raise RuntimeError("an unscaled Chamfer gradient was handed over (ops.LAZY_SCALE) but did not arrive here as the same tensor")
        L.call("gdmae_pred_head_bwd", L.ptr(dy), int(f32), None if dyb is None else L.ptr(dyb), None if scale is None else L.ptr(scale[0]),
               None if scale is None else L.ptr(scale[1]), L.ptr(x), n, n_out, L.ptr(packed),
               None if dx is None else L.ptr(dx), L.ptr(dW), None if db is None else L.ptr(db), L.ptr(ws), L.stream())
        return (dx, *ret, None)


def pred_head(x, linear):
    """``linear`` (nn.Linear 128 -> n_out) applied to fp32 rows under autocast: -> (n, n_out) fp32 values accurate to fp32 on the fused path
    (bf16 from the library path)."""
    w, b = linear.weight, linear.bias
    if (DECONV_ROWS and x.is_cuda and torch.is_autocast_enabled() and x.dtype == torch.float32 and w.dtype == torch.float32 and w.is_contiguous()
            and w.shape[1] == 128 and 8 <= w.shape[0] <= 64 and w.shape[0] % 8 == 0 and x.shape[0] > 0):
        dw, db = ops.direct_grad(w), (ops.direct_grad(b) if b is not None else None)
        direct = (dw, db) if (dw is not None and (b is None or db is not None)) else None
        y = PredHeadFn.apply(x, w, b, direct)
        y._gd_pred_head = True                 # its backward applies the scalars of a mean on load (ops.ChamferLoss lazy_scale)
        return y
    return ops.linear(x, w, b)


def upsampled_sites(stage_plan, s: int, Y: int, X: int) -> torch.Tensor:
    """(n_tok * s*s,) int32 full-resolution cell of every (token, dy, dx) of a stride-s stage (prepared with the geometry
    plan when the stage's stride matches)."""
    if s == 1:
        return stage_plan.tok_cell
    pre = getattr(stage_plan, "_up_sites", None)
    if pre is not None and pre[0] == s and stage_plan.Y * s == Y and stage_plan.X * s == X:
        return pre[1]
    return gplan.upsample_cells(stage_plan.tok_cell, stage_plan.Y, stage_plan.X, s).reshape(-1).contiguous()


def sparse_decoder(model_cfg, deblocks, conv_out, hidden, pillar_cell, cell2pillar, B, Y, X, want_dense=False, conv_impl='tiles', pre_conv_hook=None):
    """hidden: list of SparseConvTensor per stage.  Returns (features at the pillar sites (M, C) fp32,
    dense spatial_features (B, C, Y, X) or None)."""
    R = B * Y * X
    args, bns = [], []
    cdt = torch.bfloat16 if torch.is_autocast_enabled() else torch.float32
    src_idx = [int(src[-1]) - 1 for src in model_cfg.FEATURES_SOURCE]
    tiles = None
    if cdt == torch.bfloat16 and conv_impl == 'tiles':
        ep = hidden[0]._plan
        dt = gplan.decoder_tiles(ep, src_idx, Y, X)
        tiles = (dt, [hidden[i].stage_plan.map for i in src_idx], [Y // hidden[i].stage_plan.Y for i in src_idx])
    for i, src in enumerate(model_cfg.FEATURES_SOURCE):
        h = hidden[int(src[-1]) - 1]
        sp = h.stage_plan
        deconv, bn = deblocks[i][0], deblocks[i][1]
        s = int(deconv.stride[0])
        assert deconv.kernel_size == (s, s) and sp.Y * s == Y and sp.X * s == X and deconv.bias is None
        P = deconv_rows(h.features, deconv)                                      # (n_tok * s*s, cout)
        args += [upsampled_sites(sp, s, Y, X), P, bn.weight, bn.bias]
        bns.append(bn)
    conv, bn2 = conv_out[0], conv_out[1]
    outs = DecoderHead.apply((B, Y, X, bns[0].eps, bn2.eps, cdt, tiles, pre_conv_hook), conv.weight, bn2.weight, bn2.bias, pillar_cell,
                             cell2pillar, tuple(bns) + (bn2,), *args)
    out, y2, mean2, var2 = outs[:4]
    dense = None
    if want_dense:
        with torch.no_grad():
            a2 = (bn2.weight * torch.rsqrt(var2 + bn2.eps)).float().contiguous()
            b2 = (bn2.bias - a2 * mean2).float().contiguous()
            if y2.shape[0] != R and y2.dtype == torch.bfloat16 and y2.shape[1] % 8 == 0:
                # tile-compact conv output -> the reference's dense map in one pass (expand + BatchNorm affine + ReLU)
                dt = tiles[0]
                C2 = y2.shape[1]
                dmap = torch.empty(R, C2, dtype=torch.float32, device=y2.device)
                L.call("gdmae_tiles_to_dense_affine_relu", L.ptr(y2), L.ptr(dt.tile_slot), L.ptr(DecoderHead.last_ybg), B, Y, X, C2,
                       L.ptr(a2), L.ptr(b2), L.ptr(dmap), L.stream())
                dense = dmap.view(B, Y, X, -1).permute(0, 3, 1, 2)
            else:
                if y2.shape[0] != R:
                    dt = tiles[0]
                    C2 = y2.shape[1]
                    yd = torch.empty(R, C2, dtype=y2.dtype, device=y2.device)
                    L.call("gdmae_tiles_to_dense", L.ptr(y2), L.ptr(dt.tile_slot), L.ptr(DecoderHead.last_ybg), B, Y, X, C2, y2.element_size(),
                           L.ptr(yd), L.stream())
                    y2 = yd
                dense = torch.relu(y2.float() * a2 + b2).view(B, Y, X, -1).permute(0, 3, 1, 2)
    return out, dense
