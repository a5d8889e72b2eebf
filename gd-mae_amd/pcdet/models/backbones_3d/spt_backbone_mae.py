"""MAE pre-training backbone: masking -> SRA encoder on visible pillars -> generative decoder ->
reconstruction targets + Chamfer loss.  Interface and parameter names follow the reference
``SPTBackboneMAE`` (pcdet/models/backbones_3d/spt_backbone_mae.py:11-153); all index work, the sparse
convolutions' gathers, the attention, the target grouping and the loss run in libgdmae_hip.so.
"""
import torch
import torch.nn as nn

from .spt_backbone import SSTBlockV1, build_decoder, run_decoder, stage_plan_args
from ...utils.spconv_utils import SparseConvTensor
from gdmae_hip import decoder as gdec, ops, plan as gplan


def _only_the_pred_head_sees_the_gradient(pred) -> bool:
    """pred = view of a PredHeadFn output, no tensor hooks and no retained gradient on it or on the head's output: the gradient that
    ops.ChamferLoss hands back reaches gdmae_hip.decoder.PredHeadFn.backward and nothing else."""
    fn = pred.grad_fn
    if fn is None or not type(fn).__name__.startswith('ViewBackward') or pred._backward_hooks or pred.retains_grad:
        return False
    nxt = [f for f, _ in fn.next_functions if f is not None]
    if len(nxt) != 1 or 'PredHeadFn' not in type(nxt[0]).__name__:
        return False
    return not getattr(nxt[0], '_tensor_pre_hooks', None) and not getattr(nxt[0], '_pre_hooks', None) and not getattr(nxt[0], '_retains_grad_hooks', None)


class _DeferredPlan:
    """Handle of SPTBackboneMAE.prefetch_plan_under_decoder.  A plain object with explicit slots - no closures, no class defined per
    call: those were reference CYCLES, so the plan they pointed to (an 832 MB arena, its points, its noise) stayed alive until
    Python's cyclic collector ran and every prefetch in between needed a fresh hipMalloc (21 device-synchronising allocations in 50
    timed steps of round 5's bench)."""
    __slots__ = ("_mod", "_args", "_pf")

    def __init__(self, mod, args):
        self._mod, self._args, self._pf = mod, args, None

    def __call__(self):
        """issue the plan (once)"""
        if self._pf is None and self._args is not None:
            points, batch_size, noise, ready = self._args
            self._args = None
            self._pf = self._mod.prefetch_plan(points, batch_size, noise=noise, ready=ready)

    def ensure_issued(self):
        """after the forward: a forward that did not pass the tile convolution issues the plan here"""
        if self._mod is not None and getattr(self._mod, '_pre_conv_hook', None) is self:
            self._mod._pre_conv_hook = None
        self()

    def finish(self):
        self.ensure_issued()
        pf, self._pf, self._mod = self._pf, None, None
        return pf.finish()


class SPTBackboneMAE(nn.Module):
    # 'sparse': exact sparse-aware decoder (gdmae_hip/decoder.py); 'dense': the reference's dataflow through
    # the torch modules (kept for A/B tests).  dense_spatial_features: materialise batch_dict['spatial_features']
    # (B, C, Y, X) as the reference does; the pre-training step itself only needs the rows at the pillar sites.
    decoder_impl = 'sparse'
    dense_spatial_features = True
    # bf16 mode of the sparse decoder: 'tiles' = the library's bf16-MFMA 3x3 convolution over the active tiles
    # (csrc/conv_tiles.hip), 'dense' = materialised input map + F.conv2d (kept for A/B tests; always used in fp32)
    decoder_conv_impl = 'tiles'

    def __init__(self, model_cfg, input_channels, grid_size, voxel_size, point_cloud_range, **kwargs):
        super().__init__()
        self.model_cfg, self.grid_size, self.voxel_size, self.point_cloud_range = model_cfg, grid_size, voxel_size, point_cloud_range
        self.sparse_shape = grid_size[[1, 0]]
        self.mask_cfg = model_cfg.get('MASK_CONFIG', None)
        self.mask_ratio = self.mask_cfg.RATIO if self.mask_cfg is not None else 0.0
        c = input_channels
        self.sst_blocks = nn.ModuleList()
        for b in model_cfg.SST_BLOCK_LIST:
            self.sst_blocks.append(SSTBlockV1(b, c, b.NAME))
            c = b.ENCODER.D_MODEL
        self.decoder_deblocks, self.decoder_conv_out, c = build_decoder(model_cfg)
        self.decoder_pred = nn.Linear(c, self.mask_cfg.NUM_PRD_POINTS * 3, bias=True)
        self.forward_ret_dict = {}
        self.num_point_features = c

    def get_loss(self, tb_dict=None):
        tb_dict = {} if tb_dict is None else tb_dict
        r = self.forward_ret_dict
        # pred straight from the fused prediction head: the scalars of the mean are applied by the head's backward on load.  That
        # hand-over (an UNSCALED gradient travelling through autograd with its scale on the side) is only safe when nothing but the
        # head's own backward can see the gradient of pred_points: checked here, the scaled gradient otherwise
        pred = r['pred_points']
        lazy = bool(r.get('pred_lazy_scale', False)) and _only_the_pred_head_sees_the_gradient(pred)
        return ops.ChamferLoss.apply(pred, r['gt_points'], r['mask'], lazy), tb_dict

    def forward(self, batch_dict):
        """Inputs: DynVFE's ``voxel_features`` + the voxel plan.  Optional ``mae_noise`` (M,) injects the
        masking noise (one value per pillar, samples concatenated) for bit-exact mask parity tests."""
        vox = batch_dict['_gdmae_vox']
        all_feat, all_coords = batch_dict['voxel_features'], batch_dict['voxel_coords']
        ep = batch_dict.get('_gdmae_plan', None)          # prefetched geometry plan (gdmae_hip.plan.PlanPrefetch)
        if ep is None:
            ep = gplan.encoder_plan(vox, *stage_plan_args(self.model_cfg.SST_BLOCK_LIST),
                                    keep_frac=1 - self.mask_ratio, noise=batch_dict.get('mae_noise', None),
                                    dec_sources=self._dec_sources())
        batch_dict['voxel_mae_mask'] = ep.mask
        batch_dict['_gdmae_plan'] = ep
        x = SparseConvTensor(ops.GatherUnique.apply(all_feat, ep.tok_pillar), ep, 0)
        # data-parallel gradient exchange overlapped with the backward (gdmae_hip.optim.GradSync): when autograd delivers
        # the gradient of a stage's INPUT, that stage (and everything after it) has finished its weight gradients
        sync = batch_dict.get('_gdmae_grad_sync', None)
        hidden = []
        for i, blk in enumerate(self.sst_blocks):
            if sync is not None:
                sync.mark(x.features, [f'backbone_3d.sst_blocks.{i}'])
            x = blk(x)
            hidden.append(x)
        if sync is not None:
            sync.mark(x.features, ['backbone_3d.decoder'])
        feats, strides = {}, {}
        Y0 = int(self.sparse_shape[0])
        for i, h in enumerate(hidden):
            feats[f'x_conv{i + 1}'] = h
            strides[f'x_conv{i + 1}'] = Y0 // h.spatial_shape[0]
        B, X, Y = int(batch_dict['batch_size']), int(self.grid_size[0]), int(self.grid_size[1])
        if self.decoder_impl == 'sparse' and self.training:
            # (eval: the BatchNorm2d layers must use their running statistics -> the module path below)
            # the one-shot hook of prefetch_plan_under_decoder belongs to THIS module and THIS forward: taken off the module before the
            # decoder runs, so a forward that raises leaves nothing behind for an unrelated forward to fire
            hook, self._pre_conv_hook = getattr(self, '_pre_conv_hook', None), None
            pyramid, sf = gdec.sparse_decoder(self.model_cfg, self.decoder_deblocks, self.decoder_conv_out, hidden,
                                              vox.pillar_cell, vox.cell2pillar, B, Y, X,
                                              want_dense=self.dense_spatial_features, conv_impl=self.decoder_conv_impl,
                                              pre_conv_hook=hook)
        else:
            sf = run_decoder(self.model_cfg, self.decoder_deblocks, self.decoder_conv_out, hidden)   # (B, C, Y, X)
            assert sf.shape[0] == B and sf.shape[2] == Y and sf.shape[3] == X
            # features of ALL pillars (masked and visible) gathered from the channels-last map
            pyramid = ops.GatherUnique.apply(sf.permute(0, 2, 3, 1).reshape(B * Y * X, sf.shape[1]), vox.pillar_cell)
        src0 = self.model_cfg.FEATURES_SOURCE[0]
        batch_dict.update({'encoded_spconv_tensor': hidden[-1], 'encoded_spconv_tensor_stride': Y0 // hidden[-1].spatial_shape[0],
                           'multi_scale_3d_features': feats, 'multi_scale_3d_strides': strides, 'spatial_features': sf,
                           'spatial_features_stride': strides[src0] // self.model_cfg.FUSE_LAYER[src0].UPSAMPLE_STRIDE})
        batch_dict.update({'voxel_features': pyramid, 'voxel_coords': all_coords,
                           'voxel_shuffle_inds': torch.arange(all_coords.shape[0], device=all_coords.device)})
        self.forward_ret_dict = self.target_assigner(batch_dict)
        return batch_dict

    def prefetch_plan(self, points, batch_size, noise=None, ready=None):
        """Start building the geometry plan of a batch on a side stream; ``.finish()`` -> (vox, plan) to be placed
        in batch_dict['_gdmae_vox'] / ['_gdmae_plan'] before calling the detector."""
        return gplan.PlanPrefetch(points, self.point_cloud_range, self.voxel_size, self.grid_size, int(batch_size),
                                  *stage_plan_args(self.model_cfg.SST_BLOCK_LIST), keep_frac=1 - self.mask_ratio, noise=noise,
                                  ready=ready, dec_sources=self._dec_sources())

    def prefetch_plan_under_decoder(self, points, batch_size, noise=None, ready=None):
        """``prefetch_plan`` of the NEXT batch, issued from inside this module's next forward right before the decoder's tile
        convolution is launched: the plan's kernels (atomics and scattered rows over the points) then share the device with the one
        matrix-core-bound launch of the step instead of the memory-bound start of a forward (same-box: 7.54 -> 7.44 ms per 8-frame
        step, 4.78 -> 4.63 at 4 frames).  Call it BEFORE the forward of the current batch; ``.finish()`` (after that forward) ->
        (vox, plan) as ``prefetch_plan(...).finish()``.  If the forward never reaches the tile convolution (dense decoder, fp32 mode)
        the plan is issued by ``ensure_issued()`` (call it right after the forward) or, at the latest, by ``finish()``."""
        # the hook lives on the module, not process-global: two models cannot cross-fire, and one left over from an aborted forward
        # is replaced here (its own finish() still issues it)
        self._pre_conv_hook = d = _DeferredPlan(self, (points, batch_size, noise, ready))
        return d

    def _dec_sources(self):
        """Stage indices feeding the decoder (their active sets define the active tiles of conv_out), or None when the
        tile convolution is not in use."""
        if self.decoder_impl != 'sparse' or self.decoder_conv_impl != 'tiles':
            return None
        return [int(src[-1]) - 1 for src in self.model_cfg.FEATURES_SOURCE]

    def target_assigner(self, batch_dict):
        vox = batch_dict['_gdmae_vox']
        gt = ops.group_gt_points(vox, self.mask_cfg.NUM_GT_POINTS)         # (M, K, 3), centre-relative
        flat = gdec.pred_head(batch_dict['voxel_features'], self.decoder_pred)
        return {'pred_points': flat.view(vox.M, -1, 3), 'gt_points': gt, 'mask': batch_dict['voxel_mae_mask'],
                'pred_lazy_scale': bool(getattr(flat, '_gd_pred_head', False))}
