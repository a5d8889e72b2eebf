"""ctypes binding of libgdmae_hip.so (the C ABI declared in include/gdmae_hip.h).

The product path has NO fallback: if the HIP library is missing or an entry point fails, a
``GdmaeHipError`` is raised.  Nothing here imports ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.abspath(os.path.join(_HERE, "..", "csrc", "libgdmae_hip.so"))
if os.environ.get("GDMAE_LIB"):          # kernel experiments: a variant library built next to the product one (tools/build_variant.sh)
    LIB_PATH = os.path.abspath(os.environ["GDMAE_LIB"])


class GdmaeHipError(RuntimeError):
    pass


_P, _I, _L, _F, _D, _Z = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/gdmae_hip.h
SIGNATURES = {
    "gdmae_abi_version": (_I, []),
    "gdmae_target_arch": (C.c_char_p, []),
    "gdmae_last_error": (C.c_char_p, []),
    "gdmae_voxelize_workspace_bytes": (_Z, [_L, _I, _I, _I, _I]),
    "gdmae_voxelize": (_I, [_P, _L, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "gdmae_decorate_points": (_I, [_P, _P, _P, _P, _L, _I, _P, _P, _P, _P]),
    "gdmae_vfe_point_layer_workspace_bytes": (_Z, [_I]),
    "gdmae_pillar_major_rows": (_I, [_P, _I, _P, _P, _P, _L, _P, _P, _P]),
    "gdmae_vfe_point_layer_fwd": (_I, [_P, _P, _P, _P, _I, _L, _I, _P, _P, _P, _I, _P, _P, _D, _D, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "gdmae_vfe_point_layer_bwd": (_I, [_P, _P, _P, _P, _I, _L, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P]),
    "gdmae_vfe_max_layer_workspace_bytes": (_Z, []),
    "gdmae_vfe_max_layer_fwd": (_I, [_P, _L, _P, _P, _P, _I, _P, _P, _D, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gdmae_vfe_max_layer_bwd": (_I, [_P, _L, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "gdmae_vfe_max_layer_fwd_f16": (_I, [_P, _L, _P, _P, _P, _I, _P, _P, _D, _D, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gdmae_vfe_max_layer_bwd_f16": (_I, [_P, _L, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    "gdmae_fill_rows": (_I, [_P, _L, _I, _I, _P, _P]),
    "gdmae_weighted_mean_finish": (_I, [_P, _P, _L, _P, _P]),
    "gdmae_decoder_region_workspace_bytes": (_Z, [_I, _I]),
    "gdmae_decoder_region_algebra": (_I, [_P, _P, _P, _P, _P, _P, _P, _D, _I, _I, _P, _P, _I, _P, _P, _P, _P, _P, _P]),
    "gdmae_segment_max": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "gdmae_segment_max_bwd": (_I, [_P, _P, _P, _L, _I, _P, _P]),
    "gdmae_random_mask": (_I, [_P, _P, _I, _D, _P, _P, _P]),
    "gdmae_visible_tokens": (_I, [_P, _P, _P, _L, _L, _P, _P, _P, _P, _P, _P]),
    "gdmae_downsample_tokens": (_I, [_P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "gdmae_rulebook": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _P, _I, _P, _P]),
    "gdmae_window_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "gdmae_window_partition": (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "gdmae_gather_rows": (_I, [_P, _P, _L, _I, _P, _P]),
    "gdmae_scatter_rows": (_I, [_P, _P, _L, _I, _P, _P]),
    "gdmae_gather_rows_strided": (_I, [_P, _P, _L, _I, _I, _I, _P, _P]),
    "gdmae_scatter_rows_strided": (_I, [_P, _P, _L, _I, _I, _I, _P, _P]),
    "gdmae_colstats_workspace_bytes": (_Z, [_I]),
    "gdmae_colstats": (_I, [_P, _L, _I, _I, _P, _P, _P]),
    "gdmae_rows_affine_relu_scatter": (_I, [_P, _I, _P, _L, _I, _P, _P, _P, _I, _I, _I, _P]),
    "gdmae_rows_affine_relu_sub": (_I, [_P, _I, _P, _L, _I, _P, _P, _P, _P, _I, _I, _I, _P]),
    "gdmae_rows_affine_relu_add": (_I, [_P, _I, _L, _I, _P, _P, _P, _P, _I, _P]),
    "gdmae_rows_bwd_stats_workspace_bytes": (_Z, [_I]),
    "gdmae_rows_bwd_stats": (_I, [_P, _I, _P, _L, _I, _P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "gdmae_bn_fold": (_I, [_P, _L, _I, _I, _D, _P, _P, _D, _D, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gdmae_bn_bwd_coeffs": (_I, [_P, _I, _P, _P, _P, _I, _D, _P, _P, _P, _I, _P, _P]),
    "gdmae_rows_bwd_stats_rows": (_I, [_L]),
    "gdmae_bn_bwd_coeffs_rows": (_I, [_P, _I, _I, _P, _P, _P, _I, _D, _P, _P, _P, _I, _P, _P]),
    "gdmae_border_sums_workspace_bytes": (_Z, [_I, _I]),
    "gdmae_border_sums": (_I, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "gdmae_conv3x3_grad_taps": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P]),
    "gdmae_decoder_tiles_workspace_bytes": (_Z, [_I, _I, _I]),
    "gdmae_decoder_tiles": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "gdmae_conv3x3_tiles_packed_bytes": (_Z, [_I]),
    "gdmae_conv3x3_tiles_pack": (_I, [_P, _I, _I, _P, _I, _P, _P, _P, _P]),
    "gdmae_conv3x3_tiles_workspace_bytes": (_Z, [_I]),
    "gdmae_conv3x3_tiles_fwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _D, _D, _P, _P, _P, _P, _P, _P,
                                     _P, _P]),
    "gdmae_tiles_gather_rows": (_I, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _P]),
    "gdmae_tiles_to_dense": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "gdmae_tiles_to_dense_affine_relu": (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "gdmae_rows_bwd": (_I, [_P, _I, _P, _L, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P]),
    "gdmae_segment_max_affine": (_I, [_P, _I, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "gdmae_segmax_bwd_stats": (_I, [_P, _I, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P]),
    "gdmae_segmax_bn_bwd": (_I, [_P, _I, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _I, _P]),
    "gdmae_window_attention_fwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _F, _P]),
    "gdmae_window_attention_bwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P, _F, _P]),
    "gdmae_window_attention_levels_writes_lse": (_I, [_I, _I, _P, _I, _I]),
    "gdmae_window_attention_levels_fwd": (_I, [_P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _I, _I, _P, _F, _P, _P]),
    "gdmae_window_attention_levels_bwd": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _I, _P, _P, _I, _I, _P, _F, _P, _P, _P]),
    "gdmae_conv3x3_dense_packed_bytes": (_Z, [_I, _I]),
    "gdmae_conv3x3_dense_pack": (_I, [_P, _I, _I, _I, _I, _P, _P]),
    "gdmae_conv3x3_dense": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    "gdmae_conv3x3_dense_add": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    "gdmae_conv3x3_dense_stats_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "gdmae_conv3x3_dense_stat_rows": (_I, []),
    "gdmae_conv3x3_dense_stats": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "gdmae_bn_fold_partials": (_I, [_P, _I, _I, _D, _P, _P, _D, _D, _P, _P, _P, _P, _P, _P, _P]),
    "gdmae_conv3x3_dense_f32out": (_I, [_P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "gdmae_conv3x3_dense_dw_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "gdmae_conv3x3_dense_bwd_weight": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    "gdmae_attention_timing": (_I, [_I]),
    "gdmae_attention_timing_read": (_I, [_I, _P, _P]),
    "gdmae_kernel_timing": (_I, [_I]),
    "gdmae_kernel_timing_slots": (_I, []),
    "gdmae_kernel_timing_name": (C.c_char_p, [_I]),
    "gdmae_kernel_timing_read": (_I, [_I, _P, _P, _P, _P]),
    "gdmae_kernel_timing_read_side": (_I, [_I, _P]),
    "gdmae_add_layernorm_workspace_bytes": (_Z, [_I]),
    "gdmae_add_layernorm_fwd": (_I, [_P, _P, _I, _P, _P, _L, _I, _F, _P, _P, _P, _P]),
    "gdmae_add_layernorm_bwd": (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _L, _I, _P, _P, _P, _P, _P]),
    "gdmae_prep_tokens": (_I, [_P, _P, _P, _L, _I, _P, _P, _I, _P]),
    "gdmae_sum_bf16": (_I, [_P, _I, _L, _P, _P]),
    "gdmae_add3": (_I, [_P, _P, _I, _P, _I, _L, _P, _P]),
    "gdmae_add3_to": (_I, [_P, _P, _I, _P, _I, _L, _P, _I, _P]),
    "gdmae_set_attention_impl": (_I, [_I]),
    "gdmae_sum_partials": (_I, [_P, _L, _F, _P, _I, _P]),
    "gdmae_sum_partials_gated": (_I, [_P, _L, _F, _P, _P, _F, _P]),
    "gdmae_gemm_workspace_bytes": (_Z, []),
    "gdmae_gemm_stats": (_I, [_P]),
    "gdmae_gemm_tuning": (_I, [_I]),
    "gdmae_gemm": (_I, [_P, _P, _P, _L, _L, _L, _I, _I, _I, _I, _P, _P, _P]),
    "gdmae_gemm_tn_splitk_workspace_bytes": (_Z, [_L, _I, _I]),
    "gdmae_gemm_tn_splitk": (_I, [_P, _P, _P, _L, _I, _I, _I, _I, _P, _P]),
    "gdmae_conv_block_scratch_bytes": (_Z, [_L, _L, _I, _I, _I]),
    "gdmae_spconv_packed_bytes": (_Z, [_I, _I]),
    "gdmae_spconv_pack_jobs": (_I, [_P, _I, _I, _I, _P, _P]),
    "gdmae_spconv": (_I, [_P, _I, _P, _P, _L, _I, _I, _P, _I, _P]),
    "gdmae_spconv_stat_rows": (_I, [_I, _I, _I]),
    "gdmae_spconv_stats": (_I, [_P, _I, _P, _P, _L, _I, _I, _P, _P, _P]),
    "gdmae_decoder_dy": (_I, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "gdmae_decoder_site_rulebook": (_I, [_P, _P, _I, _L, _P, _I, _I, _P, _P]),
    "gdmae_tap_dw_rows": (_L, [_L, _I, _I]),
    "gdmae_tap_dw_workspace_bytes": (_Z, [_L, _I, _I]),
    "gdmae_tap_dw": (_I, [_P, _L, _L, _I, _P, _P, _I, _P, _I, _I, _P, _P]),
    "gdmae_conv_block_fwd": (_I, [_P, _P]),
    "gdmae_conv_block_bwd": (_I, [_P, _P]),
    "gdmae_encoder_layer_bytes": (_I, [_L, _I, _I, _I, _I, _P, _P, _P]),
    "gdmae_dw_gemm_workspace_bytes": (_Z, [_L, _I, _I]),
    "gdmae_dw_gemm": (_I, [_P, _P, _L, _I, _I, _P, _P, _P, _P]),
    "gdmae_layer_packed_bytes": (_Z, [_I, _I]),
    "gdmae_layer_pack_jobs": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "gdmae_layer_pack_job_count": (_I, [_I, _I]),
    "gdmae_tok_gemm_pack": (_I, [_P, _I, _P]),
    "gdmae_tok_gemm": (_I, [_P, _P, _P, _L, _L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P]),
    "gdmae_tok_gemm_qkv": (_I, [_P, _P, _P, _P, _P, _L, _I, _P, _P, _P]),
    "gdmae_tok_gemm_ffn": (_I, [_P, _P, _P, _P, _P, _L, _L, _I, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gdmae_tok_gemm_ln_bwd_rows": (_I, [_I]),
    "gdmae_tok_gemm_ln_bwd": (_I, [_P, _P, _L, _L, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gdmae_encoder_layer_fwd": (_I, [_P, _P]),
    "gdmae_encoder_layer_bwd": (_I, [_P, _P]),
    "gdmae_encoder_stage_fwd": (_I, [_P, _I, _P]),
    "gdmae_encoder_stage_bwd": (_I, [_P, _I, _P]),
    "gdmae_deconv_rows_packed_bytes": (_Z, [_I, _I, _I]),
    "gdmae_deconv_rows_pack": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "gdmae_deconv_rows_fwd": (_I, [_P, _L, _I, _I, _I, _P, _P, _P]),
    "gdmae_deconv_rows_bwd_input": (_I, [_P, _L, _I, _I, _I, _P, _P, _P]),
    "gdmae_deconv_rows_dw_workspace_bytes": (_Z, [_L, _I, _I, _I]),
    "gdmae_deconv_rows_bwd_weight": (_I, [_P, _P, _L, _I, _I, _I, _P, _P, _P]),
    "gdmae_pred_head_packed_bytes": (_Z, []),
    "gdmae_pred_head_pack": (_I, [_P, _P, _I, _I, _P, _P]),
    "gdmae_pred_head_fwd": (_I, [_P, _L, _I, _P, _P, _P, _P, _P]),
    "gdmae_pred_head_bwd_workspace_bytes": (_Z, [_L]),
    "gdmae_pred_head_bwd": (_I, [_P, _I, _P, _P, _P, _P, _L, _I, _P, _P, _P, _P, _P, _P]),
    "gdmae_encoder_set_layer_path": (_I, [_I]),
    "gdmae_encoder_stage_fused": (_I, [_P, _I]),
    "gdmae_group_gt_points": (_I, [_P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P]),
    "gdmae_chamfer": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P]),
    "gdmae_augment_collate_workspace_bytes": (_Z, [_L]),
    "gdmae_augment_collate": (_I, [_P, _L, _I, _P, _I, _P, _P, _P, _P, _P, _P]),
    "gdmae_group_workspace_bytes": (_Z, [_L, _L]),
    "gdmae_ingroup_inds": (_I, [_P, _L, _L, _P, _P, _Z, _P]),
    "gdmae_group_inner_inds": (_I, [_P, _L, _L, _I, _P, _P, _Z, _P]),
    "gdmae_center_head_targets_workspace_bytes": (_Z, [_I, _I]),
    "gdmae_center_head_targets": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _P, _F, _I, _I, _I, _D, _I, _P, _P, _P, _P, _P, _P]),
    "gdmae_center_head_targets_iou": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _P, _F, _I, _I, _I, _D, _I, _P, _P, _P, _P, _P, _P, _P]),
    "gdmae_focal_loss_rows": (_I, []),
    "gdmae_focal_loss_fwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "gdmae_focal_loss_bwd": (_I, [_P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "gdmae_center_head_decode": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _F, _P, _F, _I, _P, _P, _P, _P, _P]),
    "gdmae_boxes_bev_pairs": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "gdmae_nms_workspace_bytes": (_Z, [_I]),
    "gdmae_nms_bev": (_I, [_P, _I, _F, _I, _P, _P, _P, _P]),
    "gdmae_geometry_plan_layout": (_I, [_P, _P, _I, _P, _P]),
    "gdmae_geometry_plan": (_I, [_P, _P, _P, _P, _Z, _P]),
    "gdmae_grad_sq_norm": (_I, [_P, _L, _P, _P, _P]),
    "gdmae_adam_step": (_I, [_P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _I, _F, _F, _P, _P]),
    "gdmae_adam_step_shadow": (_I, [_P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _I, _F, _F, _P, _P, _P]),
}

class LayerArgs(C.Structure):
    """ctypes mirror of ``gdmae_layer_args`` (include/gdmae_hip.h)."""
    _fields_ = ([("n", _L), ("d", _I), ("ff", _I), ("nhead", _I), ("bf16", _I), ("eps", _F), ("tau_min", _F), ("n_levels", _I),
                 ("n_win", _I * 4), ("max_tokens", _I * 4)]
                + [(k, _P) for k in ("tok_pos", "csr_tok", "win_start", "win_len", "pos_table", "Win", "bin", "Wo", "bo", "W1", "b1",
                                     "W2", "b2", "g1", "be1", "g2", "be2", "tau", "x", "y", "dy", "dx", "dWin", "dbin", "dtau", "dWo",
                                     "dbo", "dW1", "db1", "dW2", "db2", "dg1", "dbe1", "dg2", "dbe2", "saved", "scratch", "packed")]
                + [("x_bf16", _I), ("res_out", _P), ("dres", _P), ("dx_bf16", _P)])


class PlanParams(C.Structure):
    """ctypes mirror of ``gdmae_plan_params`` (include/gdmae_hip.h)."""
    _fields_ = [("n_points", _L), ("n_cols", _I), ("batch_size", _I), ("lo", _F * 3), ("vs", _F * 3), ("grid", _I * 3), ("n_stages", _I),
                ("stride", _I * 4), ("win_x", _I * 4), ("win_y", _I * 4), ("n_levels", _I * 4), ("drop_lo", (_I * 3) * 4),
                ("drop_hi", (_I * 3) * 4), ("max_tokens", (_I * 3) * 4), ("masked", _I), ("keep_frac", _D), ("n_dec", _I),
                ("dec_sources", _I * 4), ("want_pm", _I), ("cap_points", _L)]


class PlanBuffer(C.Structure):
    """ctypes mirror of ``gdmae_plan_buffer``."""
    _fields_ = [("name", C.c_char * 40), ("offset", _L), ("bytes", _L)]


class ConvBlockArgs(C.Structure):
    """ctypes mirror of ``gdmae_conv_block_args`` (include/gdmae_hip.h)."""
    _fields_ = ([("n_in", _L), ("n_out", _L), ("cin", _I), ("cout", _I), ("bf16", _I), ("x_f32", _I), ("g_f32", _I),
                 ("eps", _F), ("momentum", _F)]
                + [(k, _P) for k in ("x", "nbr", "nbr_t", "W", "gamma", "beta", "running_mean", "running_var", "num_batches", "cols",
                                     "y", "stats", "ab", "mv", "out", "g", "dx", "dW", "dgamma", "dbeta", "scratch", "packed_fwd", "packed_bwd")]
                + [("out_f32", _I)])


_lib = None


def load():
    """Load the shared library and bind every declared symbol (works without a GPU)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GdmaeHipError(f"{LIB_PATH} not found - build it with `python -m gdmae_hip.build` "
                            "(__graft_entry__.build()); there is no CPU fallback for the product path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "HIP entry points need contiguous device tensors"
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """Raw hipStream_t of torch's current stream on the current device (~0.3 us through the C binding; building a
    torch.cuda.Stream object first cost 11 us x 35 calls per step)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise GdmaeHipError(f"{name} failed ({rc}): {lib.gdmae_last_error().decode()}")


def host_f32(vals):
    return (C.c_float * len(vals))(*[float(v) for v in vals])


def host_i32(vals):
    return (C.c_int * len(vals))(*[int(v) for v in vals])


def host_i64(vals):
    return (C.c_longlong * len(vals))(*[int(v) for v in vals])


def ptr_any(t):
    """data pointer of a device tensor whose elements are dense in memory in ANY dimension order (e.g. a channels-last map)."""
    assert t.is_cuda
    return t.data_ptr()


def host_ptrs(tensors):
    """Host array of device pointers (the `const T* const*` arguments of the C ABI)."""
    return (C.c_void_p * len(tensors))(*[ptr(t) for t in tensors])


def host_ptrs_any(tensors):
    """... of tensors that are dense in memory in any dimension order (ptr_any)."""
    return (C.c_void_p * len(tensors))(*[ptr_any(t) for t in tensors])
