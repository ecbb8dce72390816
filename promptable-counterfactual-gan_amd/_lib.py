"""ctypes binding of libpcgan_hip.so (the C ABI declared in include/pcgan_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this module raises.  `import torch`
happens first on purpose — PyTorch-ROCm ships its own libamdhip64.so.7, and the dynamic linker must
resolve our library's HIP runtime to that already-loaded copy so that streams and device pointers are
shared with PyTorch.
"""
import ctypes
import os

import torch  # noqa: F401  (loads the HIP runtime this process will use)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCG_LIB") or os.path.join(_HERE, "libpcgan_hip.so")  # PCG_LIB: A/B builds of the same ABI

PCG_OK = 0
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4


class PcgError(RuntimeError):
    pass


class ConvGeom(ctypes.Structure):
    """pcg_conv_geom (include/pcgan_hip.h)."""

    _fields_ = [(n, ctypes.c_int32) for n in ("B", "IH", "IW", "Cin", "OH", "OW", "Cout", "KH", "KW", "stride", "pad")]

    def key(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)


_I, _P = ctypes.c_int32, ctypes.c_void_p


class HouseGDesc(ctypes.Structure):
    """pcg_house_g_desc (include/pcgan_hip.h)."""
    _fields_ = ([("fc_in_w", _I), ("fc_in_b", _I)] +
                [(n, _I * 5) for n in ("fc1_w", "fc1_b", "bn1_g", "bn1_b", "fc2_w", "fc2_b", "bn2_g", "bn2_b",
                                       "film_gamma_w", "film_gamma_b", "film_beta_w", "film_beta_b")] +
                [("cont_w", _I), ("cont_b", _I), ("head_w", _I * 8), ("head_b", _I * 8), ("seg", _I * 9)] +
                [(n, _I) for n in ("nheads", "ncont", "D", "NC", "hidden", "nblocks")])


class WgradItem(ctypes.Structure):
    """pcg_wgrad_item."""
    _fields_ = [("dy", _P), ("x", _P), ("dW", _P), ("db", _P)] + [(n, _I) for n in ("ldy", "ldx", "O", "I", "accumulate_w", "accumulate_b",
                                                                                      "tile_x", "tile_y")]


class HouseGFwdArgs(ctypes.Structure):
    """pcg_house_g_fwd_args."""
    _fields_ = ([(n, _P) for n in ("params", "x", "onehot", "mask", "noise", "inp", "H", "Z1", "Z2", "P", "SM")] +
                [("running_mean", _P * 10), ("running_var", _P * 10), ("num_batches_tracked", _P * 10)] +
                [(n, _P) for n in ("cont", "logits", "soft", "hard")] + [("B", ctypes.c_int32)] +
                [(n, ctypes.c_float) for n in ("eps", "momentum", "tau", "res_scale")])


class HouseGBwdArgs(ctypes.Structure):
    """pcg_house_g_bwd_args."""
    _fields_ = ([(n, _P) for n in ("params", "grads", "onehot", "mask", "H", "Z1", "Z2", "SM", "soft", "d_cont", "d_logits", "d_samples",
                                   "DH", "DZ1", "DZ2", "A1", "DN1", "DG", "DB", "DZIN", "DL", "DC", "Q")] +
                [("B", ctypes.c_int32), ("accumulate", ctypes.c_int32), ("tau", ctypes.c_float), ("res_scale", ctypes.c_float)])


class InXform(ctypes.Structure):
    """pcg_in_xform."""
    _fields_ = [("scale", _P), ("shift", _P), ("act", _I), ("slope", ctypes.c_float)]


_c = ctypes
_vp, _f, _i, _i64, _sz = _c.c_void_p, _c.c_float, _c.c_int, _c.c_int64, _c.c_size_t
_d = _c.c_double
_i32 = _c.c_int32
_gp = _c.POINTER(ConvGeom)
_xp = _c.POINTER(InXform)

# name -> (restype, argtypes); every symbol include/pcgan_hip.h declares
PROTOTYPES = {
    "pcg_abi_version": (_i, []),
    "pcg_last_error": (_c.c_char_p, []),
    "pcg_target_arch": (_c.c_char_p, []),
    "pcg_tune_set": (_i, [_c.c_char_p, _i32]),
    "pcg_debug_stamp_buffer": (_i, [_vp, _i64]),
    "pcg_conv_plan_describe": (_i, [_c.POINTER(ConvGeom), _i32, _i32, _c.c_char_p, _sz]),
    "pcg_conv_scratch_parts_bytes": (_sz, []),
    "pcg_conv_scratch_arrivals_bytes": (_sz, []),
    "pcg_conv_set_scratch": (_i, [_vp, _vp, _sz, _vp, _sz]),
    "pcg_conv_reset_scratch": (_i, [_vp]),
    "pcg_conv2d_fwd_bn_g": (_i, [_gp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    "pcg_bn_apply_act_g": (_i, [_vp, _i64, _i32, _vp, _vp, _vp, _vp, _i, _f, _vp, _i32, _vp]),
    "pcg_conv2d_dgrad_bn_phases": (_i32, [_gp]),
    "pcg_conv2d_dgrad_bnbwd_g": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _sz, _i32, _vp]),
    "pcg_bn_bwd_partial_g_workspace_bytes": (_sz, [_i32, _i32]),
    "pcg_bn_bwd_partial_g": (_i, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i, _i32, _vp, _sz, _vp]),
    "pcg_bn_act_bwd_g_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "pcg_bn_act_bwd_premask_g": (_i, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _i32, _vp, _sz, _vp]),
    "pcg_bce_pair": (_i, [_vp, _i64, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_conv2d_fwd_bnbwd_thin_ok": (_i32, [_gp]),
    "pcg_conv2d_fwd_bnbwd_thin_workspace_bytes": (_sz, [_gp]),
    "pcg_conv2d_fwd_bnbwd_thin": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "pcg_conv2d_bnin_full_ok": (_i32, [_gp, _i32]),
    "pcg_conv2d_fwd_bnin_full": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i32, _vp, _vp, _i, _f, _vp, _vp]),
    "pcg_conv2d_wgrad_bnin_full": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _i, _f, _i32, _vp, _vp, _i, _vp, _sz, _vp]),
    "pcg_slab_defer_begin": (_i, [_vp]),
    "pcg_slab_defer_flush": (_i, [_vp]),
    "pcg_slab_defer_pending": (_i32, []),
    "pcg_conv2d_dgrad_bnbwd_full_ok": (_i32, [_gp, _i32]),
    "pcg_conv2d_dgrad_bnbwd_full_workspace_bytes": (_sz, [_gp, _i32]),
    "pcg_conv2d_dgrad_bnbwd_full": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _vp, _i, _i32, _vp, _sz, _vp]),
    "pcg_calib_mfma_blocks": (_i32, [_i32]),
    "pcg_calib_mfma_workspace_bytes": (_sz, [_i32]),
    "pcg_calib_mfma": (_i, [_i32, _i32, _vp, _sz, _c.POINTER(_d), _c.POINTER(_c.c_uint64), _vp]),
    "pcg_calib_copy": (_i, [_vp, _vp, _i64, _vp]),
    "pcg_conv2d_fwd_workspace_bytes": (_sz, [_gp]),
    "pcg_conv2d_dgrad_workspace_bytes": (_sz, [_gp]),
    "pcg_conv2d_fwd": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_dgrad": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_fwd_act": (_i, [_gp, _vp, _vp, _vp, _i, _f, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_dgrad_act": (_i, [_gp, _vp, _vp, _vp, _i, _f, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_fwd_bn_workspace_bytes": (_sz, [_gp]),
    "pcg_conv2d_dgrad_bn_workspace_bytes": (_sz, [_gp]),
    "pcg_conv2d_fwd_bn": (_i, [_gp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_dgrad_bn": (_i, [_gp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_dgrad_mask": (_i, [_gp, _vp, _vp, _vp, _i, _f, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_dgrad_mask_thin_ok": (_i32, [_gp]),
    "pcg_conv2d_xf_thin_ok": (_i32, [_gp]),
    "pcg_conv2d_fwd_add_mask": (_i, [_gp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp]),
    "pcg_conv2d_dgrad_add_mask": (_i, [_gp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp]),
    "pcg_conv2d_fwd_mask": (_i, [_gp, _vp, _vp, _vp, _i, _f, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_dgrad_bnbwd": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_fwd_bnbwd": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _f, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_dgrad_add": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_fwd_bn_partial_rows": (_c.c_int32, [_gp]),
    "pcg_conv2d_dgrad_bn_partial_rows": (_c.c_int32, [_gp]),
    "pcg_bn_bwd_partial_workspace_bytes": (_sz, [_c.c_int32]),
    "pcg_bn_bwd_partial": (_i, [_vp, _vp, _i64, _c.c_int32, _vp, _vp, _vp, _vp, _c.c_int32, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "pcg_conv2d_fwd_bn_xf": (_i, [_gp, _vp, _xp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_dgrad_bn_xf": (_i, [_gp, _vp, _xp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_fwd_xf": (_i, [_gp, _vp, _xp, _vp, _vp, _i, _f, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_dgrad_xf": (_i, [_gp, _vp, _xp, _vp, _vp, _i, _f, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_wgrad_xf": (_i, [_gp, _vp, _xp, _vp, _xp, _vp, _i, _vp, _sz, _vp]),
    "pcg_bn_train_stats_coef": (_i, [_vp, _i64, _c.c_int32, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcg_bn_db_workspace_bytes": (_sz, [_i64, _c.c_int32]),
    "pcg_bn_act_bwd_db": (_i, [_vp, _vp, _vp, _i64, _c.c_int32, _vp, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "pcg_bn_bwd_partial_db_workspace_bytes": (_sz, [_c.c_int32]),
    "pcg_bn_bwd_partial_db": (_i, [_vp, _vp, _i64, _c.c_int32, _vp, _vp, _vp, _vp, _c.c_int32, _f, _vp, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp]),
    "pcg_conv2d_dgrad_add_bnsum": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_fwd_add": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcg_conv2d_fwd_add_bnsum": (_i, [_gp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _sz, _vp]),
    "pcg_conv_weight_adjoint": (_i, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "pcg_conv_weight_adjoint_many": (_i, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "pcg_house_losses": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _vp]),
    "pcg_dp_unique_id": (_i, [_vp]),
    "pcg_dp_init": (_i, [_vp, _i32, _i32]),
    "pcg_dp_world": (_i32, []),
    "pcg_dp_rank": (_i32, []),
    "pcg_dp_side_stream": (_vp, []),
    "pcg_dp_allreduce": (_i, [_vp, _i64, _vp]),
    "pcg_dp_allreduce_begin": (_i, [_vp, _i64, _i32, _vp]),
    "pcg_dp_record": (_i, [_i32]),
    "pcg_dp_allreduce_wait": (_i, [_i32, _vp]),
    "pcg_dp_allreduce_sum_f64": (_i, [_vp, _i64, _vp]),
    "pcg_dp_broadcast": (_i, [_vp, _i64, _i32, _vp]),
    "pcg_dp_sync_batchnorm": (_i, [_i32]),
    "pcg_dp_barrier": (_i, [_vp]),
    "pcg_dp_rccl_version": (_i32, []),
    "pcg_dp_shutdown": (_i, []),
    "pcg_conv2d_wgrad_workspace_bytes": (_sz, [_gp]),
    "pcg_conv2d_wgrad": (_i, [_gp, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "pcg_colsum_workspace_bytes": (_sz, [_i64, _c.c_int32]),
    "pcg_colsum": (_i, [_vp, _i64, _c.c_int32, _vp, _i, _vp, _sz, _vp]),
    "pcg_bn_workspace_bytes": (_sz, [_i64, _c.c_int32]),
    "pcg_bn_train_stats": (_i, [_vp, _i64, _c.c_int32, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pcg_bn_apply_act": (_i, [_vp, _i64, _c.c_int32, _vp, _vp, _f, _vp, _vp, _i, _f, _vp, _f, _vp, _vp]),
    "pcg_bn_act_bwd": (_i, [_vp, _vp, _vp, _i64, _c.c_int32, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "pcg_bn_act_bwd_premask": (_i, [_vp, _vp, _i64, _c.c_int32, _vp, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _i, _vp, _sz, _vp]),
    "pcg_embed_concat_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _vp]),
    "pcg_embed_concat_bwd": (_i, [_vp, _vp, _vp, _vp, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _i, _vp]),
    "pcg_embed_table_grad": (_i, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i, _vp]),
    "pcg_gather_channel": (_i, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "pcg_axpby": (_i, [_vp, _f, _vp, _f, _vp, _i64, _vp]),
    "pcg_scale_mask_fwd": (_i, [_vp, _vp, _f, _vp, _vp, _i64, _vp]),
    "pcg_scale_mask_bwd": (_i, [_vp, _vp, _vp, _f, _vp, _i64, _vp]),
    "pcg_clamp_add_fwd": (_i, [_vp, _vp, _f, _f, _vp, _i64, _vp]),
    "pcg_clamp_add_bwd": (_i, [_vp, _vp, _vp, _f, _f, _vp, _i64, _vp]),
    "pcg_abs_mean_workspace_bytes": (_sz, []),
    "pcg_abs_mean_fwd": (_i, [_vp, _vp, _i, _i64, _vp, _vp, _sz, _vp]),
    "pcg_abs_mean_bwd": (_i, [_vp, _vp, _i, _i64, _vp, _f, _vp, _i, _vp]),
    "pcg_avgpool_fwd": (_i, [_vp, _vp, _c.c_int32, _c.c_int32, _c.c_int32, _vp]),
    "pcg_avgpool_bwd": (_i, [_vp, _vp, _c.c_int32, _c.c_int32, _c.c_int32, _vp]),
    "pcg_cross_entropy_fwd_bwd": (_i, [_vp, _vp, _c.c_int32, _c.c_int32, _f, _vp, _vp, _vp, _vp]),
    "pcg_act_fwd": (_i, [_vp, _i64, _i, _f, _vp, _vp]),
    "pcg_act_bwd": (_i, [_vp, _vp, _i64, _i, _f, _vp, _vp]),
    "pcg_bce_fwd_bwd": (_i, [_vp, _vp, _f, _i64, _f, _vp, _vp, _vp, _vp]),
    "pcg_bce_logits_fwd_bwd": (_i, [_vp, _f, _i64, _f, _vp, _vp, _vp, _vp]),
    "pcg_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _d, _d, _d, _d, _d, _i, _i64, _vp]),
    "pcg_adam_step_capturable": (_i, [_vp, _vp, _vp, _vp, _i64, _d, _d, _d, _d, _d, _i, _vp, _vp, _vp]),
    "pcg_patch_mask": (_i, [_vp, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_int32, _c.c_uint64, _c.c_uint64, _vp]),
    "pcg_randint": (_i, [_vp, _i64, _c.c_int32, _c.c_int32, _vp, _c.c_uint64, _c.c_uint64, _vp]),
    "pcg_randn": (_i, [_vp, _i64, _f, _f, _c.c_uint64, _c.c_uint64, _vp]),
    "pcg_fill": (_i, [_vp, _i64, _f, _vp]),
    "pcg_add_bias_rows": (_i, [_vp, _i64, _i32, _vp, _vp]),
    "pcg_sumsq": (_i, [_vp, _i64, _vp, _i, _vp]),
    "pcg_norm_sum": (_i, [_vp, _vp, _i32, _vp, _vp]),
    "pcg_instnorm_fwd": (_i, [_vp, _i32, _i32, _i32, _vp, _vp, _f, _i, _f, _vp, _vp, _vp, _vp]),
    "pcg_instnorm_bwd": (_i, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_instnorm_bwd_bwd": (_i, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_instnorm_bwd_fused": (_i, [_vp, _vp, _f, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_instnorm_bwd_bwd_act": (_i, [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp]),
    "pcg_rowsum3": (_i, [_i32, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i32, _i32, _vp]),
    "pcg_nhwc_to_nchw_flat": (_i, [_vp, _vp, _i32, _i32, _i32, _i, _vp]),
    "pcg_interpolate": (_i, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "pcg_interpolate_stack": (_i, [_vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "pcg_gradient_penalty_fwd": (_i, [_vp, _i32, _i32, _f, _vp, _vp, _vp]),
    "pcg_gradient_penalty_bwd": (_i, [_vp, _vp, _vp, _i32, _i32, _f, _vp, _vp]),
    "pcg_rand_uniform": (_i, [_vp, _i64, _c.c_uint64, _c.c_uint64, _vp]),
    "pcg_cross_entropy_weighted_fwd_bwd": (_i, [_vp, _vp, _vp, _i32, _i32, _f, _vp, _vp, _vp, _vp]),
    "pcg_dropout_apply": (_i, [_vp, _vp, _i64, _i32, _i32, _f, _vp, _vp]),
    "pcg_rand_bernoulli": (_i, [_vp, _i64, _f, _c.c_uint64, _c.c_uint64, _vp]),
    "pcg_resize8_normalize": (_i, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _f, _f, _vp, _vp]),
    "pcg_house_g_fwd": (_i, [_c.POINTER(HouseGDesc), _c.POINTER(HouseGFwdArgs), _vp]),
    "pcg_house_g_bwd": (_i, [_c.POINTER(HouseGDesc), _c.POINTER(HouseGBwdArgs), _vp]),
    "pcg_weighted_sum_fwd": (_i, [_i32, _vp, _vp, _vp, _vp]),
    "pcg_weighted_sum_bwd": (_i, [_i32, _vp, _vp, _vp, _vp]),
    "pcg_cf_metrics": (_i, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "pcg_gemm": (_i, [_i, _i, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i, _vp]),
    "pcg_linear_wgrad_workspace_bytes": (_sz, [_i32, _i32, _i32]),
    "pcg_linear_wgrad_ticket_count": (_i32, []),
    "pcg_linear_wgrad": (_i, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i, _i, _vp, _sz, _vp, _vp]),
    "pcg_house_residual_fwd": (_i, [_vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _c.POINTER(_i32), _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_house_residual_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _f, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "pcg_house_residual_fwd_sn": (_i, [_vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _c.POINTER(_i32), _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp]),
    "pcg_house_residual_bwd_losses": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _f, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp,
                                           _vp, _vp, _vp, _i32, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _i32, _vp, _vp]),
    "pcg_house_draws": (_i, [_vp, _i32, _i32, _vp, _c.c_uint64, _vp, _i32, _vp, _i32, _c.c_uint64, _vp, _i32, _c.c_uint64, _c.c_uint64, _vp, _vp, _vp]),
    "pcg_house_draws_counter": (_i, [_vp, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _c.c_uint64, _vp, _vp, _vp, _vp]),
    "pcg_house_batch_draws_counter": (_i, [_vp, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _c.c_uint64,
                                           _vp, _vp, _vp, _vp]),
    "pcg_house_diag": (_i, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _f, _vp, _vp, _vp]),
    "pcg_house_residual_bwd_losses_diag": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _f, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp,
                                                _vp, _vp, _vp, _i32, _vp, _vp, _vp, _f, _f, _f, _f, _vp, _i32, _vp,
                                                _vp, _vp, _vp, _vp, _i32, _f, _vp, _vp, _vp]),
    "pcg_house_critic_fwd_n": (_i, [_i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_house_critic_bwd_n": (_i, [_i32, _vp, _i32, _i32, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_spectral_norm_fwd_batched_reps": (_i, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "pcg_spectral_norm_bwd_batched_seq": (_i, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_house_classifier_fwd": (_i, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_house_classifier_bwd": (_i, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_house_classifier_fwd_snbwd": (_i, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                            _vp, _vp, _vp, _f, _vp, _vp, _vp]),
    "pcg_house_classifier_bwd_snfwd": (_i, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f, _vp, _vp, _vp, _vp,
                                            _vp]),
    "pcg_linear_wgrad_grouped_slabs": (_i32, [_i32]),
    "pcg_linear_wgrad_grouped_workspace_bytes": (_sz, [_i32, _c.POINTER(WgradItem), _i32]),
    "pcg_house_critic_fwd": (_i, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_house_critic_bwd": (_i, [_vp, _i32, _i32, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_linear_wgrad_grouped": (_i, [_c.POINTER(WgradItem), _i32, _i32, _vp, _sz, _vp, _vp]),
    "pcg_gemm_act": (_i, [_i, _i, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i, _i, _f, _vp]),
    "pcg_spectral_norm_fwd_batched": (_i, [_i32, _vp, _vp, _vp, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "pcg_spectral_norm_bwd_batched": (_i, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "pcg_onehot": (_i, [_vp, _i32, _i32, _vp, _vp]),
    "pcg_concat_cols": (_i, [_vp, _i32, _vp, _i32, _i32, _vp, _vp]),
    "pcg_split_cols": (_i, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "pcg_film_fwd": (_i, [_vp, _vp, _vp, _vp, _i64, _vp]),
    "pcg_film_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "pcg_gumbel_softmax_fwd": (_i, [_vp, _vp, _vp, _i32, _i32, _i32, _f, _vp, _vp, _vp]),
    "pcg_rand_gumbel": (_i, [_vp, _i64, _c.c_uint64, _c.c_uint64, _vp]),
    "pcg_feature_mask": (_i, [_vp, _i32, _i32, _vp, _i32, _c.c_uint64, _c.c_uint64, _vp]),
    "pcg_gumbel_softmax_bwd": (_i, [_vp, _vp, _vp, _i32, _i32, _i32, _f, _vp, _vp]),
    "pcg_assemble_residual_fwd": (_i, [_vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "pcg_assemble_residual_bwd": (_i, [_vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "pcg_mean_workspace_bytes": (_sz, []),
    "pcg_mean_fwd": (_i, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "pcg_mean_bwd": (_i, [_vp, _f, _i64, _vp, _vp]),
    "pcg_spectral_norm_fwd": (_i, [_vp, _i32, _i32, _vp, _vp, _f, _i, _vp, _vp, _vp, _vp, _vp]),
    "pcg_spectral_norm_bwd": (_i, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _i, _vp]),
}

_lib = None


def load():
    """Load the library (once) and attach prototypes.  Raises PcgError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PcgError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {os.path.join(_HERE, 'csrc')}` — there is no CPU/PyTorch fallback path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_TRACE = os.environ.get("PCG_TRACE_OPS")     # diagnostic: a file; every library call is followed by a device synchronise and logged there


def check(rc, what=""):
    if rc != PCG_OK:
        msg = load().pcg_last_error().decode("utf-8", "replace")
        raise PcgError(f"{what or 'libpcgan_hip'} failed (status {rc}): {msg}")
    if _TRACE:      # (a faulting kernel is then the call AFTER the last line of the file)
        with open(_TRACE, "a") as f:
            f.write(f"{what} issued\n")
        torch.cuda.synchronize()
        with open(_TRACE, "a") as f:
            f.write(f"{what} done\n")
