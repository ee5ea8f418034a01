"""ctypes binding of lib/libvdetr_hip.so (the C-ABI declared in include/vdetr_hip.h).

The binding is deliberately thin: raw device pointers, explicit sizes, the caller's current HIP stream.
It fails loudly — there is no CPU path behind any of these calls.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libvdetr_hip.so")

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

VDETR_ATTN_SHARED_KV, VDETR_ATTN_PER_HEAD = 0, 1
VDETR_MASK_NONE, VDETR_MASK_BOOL, VDETR_MASK_FLOAT = 0, 1, 2


class AttnDesc(ctypes.Structure):
    """Mirror of ``vdetr_attn_desc`` (include/vdetr_hip.h)."""

    _fields_ = [
        ("kind", ctypes.c_int32), ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("nQ", ctypes.c_int32),
        ("nK", ctypes.c_int32), ("scale", c_float),
        ("table", c_void_p), ("table_size", ctypes.c_int32), ("log_scale", c_float), ("inv_log_norm", c_float),
        ("vertices", c_void_p), ("xyz", c_void_p), ("cos_sin", c_void_p),
        ("mask", c_void_p), ("mask_kind", ctypes.c_int32),
        ("dropout_p", c_float), ("seed", ctypes.c_uint64), ("offset", ctypes.c_uint64),
        ("rng_state", c_void_p),
        ("k_row_stride", ctypes.c_int32), ("v_row_stride", ctypes.c_int32),
        ("bwd_aux", c_void_p),
        ("table_grid", ctypes.c_int32), ("kv_waves", ctypes.c_int32), ("fwd_kernel", ctypes.c_int32), ("bwd_kernel", ctypes.c_int32), ("kv_halves", ctypes.c_int32), ("kv_img", c_void_p),
        ("fwd_sched", c_void_p),
    ]


VDETR_CLS_SOFTMAX, VDETR_CLS_SIGMOID = 0, 1

_BOX_IN = ("center", "size", "angle_cls", "angle_res", "cls", "pre_center_norm", "pre_size_norm", "dims_min", "dims_max")
_BOX_OUT = ("center_reg", "size_reg", "center_unnorm", "center_norm", "size_unnorm", "size_norm", "pre_center_unnorm",
            "pre_size_unnorm", "angle_residual", "angle_cont", "angle_prob", "angle_class", "corners", "corners_aa",
            "cls_prob", "objectness")


class BoxDecodeDesc(ctypes.Structure):
    """Mirror of ``vdetr_box_decode_desc`` (include/vdetr_hip.h)."""

    _fields_ = ([(n, ctypes.c_int32) for n in ("B", "N", "A", "C1", "num_angle_bin", "cls_kind")] +
                [(n, c_void_p) for n in _BOX_IN + _BOX_OUT] + [("in_batch_stride", ctypes.c_int32)] +
                [(n, c_void_p) for n in ("cls_logits_t", "angle_logits_t", "angle_res_norm_t", "corners_lidar", "center_size")])


_BOX_GRAD_IN = ("center_reg", "size_reg", "center_unnorm", "center_norm", "size_unnorm", "size_norm", "angle_residual",
                "angle_cont", "angle_prob", "corners", "corners_aa")
_BOX_GRAD_OUT = ("d_center", "d_size", "d_angle_cls", "d_angle_res")


class BoxDecodeGrads(ctypes.Structure):
    """Mirror of ``vdetr_box_decode_grads``."""

    _fields_ = ([(n, c_void_p) for n in _BOX_GRAD_IN + _BOX_GRAD_OUT] +
                [(n, c_void_p) for n in ("cls_logits_t", "angle_logits_t", "angle_res_norm_t", "d_cls")] +
                [("out_batch_stride", ctypes.c_int32), ("slab_rows", ctypes.c_int32)])


class AddLnDesc(ctypes.Structure):
    """Mirror of ``vdetr_addln_desc``."""

    _fields_ = [("rows", ctypes.c_int32), ("C", ctypes.c_int32), ("eps", c_float), ("dropout_p", c_float),
                ("seed", ctypes.c_uint64), ("offset", ctypes.c_uint64), ("rng_state", c_void_p)] + [
        (n, c_void_p) for n in ("x", "r", "gamma", "beta", "gamma2", "beta2", "y", "out", "out2", "mean", "rstd")]


class AddLnGrads(ctypes.Structure):
    """Mirror of ``vdetr_addln_grads``."""

    _fields_ = [(n, c_void_p) for n in ("d_out", "d_out2", "d_y", "d_x", "d_r", "d_gamma", "d_beta", "d_gamma2", "d_beta2",
                                        "partials")]


class AddLnReduce(ctypes.Structure):
    """Mirror of ``vdetr_addln_reduce``."""

    _fields_ = [("partials", c_void_p), ("nparts", ctypes.c_int32), ("C", ctypes.c_int32), ("d_gamma", c_void_p), ("d_beta", c_void_p),
                ("d_gamma2", c_void_p), ("d_beta2", c_void_p)]


class RbLinear(ctypes.Structure):
    """Mirror of ``vdetr_rb_linear``."""

    _fields_ = [("w", c_void_p), ("b", c_void_p), ("wt", c_void_p)]


class RbNorm(ctypes.Structure):
    """Mirror of ``vdetr_rb_norm``."""

    _fields_ = [("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float)]


class RbDrop(ctypes.Structure):
    """Mirror of ``vdetr_rb_drop``."""

    _fields_ = [("p", c_float), ("seed", ctypes.c_uint64)]


class RbQkvDesc(ctypes.Structure):
    """Mirror of ``vdetr_rb_qkv_desc``."""

    _fields_ = [("rows", ctypes.c_int32), ("B", ctypes.c_int32)] + [(n, c_void_p) for n in ("t", "pos", "w", "b", "wt", "x", "out")]


class RbProjQDesc(ctypes.Structure):
    """Mirror of ``vdetr_rb_projq_desc``."""

    _fields_ = ([("rows", ctypes.c_int32), ("B", ctypes.c_int32)] + [(n, c_void_p) for n in ("rng_state", "a", "tgt", "pos")] +
                [("proj", RbLinear), ("q", RbLinear), ("drop1", RbDrop), ("norm2", RbNorm)] +
                [(n, c_void_p) for n in ("y", "mean_y", "rstd_y", "t2", "xq", "qout")])


class RbFfnDesc(ctypes.Structure):
    """Mirror of ``vdetr_rb_ffn_desc``."""

    _fields_ = ([("rows", ctypes.c_int32), ("B", ctypes.c_int32)] + [(n, c_void_p) for n in ("rng_state", "a", "tgt")] +
                [("proj", RbLinear), ("lin1", RbLinear), ("lin2", RbLinear), ("drop2", RbDrop), ("drop_act", RbDrop), ("drop3", RbDrop),
                 ("norm3", RbNorm), ("post1", RbNorm), ("post2", RbNorm)] +
                [(n, c_void_p) for n in ("y", "mean_y", "rstd_y", "t2", "h", "z", "mean_z", "rstd_z", "o1", "o2")])


class RbQkvGrads(ctypes.Structure):
    """Mirror of ``vdetr_rb_qkv_grads``."""

    _fields_ = [(n, c_void_p) for n in ("dq", "dk", "dv", "dq_rows", "dk_rows", "dv_rows", "d_x", "d_x_add", "d_t")]


class RbProjQGrads(ctypes.Structure):
    """Mirror of ``vdetr_rb_projq_grads``."""

    _fields_ = [(n, c_void_p) for n in ("d_y", "d_qout", "d_tgt", "d_a", "d_t2", "dq_rows", "d_proj", "part_n2")]


class RbFfnGrads(ctypes.Structure):
    """Mirror of ``vdetr_rb_ffn_grads``."""

    _fields_ = [(n, c_void_p) for n in ("d_z", "d_o1", "d_o2", "d_tgt", "d_a", "d_lin2", "d_lin1", "d_proj", "part_post", "part_n3")]


class BnActDesc(ctypes.Structure):
    """Mirror of ``vdetr_bnact_desc``."""

    _fields_ = [(n, ctypes.c_int32) for n in ("B", "C", "N", "training", "relu")] + [
        ("eps", c_float), ("momentum", c_float), ("dropout_p", c_float), ("seed", ctypes.c_uint64),
        ("offset", ctypes.c_uint64), ("rng_state", c_void_p)] + [
        (n, c_void_p) for n in ("x", "gamma", "beta", "running_mean", "running_var", "y", "save_mean", "save_invstd",
                                "pre_bias")] + [("counters", c_void_p * 8), ("ncounters", ctypes.c_int32),
                                                ("stats_given", ctypes.c_int32)]


class HeadsDesc(ctypes.Structure):
    """Mirror of ``vdetr_heads_desc``."""

    _fields_ = [(n, ctypes.c_int32) for n in ("B", "N", "G", "rows")] + [
        (n, c_void_p) for n in ("x", "w1t", "w2t", "w3", "b3", "gamma1", "beta1", "gamma2", "beta2", "running_mean1", "running_var1",
                                "running_mean2", "running_var2")] + [
        ("counters1", c_void_p * 8), ("counters2", c_void_p * 8), ("eps", c_float), ("momentum", c_float), ("p1", c_float),
        ("p2", c_float), ("salt1", ctypes.c_uint64), ("salt2", ctypes.c_uint64), ("rng_state", c_void_p)] + [
        (n, c_void_p) for n in ("pre1", "h1", "pre2", "h2", "save_mean1", "save_invstd1", "save_mean2", "save_invstd2", "y",
                                "workspace")]


class AttnParts(ctypes.Structure):
    """Mirror of ``vdetr_attn_parts``."""

    _fields_ = [("part_o", c_void_p), ("part_lse", c_void_p), ("ksplit", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("rows", ctypes.c_int64)]


class AttnKvPrep(ctypes.Structure):
    """Mirror of ``vdetr_attn_kv_prep``."""

    _fields_ = [(n, ctypes.c_int32) for n in ("kind", "B", "H", "nQ", "nK", "v_row_stride")] + [
        (n, c_void_p) for n in ("q", "v", "vertices", "cos_sin", "workspace", "dk", "dv", "bwd_aux")]


class RbAttnEmit(ctypes.Structure):
    """Mirror of ``vdetr_rb_attn_emit``."""

    _fields_ = [(n, c_void_p) for n in ("workspace", "delta", "out", "bwd_aux")] + [("per_head", ctypes.c_int32), ("nQ", ctypes.c_int32)]


class AdamWDesc(ctypes.Structure):
    """Mirror of ``vdetr_adamw_desc``."""

    _fields_ = [(n, c_void_p) for n in ("param", "grad", "exp_avg", "exp_avg_sq")] + [("n", ctypes.c_int64)] + [
        (n, c_void_p) for n in ("step", "ticket", "sumsq")] + [("nsumsq", ctypes.c_int32), ("max_norm", c_float), ("norm_eps", c_float),
                                                                ("norm_out", c_void_p)] + [
        (n, ctypes.c_double) for n in ("lr", "beta1", "beta2", "eps", "weight_decay")]


class PosMlpDesc(ctypes.Structure):
    """Mirror of ``vdetr_posmlp_desc``."""

    _fields_ = [(n, ctypes.c_int32) for n in ("B", "N", "cin")] + [
        (n, c_void_p) for n in ("x", "w1", "b1", "gamma", "beta", "running_mean", "running_var", "counter")] + [
        ("eps", c_float), ("momentum", c_float)] + [
        (n, c_void_p) for n in ("w2t", "b2", "hpre", "hact", "save_mean", "save_invstd", "out")]


class BnActGrads(ctypes.Structure):
    """Mirror of ``vdetr_bnact_grads``."""

    _fields_ = [(n, c_void_p) for n in ("dy", "dx", "d_gamma", "d_beta", "sum_dy_xhat", "sum_dy", "inv_count")]


# ---- set criterion (criterion.hip) -------------------------------------------------------------------------------
VDETR_GT_CORNERS, VDETR_GT_CENTER, VDETR_GT_SIZE, VDETR_GT_ANGLE, VDETR_GT_LABEL = 0, 24, 27, 30, 31
VDETR_GT_ANGLE_CLS, VDETR_GT_ANGLE_RES, VDETR_GT_PRESENT, VDETR_GT_FLOATS = 32, 33, 34, 36
VDETR_LSA_MAX_PROBLEMS = 16

_MATCH_W = ("w_cls", "w_objectness", "w_center", "w_giou", "w_size", "w_angle_cls", "w_angle_reg")
_MATCH_PTR = ("cls", "objectness", "center_reg", "size_reg", "pre_center", "pre_size", "corners", "angle_logits",
              "angle_res_norm", "gt", "nactual", "cost_t", "giou_t", "rotated")


class MatchDesc(ctypes.Structure):
    """Mirror of ``vdetr_match_desc``."""

    _fields_ = ([(n, ctypes.c_int32) for n in ("B", "P", "G", "C", "A", "cls_kind", "label_override")] +
                [(n, c_float) for n in _MATCH_W] + [(n, c_void_p) for n in _MATCH_PTR])


class LsaProblem(ctypes.Structure):
    """Mirror of ``vdetr_lsa_problem``."""

    _fields_ = [(n, c_void_p) for n in ("cost_t", "nactual", "inds", "mask")] + [
        (n, ctypes.c_int32) for n in ("B", "P", "G", "row_repeat")]


class LsaBatch(ctypes.Structure):
    """Mirror of ``vdetr_lsa_batch``."""

    _fields_ = [("nproblems", ctypes.c_int32), ("reserved", ctypes.c_int32), ("p", LsaProblem * VDETR_LSA_MAX_PROBLEMS)]


_LOSS_W = ("w_cls", "w_angle_cls", "w_angle_reg", "w_center", "w_size", "w_giou")
_LOSS_PTR = ("cls_logits", "center_reg", "size_reg", "pre_center", "pre_size", "corners", "angle_logits", "angle_res_norm",
             "gt", "nactual", "inds", "mask", "labels", "num_boxes", "losses", "card_ws", "d_cls_logits", "d_center_reg", "d_size_reg",
             "d_corners", "d_angle_logits", "d_angle_res_norm", "rotated", "ce_rows_matched")


class SetLossDesc(ctypes.Structure):
    """Mirror of ``vdetr_setloss_desc``."""

    _fields_ = ([(n, ctypes.c_int32) for n in ("B", "P", "G", "C", "A", "label_override")] + [("focal_alpha", c_float)] +
                [(n, c_float) for n in _LOSS_W] + [("cls_kind", ctypes.c_int32), ("w_no_object", c_float)] +
                [(n, c_void_p) for n in _LOSS_PTR])


class SpBnDesc(ctypes.Structure):
    """Mirror of ``vdetr_spbn_desc``."""

    _fields_ = ([(n, ctypes.c_int32) for n in ("N", "C", "act", "training")] + [("eps", c_float), ("momentum", c_float)] +
                [(n, c_void_p) for n in ("x", "gamma", "beta", "residual", "running_mean", "running_var", "num_batches_tracked",
                                         "y", "save_mean", "save_invstd", "workspace")])


# name -> (restype, argtypes); must list every symbol of include/vdetr_hip.h (tests check this)
_SIGNATURES = {
    "vdetr_abi_version": (c_int, []),
    "vdetr_ab_switches": (c_int, []),
    "vdetr_last_error": (ctypes.c_char_p, []),
    "vdetr_fps_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vdetr_furthest_point_sampling_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vdetr_fps_varlen_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "vdetr_furthest_point_sampling_varlen_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vdetr_pos_embed_fourier_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "vdetr_pos_embed_sine_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_float, c_float, c_void_p, c_void_p]),
    "vdetr_gather_rows_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "vdetr_gather_rows_grad_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "vdetr_gather_points_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vdetr_gather_points_grad_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vdetr_gather_points_grad_set_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vdetr_group_points_grad_set_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vdetr_ball_query_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "vdetr_group_points_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vdetr_group_points_grad_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vdetr_three_nn_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "vdetr_three_interpolate_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vdetr_three_interpolate_grad_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "vdetr_attn_fwd_workspace_bytes": (c_size_t, [ctypes.POINTER(AttnDesc)]),
    "vdetr_attn_kv_image_bytes": (c_size_t, [c_int, c_int]),
    "vdetr_attn_pack_kv_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_int64, c_void_p, c_void_p]),
    "vdetr_attn_kv_image_parts_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vdetr_attn_pack_kv_parts_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, ctypes.c_int64, c_int, c_void_p, c_void_p]),
    "vdetr_attn_fwd_f32": (c_int, [ctypes.POINTER(AttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vdetr_attn_fwd_bf16": (c_int, [ctypes.POINTER(AttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vdetr_attn_fwd_parts_f32": (c_int, [ctypes.POINTER(AttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                         ctypes.POINTER(AttnParts), c_void_p]),
    "vdetr_attn_bwd_workspace_bytes": (c_size_t, [ctypes.POINTER(AttnDesc)]),
    "vdetr_attn_bwd_scores_f32": (c_int, [ctypes.POINTER(AttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vdetr_attn_bwd_kv_workspace_bytes": (c_size_t, [ctypes.POINTER(AttnDesc)]),
    "vdetr_attn_bwd_kv_f32": (c_int, [ctypes.POINTER(AttnDesc)] + [c_void_p] * 10 + [c_size_t, c_void_p]),
    "vdetr_attn_bwd_kv_delta_f32": (c_int, [ctypes.POINTER(AttnDesc)] + [c_void_p] * 11 + [c_size_t, c_void_p]),
    "vdetr_attn_bwd_dq_f32": (c_int, [ctypes.POINTER(AttnDesc), c_void_p, c_void_p, c_void_p, c_void_p]),
    "vdetr_attn_bwd_table_f32": (c_int, [ctypes.POINTER(AttnDesc), c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "vdetr_attn_bwd_table_kernel_names": (c_int, [ctypes.POINTER(AttnDesc), ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_char_p)]),
    "vdetr_attn_delta_f32": (c_int, [ctypes.POINTER(AttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vdetr_attn_dropout_mask_u8": (c_int, [ctypes.POINTER(AttnDesc), c_void_p, c_void_p]),
    "vdetr_rpe_bias_f32": (c_int, [ctypes.POINTER(AttnDesc), c_void_p, c_void_p]),
    "vdetr_box_decode_fwd_f32": (c_int, [ctypes.POINTER(BoxDecodeDesc), c_void_p]),
    "vdetr_box_decode_bwd_f32": (c_int, [ctypes.POINTER(BoxDecodeDesc), ctypes.POINTER(BoxDecodeGrads), c_void_p]),
    "vdetr_box_decode_bwd_batch_f32": (c_int, [ctypes.POINTER(BoxDecodeDesc), ctypes.POINTER(BoxDecodeGrads), c_int, c_void_p]),
    "vdetr_add_ln_fwd_f32": (c_int, [ctypes.POINTER(AddLnDesc), c_void_p]),
    "vdetr_add_ln_bwd_workspace_bytes": (c_size_t, [ctypes.POINTER(AddLnDesc)]),
    "vdetr_add_ln_bwd_f32": (c_int, [ctypes.POINTER(AddLnDesc), ctypes.POINTER(AddLnGrads), c_void_p]),
    "vdetr_add_ln_param_reduce_batch_f32": (c_int, [ctypes.POINTER(AddLnReduce), c_int, c_void_p]),
    "vdetr_relu_dropout_fwd_f32": (c_int, [c_void_p, c_void_p, ctypes.c_long, c_float, ctypes.c_uint64, ctypes.c_uint64, c_void_p,
                                           c_void_p]),
    "vdetr_relu_dropout_bwd_f32": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.c_long, c_float, c_void_p]),
    "vdetr_bn_act_fwd_f32": (c_int, [ctypes.POINTER(BnActDesc), c_void_p]),
    "vdetr_bn_stats_f32": (c_int, [ctypes.POINTER(BnActDesc), c_void_p, c_void_p, c_void_p]),
    "vdetr_heads_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vdetr_heads_fwd_f32": (c_int, [ctypes.POINTER(HeadsDesc), c_void_p]),
    "vdetr_pos_mlp_fwd_f32": (c_int, [ctypes.POINTER(PosMlpDesc), c_void_p]),
    "vdetr_bn_act_bwd_f32": (c_int, [ctypes.POINTER(BnActDesc), ctypes.POINTER(BnActGrads), c_void_p]),
    "vdetr_bn_act_bwd_batch_f32": (c_int, [ctypes.POINTER(BnActDesc), ctypes.POINTER(BnActGrads), c_int, c_void_p]),
    "vdetr_colsum_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.c_long, c_void_p]),
    "vdetr_colsum_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "vdetr_colsum_batched_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, ctypes.c_long, ctypes.c_long, c_void_p, c_size_t, c_void_p]),
    "vdetr_colsum_ptrs_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, ctypes.c_long, c_void_p, c_size_t, c_void_p]),
    "vdetr_nms3d_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vdetr_nms3d_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, ctypes.c_double, c_int, c_void_p,
                        c_void_p, c_size_t, c_void_p]),
    "vdetr_box_point_count_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vdetr_box3d_iou_max_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p]),
    "vdetr_morton_sort_max": (c_int, []),
    "vdetr_morton_order_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vdetr_pack_chunk_floats": (c_int, []),
    "vdetr_pack_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "vdetr_pack_sumsq_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "vdetr_adamw_clip_f32": (c_int, [ctypes.POINTER(AdamWDesc), c_void_p]),
    "vdetr_cpb_tables_f32": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p] * 3),
    "vdetr_sumsq_blocks": (c_int, [ctypes.c_long]),
    "vdetr_sumsq_f32": (c_int, [c_void_p, ctypes.c_long, c_void_p, c_int, c_void_p]),
    "vdetr_gt_prepare_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vdetr_match_cost_f32": (c_int, [ctypes.POINTER(MatchDesc), c_void_p]),
    "vdetr_match_cost_batch_f32": (c_int, [ctypes.POINTER(MatchDesc), c_int, c_void_p]),
    "vdetr_lsa_f64": (c_int, [ctypes.POINTER(LsaBatch), c_void_p, c_void_p]),
    "vdetr_point_labels_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "vdetr_set_loss_f32": (c_int, [ctypes.POINTER(SetLossDesc), c_void_p]),
    "vdetr_set_loss_batch_f32": (c_int, [ctypes.POINTER(SetLossDesc), c_int, c_void_p]),
    "vdetr_sp_kernel_map_i32": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "vdetr_sp_inverse_map_i32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vdetr_sp_gather_cols_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vdetr_sp_gather_sum_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vdetr_sp_pairs_gemm_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vdetr_sp_pair_plan_workspace_ints": (c_int, [c_int, c_int]),
    "vdetr_sp_pair_plan_i32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                               c_void_p]),
    "vdetr_sp_wgrad_reduce_f32": (c_int, [c_void_p, c_void_p, c_int, ctypes.c_long, c_void_p, c_void_p]),
    "vdetr_sp_pairs_wgrad_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vdetr_sp_bn_workspace_bytes": (c_size_t, [c_int, c_int]),
    "vdetr_sp_bn_act_fwd_f32": (c_int, [ctypes.POINTER(SpBnDesc), c_void_p]),
    "vdetr_sp_bn_act_bwd_f32": (c_int, [ctypes.POINTER(SpBnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vdetr_rb_transpose_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "vdetr_rb_qkv_f32": (c_int, [ctypes.POINTER(RbQkvDesc), c_void_p]),
    "vdetr_rb_proj_q_f32": (c_int, [ctypes.POINTER(RbProjQDesc), c_void_p]),
    "vdetr_rb_qkv_pos_f32": (c_int, [ctypes.POINTER(RbQkvDesc), ctypes.POINTER(PosMlpDesc), c_void_p]),
    "vdetr_rb_ffn_f32": (c_int, [ctypes.POINTER(RbFfnDesc), c_void_p]),
    "vdetr_rb_ffn_parts_f32": (c_int, [ctypes.POINTER(RbFfnDesc), ctypes.POINTER(AttnParts), c_void_p, c_void_p, c_void_p]),
    "vdetr_rb_qkv_bwd_f32": (c_int, [ctypes.POINTER(RbQkvDesc), ctypes.POINTER(RbQkvGrads), c_void_p]),
    "vdetr_rb_proj_q_bwd_f32": (c_int, [ctypes.POINTER(RbProjQDesc), ctypes.POINTER(RbProjQGrads), c_void_p]),
    "vdetr_rb_ffn_bwd_f32": (c_int, [ctypes.POINTER(RbFfnDesc), ctypes.POINTER(RbFfnGrads), c_void_p]),
    "vdetr_rb_ffn0_f32": (c_int, [ctypes.POINTER(RbFfnDesc), c_void_p]),
    "vdetr_rb_ffn0_bwd_f32": (c_int, [ctypes.POINTER(RbFfnDesc), ctypes.POINTER(RbFfnGrads), c_void_p]),
    "vdetr_rb_ffn_bwd_emit_f32": (c_int, [ctypes.POINTER(RbFfnDesc), ctypes.POINTER(RbFfnGrads), ctypes.POINTER(RbAttnEmit), c_void_p]),
    "vdetr_rb_proj_q_bwd_emit_f32": (c_int, [ctypes.POINTER(RbProjQDesc), ctypes.POINTER(RbProjQGrads), ctypes.POINTER(RbAttnEmit), c_void_p]),
    "vdetr_attn_bwd_kv_prep_f32": (c_int, [ctypes.POINTER(AttnKvPrep), c_int, c_void_p]),
    "vdetr_attn_bwd_kv_packed_f32": (c_int, [ctypes.POINTER(AttnDesc)] + [c_void_p] * 10 + [c_size_t, c_void_p]),
    "vdetr_probe_timestamp": (c_int, [c_void_p, c_void_p]),
    "vdetr_topk_order_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "vdetr_anchor_boxes_f32": (c_int, [c_void_p] * 5 + [c_int] * 3 + [c_void_p] * 5),
    "vdetr_gather_proposals_f32": (c_int, [c_void_p, c_int, c_int, c_int, c_int] + [c_void_p] * 14),
    "vdetr_selftest_lds_atomics": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "vdetr_selftest_mfma_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Load (once) the HIP library.  Raises if it has not been built — there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  vdetr_amd has no CPU fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in _SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError = ABI mismatch: fail loudly
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(status, op):
    if status != 0:
        msg = lib().vdetr_last_error().decode(errors="replace")
        raise RuntimeError(f"vdetr_hip {op} failed (status {status}): {msg}")


def stream_ptr():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


# ---- argument checks mirroring the reference's CHECK_* macros (_ext_src/include/utils.h:8-28) ------------
def require_gpu(t, name):
    if not t.is_cuda:
        # reference: AT_ASSERT(false, "CPU not supported") (sampling.cpp:36,62,84 ...)
        raise RuntimeError(f"{name}: CPU not supported (tensor is on {t.device})")


def require_contiguous(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


def require_float(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float tensor")


def require_int(t, name):
    if t.dtype != torch.int32:
        raise RuntimeError(f"{name} must be an int tensor")


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


# ---- rocBLAS' pointer-array batched GEMM (the library GEMM torch does not expose) ------------------------------------------------------
# torch.bmm wants ONE strided tensor per operand: the parked weight-gradient operands of a step (64 separately allocated [1024, 256]
# pairs) went through two torch.stack copies of 64 MB each — 60 us on the backward's tail — before a 71 us GEMM.  rocblas_sgemm_batched
# takes arrays of device pointers: the same Tensile kernel (bit-identical results), 87 us with no copies
# (tools/probes/rocblas_batched_probe.py).  Called on torch's own librocblas; survives stream capture (its workspace is allocated by
# the eager warm-up steps every captured loop runs first).
_rocblas = {"lib": None, "handles": {}}
ROCBLAS_OP_N, ROCBLAS_OP_T = 111, 112


def rocblas():
    if _rocblas["lib"] is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librocblas.so")
        rb = ctypes.CDLL(path if os.path.exists(path) else "librocblas.so")
        rb.rocblas_create_handle.argtypes = [ctypes.POINTER(c_void_p)]
        rb.rocblas_set_stream.argtypes = [c_void_p, c_void_p]
        rb.rocblas_sgemm_batched.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                             c_void_p, c_void_p, c_int, c_int]
        _rocblas["lib"] = rb
    return _rocblas["lib"]


def sgemm_batched_ptrs(op_a, op_b, m, n, k, a_ptrs, lda, b_ptrs, ldb, c_ptrs, ldc, batch, device, alpha=1.0, beta=0.0):
    """column-major C_i [m x n] = alpha op(A_i) op(B_i) + beta C_i for `batch` matrices given by DEVICE arrays of pointers (int64
    tensors or raw addresses), on the current stream of `device`"""
    rb = rocblas()
    key = device.index if device.index is not None else torch.cuda.current_device()
    h = _rocblas["handles"].get(key)
    if h is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("rocBLAS handle: create it before a stream capture (run one step eagerly first)")
        h = c_void_p()
        with torch.cuda.device(key):
            if rb.rocblas_create_handle(ctypes.byref(h)) != 0:
                raise RuntimeError("rocblas_create_handle failed")
        _rocblas["handles"][key] = h
    if rb.rocblas_set_stream(h, c_void_p(torch.cuda.current_stream(device).cuda_stream)) != 0:
        raise RuntimeError("rocblas_set_stream failed")
    al, be = c_float(alpha), c_float(beta)
    addr = lambda p: c_void_p(p.data_ptr() if torch.is_tensor(p) else int(p))
    rc = rb.rocblas_sgemm_batched(h, op_a, op_b, m, n, k, ctypes.byref(al), addr(a_ptrs), lda, addr(b_ptrs), ldb, ctypes.byref(be),
                                  addr(c_ptrs), ldc, batch)
    if rc != 0:
        raise RuntimeError(f"rocblas_sgemm_batched failed (status {rc})")


# pointer tables for such calls: pinned staging -> device.  Eager: a ring of pinned buffers (the host waits only for the upload that
# used a buffer three calls ago); under a stream capture the copy is a captured node that re-reads ITS host buffer at every replay, so
# each captured upload keeps a buffer of its own for good (pre-allocated: no host allocation while capturing).
_ptr_tables = {}


def upload_ptrs(ptrs, device):
    n = len(ptrs)
    key = device.index if device.index is not None else torch.cuda.current_device()
    T = _ptr_tables.get(key)
    if T is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("pointer tables: allocate before a stream capture (run one step eagerly first)")
        T = _ptr_tables[key] = {"ring": [torch.empty(1024, dtype=torch.int64).pin_memory() for _ in range(4)], "events": [None] * 4, "tick": 0,
                                "spare": [torch.empty(1024, dtype=torch.int64).pin_memory() for _ in range(48)], "kept": []}
    if n > 1024:
        raise RuntimeError(f"pointer table of {n} entries (max 1024)")
    dev = torch.empty(n, dtype=torch.int64, device=device)
    if torch.cuda.is_current_stream_capturing():
        if not T["spare"]:
            raise RuntimeError("pointer tables: more than 48 captured uploads in this process")
        host = T["spare"].pop()
        T["kept"].append(host)
        host[:n] = torch.tensor(ptrs, dtype=torch.int64)
        dev.copy_(host[:n], non_blocking=True)
        return dev
    slot = T["tick"] % 4
    T["tick"] += 1
    if T["events"][slot] is not None:
        T["events"][slot].synchronize()
    host = T["ring"][slot]
    host[:n] = torch.tensor(ptrs, dtype=torch.int64)
    dev.copy_(host[:n], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(device))
    T["events"][slot] = ev
    return dev
