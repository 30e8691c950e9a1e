"""ctypes binding of libyolohip.so (include/yolohip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C yoloseries_amd/csrc``.
There is deliberately no CPU fallback: every product entry point raises if the HIP
library is missing or a kernel reports an error.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("YH_LIBRARY") or os.path.join(_HERE, "libyolohip.so")   # YH_LIBRARY: timing builds (tools/)
CSRC_DIR = os.path.join(_HERE, "csrc")


class YoloHipError(RuntimeError):
    pass


class Seg(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int32), ("C", C.c_int32), ("ups", C.c_int32), ("_pad", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [
        ("seg", Seg * 2), ("nseg", C.c_int32), ("mode", C.c_int32),
        ("B", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("Hi", C.c_int32), ("Wi", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("w", C.c_void_p), ("N", C.c_int32), ("Npad", C.c_int32),
        ("bias", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
        ("act", C.c_int32), ("accumulate", C.c_int32),
        ("out0", C.c_void_p), ("ld0", C.c_int32), ("nsplit", C.c_int32),
        ("out1", C.c_void_p), ("ld1", C.c_int32),
        ("res", C.c_void_p), ("ldr", C.c_int32),
        ("stats", C.c_void_p),
        ("tile_n", C.c_int32), ("grid_cap", C.c_int32), ("tile_k", C.c_int32), ("algo", C.c_int32),
        ("bnr_z", C.c_void_p), ("bnr_ldz", C.c_int32), ("bnr_C", C.c_int32), ("bnr_ws", C.c_void_p), ("bnr_part", C.c_void_p),
    ]


class WgradDesc(C.Structure):
    _fields_ = [
        ("gy", C.c_void_p), ("ldg", C.c_int32), ("N", C.c_int32),
        ("seg", Seg), ("coff_k", C.c_int32), ("Ctot", C.c_int32),
        ("B", C.c_int32), ("Ho", C.c_int32), ("Wo", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("dw", C.c_void_p), ("splits", C.c_int32), ("tile_k", C.c_int32),
        ("partial", C.c_void_p), ("partial_bytes", C.c_uint64),
        ("bn_z", C.c_void_p), ("bn_ldz", C.c_int32), ("reserved0", C.c_int32),
        ("bn_ws", C.c_void_p), ("bn_gamma", C.c_void_p), ("bn_coef", C.c_void_p),
    ]


class V5LossDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("maxbox", C.c_int32), ("num_class", C.c_int32), ("num_anchor", C.c_int32),
        ("num_stage", C.c_int32),
        ("H", C.c_int32 * 4), ("W", C.c_int32 * 4),
        ("img_size0", C.c_float), ("img_size1", C.c_float),
        ("anchors", C.c_float * 24),
        ("anchor_thr", C.c_float),
        ("cls_smooth", C.c_float), ("cls_pos_weight", C.c_float), ("cof_pos_weight", C.c_float),
        ("use_focal", C.c_int32), ("focal_gamma", C.c_float), ("focal_alpha", C.c_float),
        ("iou_scale", C.c_float), ("cof_scale", C.c_float), ("cls_scale", C.c_float),
        ("pred_is_f32", C.c_int32),
        ("ldp", C.c_int32 * 4),
        ("targets_xywhn", C.c_int32),
    ]


class YoloxDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("maxbox", C.c_int32), ("num_class", C.c_int32), ("num_stage", C.c_int32),
        ("H", C.c_int32 * 4), ("W", C.c_int32 * 4), ("ldp", C.c_int32 * 4),
        ("pred_is_f32", C.c_int32), ("img_size0", C.c_float),
        ("use_focal", C.c_int32), ("focal_gamma", C.c_float), ("focal_alpha", C.c_float),
        ("use_l1", C.c_int32),
        ("iou_scale", C.c_float), ("cls_scale", C.c_float), ("cof_scale", C.c_float), ("l1_scale", C.c_float),
        ("cls_smooth", C.c_float), ("cls_pos_weight", C.c_float), ("cof_pos_weight", C.c_float),
        ("iou_type", C.c_int32), ("topk", C.c_int32), ("center_radius", C.c_float), ("cls_cost_const", C.c_float),
    ]


class DecodeDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("num_class", C.c_int32), ("num_anchor", C.c_int32), ("num_stage", C.c_int32),
        ("H", C.c_int32 * 4), ("W", C.c_int32 * 4),
        ("stride", C.c_float * 4),
        ("anchors", C.c_float * 24),
        ("pred_is_f32", C.c_int32),
        ("ldp", C.c_int32 * 4),
        ("yolox", C.c_int32),
    ]


class BnFoldItem(C.Structure):
    _fields_ = [("gamma", C.c_void_p), ("beta", C.c_void_p), ("rm", C.c_void_p), ("rv", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("eps", C.c_float), ("C", C.c_int32)]


class BnPart(C.Structure):
    """yh_bn_part: one BatchNorm of a stacked ConvBnAct layer for yh_bn_silu_apply_parts / yh_bn_silu_bwd_apply_parts"""
    _fields_ = [("ws", C.c_void_p), ("C", C.c_int32), ("ldo", C.c_int32), ("out", C.c_void_p), ("ga", C.c_void_p),
                ("ldga", C.c_int32), ("nblk", C.c_int32), ("gamma", C.c_void_p), ("coef", C.c_void_p),
                ("slab", C.c_void_p), ("ldslab", C.c_int32), ("eps", C.c_float), ("beta", C.c_void_p),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p), ("num_batches", C.c_void_p),
                ("momentum", C.c_float), ("_pad", C.c_int32), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p)]


YH_BN_MAX_PARTS = 4
YH_CMD_SLOTS, YH_CMD_EVENT_RECORD, YH_CMD_STREAM_WAIT = 16, -1, -2


class Cmd(C.Structure):
    """yh_cmd: one command of a program replayed by yh_exec (include/yolohip.h)"""
    _fields_ = [("op", C.c_int32), ("nslots", C.c_int32), ("stream", C.c_int32), ("reserved", C.c_int32),
                ("slots", C.c_uint64 * YH_CMD_SLOTS)]


YH_CONV_FWD, YH_CONV_DGRAD = 0, 1
YH_ACT_NONE, YH_ACT_SILU = 0, 1

_lib = None

_vp, _i32, _i64, _f32, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
_SIGS = {
    "yh_last_error": (C.c_char_p, []),
    "yh_version": (_i32, []),
    "yh_device_cus": (_i32, []),
    "yh_conv_stat_blocks": (_i32, [C.POINTER(ConvDesc)]),
    "yh_conv_igemm": (_i32, [C.POINTER(ConvDesc), _vp]),
    "yh_conv_wgrad": (_i32, [C.POINTER(WgradDesc), _vp]),
    "yh_conv_wgrad_ws_bytes": (C.c_size_t, [C.POINTER(WgradDesc)]),
    "yh_conv_wgrad_patch_ok": (_i32, [C.POINTER(WgradDesc)]),
    "yh_conv_wgrad_patch_name": (_i32, [C.POINTER(WgradDesc), C.c_char_p, _i32]),
    "yh_conv_wgrad_wave_tiles": (_i32, [C.POINTER(WgradDesc)]),
    "yh_conv_wgrad_wave_name": (C.c_char_p, [C.POINTER(WgradDesc)]),
    "yh_conv_wgrad_tiles": (_i32, [_i32, _i32]),
    "yh_conv_wgrad_tiles2": (_i32, [_i32, _i32, _i32]),
    "yh_conv_wgrad_kernel_name": (C.c_char_p, [_i32, _i32]),
    "yh_conv_wgrad_kernel_name2": (C.c_char_p, [_i32, _i32, _i32]),
    "yh_conv_bnr_rows": (_i32, [C.POINTER(ConvDesc)]),
    "yh_conv_kernel_name": (_i32, [C.POINTER(ConvDesc), C.c_char_p, _i32]),
    "yh_bn_finalize": (_i32, [_vp, _i32, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp]),
    "yh_bn_fold": (_i32, [_vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp, _vp]),
    "yh_bn_frozen": (_i32, [_vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp]),
    "yh_bn_fold_batch": (_i32, [_vp, _i32, _vp]),
    "yh_bn_silu_apply": (_i32, [_vp, _i32, _vp, _i32, _i64, _vp, _i32, _vp, _i32, _vp]),
    "yh_ew_blocks": (_i32, [_i64]),
    "yh_bn_silu_bwd_reduce": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i64, _vp, _vp]),
    "yh_bn_bwd_finalize": (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp]),
    "yh_bn_finalize_parts": (_i32, [_vp, _i32, _i64, _vp]),
    "yh_bn_bwd_finalize_parts": (_i32, [_vp, _i32, _i64, _vp]),
    "yh_bn_silu_apply_parts": (_i32, [_vp, _i32, _i64, _vp, _i32, _vp]),
    "yh_bn_silu_bwd_apply_parts": (_i32, [_vp, _i32, _i64, _vp, _i32, _vp, _i32, _vp]),
    "yh_bn_silu_bwd_apply": (_i32, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i64, _vp, _i32, _vp, _i32, _i32, _vp]),
    "yh_colsum": (_i32, [_vp, _i32, _i32, _i64, _vp, _vp, _vp]),
    "yh_maxpool5_fwd": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp]),
    "yh_maxpool5_bwd": (_i32, [_vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "yh_sppf_pool3_ok": (_i32, [_i32, _i32, _i32]),
    "yh_sppf_pool3_fwd": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "yh_sppf_pool3_bwd": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "yh_upsample2_bwd": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp]),
    "yh_input_s2d": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "yh_fill_u32": (_i32, [_vp, C.c_uint32, _i64, _vp]),
    "yh_pack_bf16": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "yh_gather_f32": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "yh_sgd_step": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _i32, _f32, _i32, _i32, _vp, _vp]),
    "yh_sgd_step_dev": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp]),
    "yh_ema_update_dev": (_i32, [_vp, _vp, _i64, _vp, _vp]),
    "yh_ema_advance": (_i32, [_vp, _vp, C.c_double, C.c_double, _vp]),
    "yh_sumsq": (_i32, [_vp, _i64, _vp, _vp, _vp]),
    "yh_clip_scale": (_i32, [_vp, _f32, _vp, _vp]),
    "yh_ema_update": (_i32, [_vp, _vp, _i64, _f32, _vp]),
    "yh_v5loss_ws_bytes": (_sz, [C.POINTER(V5LossDesc)]),
    "yh_v5loss_saved_bytes": (_sz, [C.POINTER(V5LossDesc)]),
    "yh_v5_assign": (_i32, [C.POINTER(V5LossDesc), _vp, _vp, _vp, _vp, _vp, _vp]),
    "yh_v5_loss_fwd": (_i32, [C.POINTER(V5LossDesc), C.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _vp]),
    "yh_v5_loss_bwd": (_i32, [C.POINTER(V5LossDesc), C.POINTER(_vp), _vp, _vp, C.POINTER(_vp), _vp, _vp]),
    "yh_yolox_saved_bytes": (_sz, [C.POINTER(YoloxDesc)]),
    "yh_yolox_ws_bytes": (_sz, [C.POINTER(YoloxDesc)]),
    "yh_yolox_layout": (_i32, [C.POINTER(YoloxDesc), _vp]),
    "yh_yolox_loss_fwd": (_i32, [C.POINTER(YoloxDesc), C.POINTER(_vp), _vp, _vp, _vp, _vp, _vp, _vp]),
    "yh_yolox_loss_bwd": (_i32, [C.POINTER(YoloxDesc), C.POINTER(_vp), _vp, _vp, _vp, C.POINTER(_vp), _vp]),
    "yh_iou_matrix": (_i32, [_vp, _i32, _vp, _i32, _f32, _vp, _vp]),
    "yh_iou_pairwise": (_i32, [_i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "yh_decode_full": (_i32, [C.POINTER(DecodeDesc), C.POINTER(_vp), _vp, _vp]),
    "yh_decode_filter": (_i32, [C.POINTER(DecodeDesc), C.POINTER(_vp), _f32, _f32, _vp, _vp, _i32, _vp, _vp]),
    "yh_decode_filter_ws_bytes": (C.c_size_t, [C.POINTER(DecodeDesc)]),
    "yh_filter_decoded": (_i32, [_vp, _i32, _i32, _i32, _f32, _f32, _i32, _vp, _vp, _i32, _vp]),
    "yh_exec_op": (_i32, [C.c_char_p, C.POINTER(_i32)]),
    "yh_exec": (_i32, [C.POINTER(Cmd), _i32, C.POINTER(_vp), _i32, C.POINTER(_i32)]),
    "yh_nms_ws_bytes": (_sz, [_i32, _i32]),
    "yh_nms_batched": (_i32, [_vp, _vp, _i32, _i32, _f32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGS.keys())


def build(force=False):
    """Compile libyolohip.so for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC_DIR, "clean"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run(["make", "-C", CSRC_DIR, "-j8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise YoloHipError("building libyolohip.so failed:\n" + r.stdout[-4000:])
    return LIB_PATH


def lib():
    """Load the shared library (once) and attach the C signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YoloHipError(
            f"{LIB_PATH} not found: the HIP extension is required (run __graft_entry__.build() or "
            f"`make -C {CSRC_DIR}`); there is no CPU fallback for the product path.")
    # torch bundles its own libamdhip64: import it first so that this library binds to the SAME HIP
    # runtime (two runtimes in one process cannot see each other's device context)
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    missing = []
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            missing.append(name)
            continue
        fn.restype = res
        fn.argtypes = args
    if missing:
        raise YoloHipError(f"{LIB_PATH} lacks symbols declared in include/yolohip.h: {missing}")
    _lib = L
    return L


def check(code, what=""):
    if code != 0:
        msg = lib().yh_last_error()
        raise YoloHipError(f"{what} failed ({code}): {msg.decode() if msg else ''}")


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
