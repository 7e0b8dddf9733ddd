"""ctypes binding of libhd_ops.so (include/hd_ops.h).  No fallback: if the HIP library is missing
or a call fails, this raises — the product never routes around the kernels."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "csrc", "libhd_ops.so")
ABI_VERSION = 34
MAX_LEVELS = 8
MAX_ANCHORS = 16


class HdError(RuntimeError):
    pass


class RoiLevels(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int),
        ("feat", C.c_void_p * MAX_LEVELS),
        ("h", C.c_int * MAX_LEVELS),
        ("w", C.c_int * MAX_LEVELS),
        ("spatial_scale", C.c_float * MAX_LEVELS),
        ("sb", C.c_int64 * MAX_LEVELS),
        ("sc", C.c_int64 * MAX_LEVELS),
        ("sh", C.c_int64 * MAX_LEVELS),
        ("sw", C.c_int64 * MAX_LEVELS),
        ("k_min", C.c_int),
        ("canonical_scale", C.c_float),
        ("canonical_level", C.c_int),
        ("dtype", C.c_int),
    ]


class Rows2(C.Structure):
    """hd_rows2: optional second place of a [R, C] operand of the fused BN kernels."""
    _fields_ = [("p2", C.c_void_p), ("ld2", C.c_int64), ("split", C.c_int)]


class WgradJob(C.Structure):
    """hd_wgrad_job: row-sliced partial products of a weight gradient waiting for their summation."""
    _fields_ = [("slab", C.c_void_p), ("slices", C.c_int), ("count", C.c_int64), ("dw", C.c_void_p)]


class YoloLevels(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int),
        ("na", C.c_int),
        ("no", C.c_int),
        ("pred", C.c_void_p * MAX_LEVELS),
        ("h", C.c_int * MAX_LEVELS),
        ("w", C.c_int * MAX_LEVELS),
        ("stride", C.c_float * MAX_LEVELS),
        ("anchor_w", (C.c_float * MAX_ANCHORS) * MAX_LEVELS),
        ("anchor_h", (C.c_float * MAX_ANCHORS) * MAX_LEVELS),
        ("sb", C.c_int64 * MAX_LEVELS),
        ("sa", C.c_int64 * MAX_LEVELS),
        ("sy", C.c_int64 * MAX_LEVELS),
        ("sx", C.c_int64 * MAX_LEVELS),
        ("so", C.c_int64 * MAX_LEVELS),
    ]


class YoloLossCfg(C.Structure):
    _fields_ = [
        ("box_gain", C.c_float),
        ("obj_gain", C.c_float),
        ("cls_gain", C.c_float),
        ("balance", C.c_float * MAX_LEVELS),
        ("focal_gamma", C.c_float),
        ("focal_alpha", C.c_float),
        ("focal_terms", C.c_int),
    ]


_P, _I, _F, _L, _Z = C.c_void_p, C.c_int, C.c_float, C.c_int64, C.c_size_t

# name -> (restype, argtypes); exactly the entry points declared in include/hd_ops.h
SIGNATURES = {
    "hd_abi_version": (C.c_int, []),
    "hd_status_str": (C.c_char_p, [_I]),
    "hd_device_count": (C.c_int, []),
    "hd_profile_time_next_launch": (_I, [_P, _P]),
    "hd_anchor_base": (_I, [_P, _I, _P, _I, _P]),
    "hd_anchor_grid": (_I, [_I, _P, _P, _P, _P, _I, _F, _P, _P]),
    "hd_pairwise_iou": (_I, [_P, _L, _P, _P, _I, _I, _I, _P, _P]),
    "hd_iou_assign_workspace_bytes": (_Z, [_I, _I, _I]),
    "hd_iou_assign": (_I, [_P, _L, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    "hd_box_encode": (_I, [_P, _P, _L, _P, _P]),
    "hd_box_decode": (_I, [_P, _P, _L, _F, _F, _P, _P]),
    "hd_nms_workspace_bytes": (_Z, [_I, _I, _I]),
    "hd_nms_batched": (_I, [_P, _P, _P, _P, _I, _I, _F, _F, _I, _I, _P, _P, _P, _Z, _P]),
    "hd_rpn_proposals_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "hd_rpn_proposals": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _I, _F, _F, _F, _F, _P, _P, _P, _P, _P, _Z, _P]),
    "hd_roi_level_map": (_I, [_P, _I, _I, _I, _F, _I, _P, _P]),
    "hd_roi_align_fwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _L, _L, _L, _L, _P]),
    "hd_roi_align_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _I, _L, _L, _L, _L, _P]),
    "hd_roi_align_bwd_gather_supported": (_I, [_P, _I, _I, _I, _I, _I, _L]),
    "hd_roi_align_bwd_gather_workspace_bytes": (_Z, [_I, _I]),
    "hd_roi_align_bwd_gather": (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _L, _L, _L, _L, _P, _Z, _P]),
    "hd_roi_pool_fwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _L, _L, _L, _L, _P]),
    "hd_roi_pool_bwd": (_I, [_P, _P, _I, _I, _I, _P, _P, _L, _L, _L, _L, _P]),
    "hd_yolo_decode": (_I, [_P, _I, _P, _P]),
    "hd_yolo_assign": (_I, [_P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P]),
    "hd_yolo_loss_workspace_bytes": (_Z, [_P, _I]),
    "hd_yolo_loss_fwd": (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "hd_yolo_loss_bwd": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "hd_yolo_loss_bwd_dense": (_I, [_P, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "hd_yolo_loss_bwd_pos": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "hd_yolo_loss_bwd_dense_bf16": (_I, [_P, _P, _I, _P, _P, _P, _P, _Z, _P]),
    "hd_yolo_loss_bwd_pos_bf16": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _P, _Z, _P]),
    "hd_ciou_fwd": (_I, [_P, _P, _L, _P, _P]),
    "hd_ciou_bwd": (_I, [_P, _P, _P, _L, _P, _P]),
    "hd_yolo_nms_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "hd_yolo_nms": (_I, [_P, _I, _I, _I, _F, _F, _I, _I, _P, _P, _P, _Z, _P]),
    "hd_pointwise_conv_supported": (_I, [_L, _I, _I]),
    "hd_pointwise_conv": (_I, [_P, _L, _P, _L, _I, _I, _P, _L, _P, _P]),
    "hd_pointwise_conv_resident": (_I, [_L, _I, _I]),
    "hd_pointwise_conv_dgrad_supported": (_I, [_L, _I, _I]),
    "hd_pointwise_conv_dgrad": (_I, [_P, _L, _P, _L, _I, _I, _P, _L, _P]),
    "hd_pointwise_wgrad_supported": (_I, [_L, _I, _I]),
    "hd_pointwise_wgrad_workspace_bytes": (_Z, [_L, _I, _I]),
    "hd_pointwise_wgrad": (_I, [_P, _L, _P, _L, _L, _I, _I, _P, _P, _Z, _P]),
    "hd_wgrad_reduce": (_I, [_P, _P]),
    "hd_pointwise_conv_dgrad_add": (_I, [_P, _L, _P, _L, _I, _I, _P, _L, _P, _L, _P]),
    "hd_pointwise_wgrad_partial": (_I, [_P, _L, _P, _L, _L, _I, _I, _P, _P, _Z, _P, _P, _P]),
    "hd_pointwise_bwd_supported": (_I, [_L, _I, _I]),
    "hd_pointwise_bwd_workspace_bytes": (_Z, [_L, _I, _I]),
    "hd_pointwise_bwd": (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _P, _L, _P, _L, _P, _P, _Z, _P, _P, _P]),
    "hd_head_conv_pitch": (_I, [_I]),
    "hd_head_conv_fwd_supported": (_I, [_L, _I, _I]),
    "hd_head_conv_fwd": (_I, [_P, _L, _P, _P, _L, _I, _I, _P, _L, _P]),
    "hd_head_conv_bwd_supported": (_I, [_L, _I, _I]),
    "hd_head_conv_bwd_workspace_bytes": (_Z, [_L, _I, _I]),
    "hd_head_conv_bwd": (_I, [_P, _L, _P, _P, _L, _L, _I, _I, _P, _L, _P, _P, _Z, _P, _P, _P]),
    "hd_stem_conv_supported": (_I, [_I, _I, _I, _I, _I]),
    "hd_stem_conv_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "hd_stem_conv_fwd": (_I, [_P, _I, _P, _I, _I, _I, _I, _P, _L, _P, _P]),
    "hd_stem_conv_wgrad_partial": (_I, [_P, _L, _P, _I, _I, _I, _I, _I, _P, _P, _Z, _P, _P]),
    "hd_upsample2x_cat_fwd": (_I, [_P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _P, _L, _P]),
    "hd_upsample2x_cat_bwd": (_I, [_P, _L, _I, _I, _I, _I, _I, _P, _L, _P]),
    "hd_conv3x3_fwd_supported": (_I, [_I, _I, _I, _I, _I]),
    "hd_conv3x3_fwd": (_I, [_P, _L, _P, _I, _I, _I, _I, _I, _P, _L, _P, _P]),
    "hd_conv3x3s2_fwd_supported": (_I, [_I, _I, _I, _I, _I]),
    "hd_conv3x3s2_fwd": (_I, [_P, _L, _P, _I, _I, _I, _I, _I, _P, _L, _P, _P]),
    "hd_conv3x3s2_wgrad_c32_supported": (_I, [_I, _I, _I, _I, _I]),
    "hd_conv3x3s2_wgrad_c32_workspace_bytes": (_Z, [_I, _I, _I, _I, _I]),
    "hd_conv3x3s2_wgrad_c32": (_I, [_P, _L, _P, _L, _I, _I, _I, _I, _I, _P, _P, _Z, _P]),
    "hd_conv3x3s2_dgrad_supported": (_I, [_I, _I, _I, _I, _I]),
    "hd_conv3x3s2_dgrad": (_I, [_P, _L, _P, _I, _I, _I, _I, _I, _P, _L, _P]),
    "hd_conv3x3_wgrad_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "hd_conv3x3_wgrad_select": (None, [_I]),
    "hd_conv3x3_wgrad_workspace_bytes": (_Z, [_I, _I, _I, _I, _I, _I]),
    "hd_conv3x3_wgrad": (_I, [_P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _Z, _P]),
    "hd_conv3x3_wgrad_partial": (_I, [_P, _L, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P, _Z, _P, _P, _P]),
    "hd_conv3x3_flip_weights": (_I, [_I, _P, _P, _P, _P, _P]),
    "hd_conv_flip_weights": (_I, [_I, _P, _P, _P, _P, _P, _P]),
    "hd_adamw_multi": (_I, [_P, _P, _P, _I, _I, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    "hd_maxpool5_fwd": (_I, [_P, _L, _I, _I, _I, _I, _I, _P, _L, _P, _P]),
    "hd_maxpool5_bwd": (_I, [_P, _L, _P, _I, _I, _I, _I, _I, _P, _P]),
    "hd_maxpool3s2_fwd": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "hd_maxpool3s2_bwd": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P]),
    "hd_bn_act_replicas": (C.c_int, []),
    "hd_bn_act_stats": (_I, [_P, _I, _L, _I, _P, _P]),
    "hd_bn_act_apply": (_I, [_P, _I, _L, _L, _I, _P, _P, _P, _F, _F, _P, _P, _P, _P, _I, _P, _I, _P, _L, _P, _P]),
    "hd_bn_act_infer": (_I, [_P, _I, _L, _I, _P, _P, _P, _P, _F, _I, _P, _L, _P, _P]),
    "hd_bias_act_fwd": (_I, [_P, _I, _L, _I, _P, _I, _P, _P]),
    "hd_bias_act_bwd": (_I, [_P, _P, _I, _L, _I, _I, _P, _P, _P]),
    "hd_yolo_detect_workspace_bytes": (_Z, [_P, _I, _I, _I]),
    "hd_yolo_detect": (_I, [_P, _I, _F, _F, _I, _I, _P, _P, _P, _Z, _P]),
    "hd_mosaic4": (_I, [_P, _L, _L, _L, _L, _I, _I, _P, _P, _P, C.c_float, _P, _L, _L, _L, _L, _P]),
    "hd_mosaic4_boxes": (_I, [_P, _P, _P, _I, _I, _I, _P, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P]),
    "hd_bn_act_bwd_reduce": (_I, [_P, _P, _L, _I, _L, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "hd_bn_act_bwd_dx": (_I, [_P, _P, _L, _I, _L, _L, _I, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "hd_graph_replace_memsets": (_I, [_P, _P, _P]),
    "hd_graph_mark": (_I, [_P, _I, _P]),
    "hd_graph_split_instantiate": (_I, [_P, _P, _I, _P, _P, _P, _P]),
    "hd_graph_exec_launch": (_I, [_P, _P]),
    "hd_graph_exec_destroy": (_I, [_P]),
    "hd_wbf_workspace_bytes": (_Z, [_I, _I]),
    "hd_wbf": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _F, _F, _P, _P, _P, _Z, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Loads libhd_ops.so; raises HdError if it was not built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise HdError(f"{SO_PATH} not found: build the HIP extension first "
                      "(python -c 'import __graft_entry__ as g; g.build()'); there is no CPU fallback")
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    v = lib.hd_abi_version()
    if v != ABI_VERSION:
        raise HdError(f"libhd_ops.so ABI {v} != binding ABI {ABI_VERSION}; rebuild")
    trace = os.environ.get("HD_TRACE_CALLS")
    _lib = _Traced(lib, trace) if trace else lib
    return _lib


class _Traced:
    """HD_TRACE_CALLS=<file>: every launching hd_* call appends one line (name + its integer / float arguments) to <file> BEFORE
    it is made.  With HIP_LAUNCH_BLOCKING=1 / AMD_SERIALIZE_KERNEL=3 the last line names the call whose kernel faulted.  Debugging
    aid only; the queries (*_supported, *_bytes, ...) are not traced."""
    _QUIET = ("_supported", "_bytes", "_pitch", "_resident", "_replicas", "_str", "_version", "_tune", "_select")

    def __init__(self, lib, path):
        self._lib = lib
        self._f = open(path, "a", buffering=1)
        self._cache = {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is not None:
            return fn
        raw = getattr(self._lib, name)
        if not name.startswith("hd_") or name.endswith(self._QUIET):
            fn = raw
        else:
            out = self._f

            def fn(*a, _raw=raw, _name=name):
                vals = [getattr(x, "value", x) for x in a]
                out.write(_name + " " + " ".join(str(v) for v in vals if isinstance(v, (int, float))) + "\n")
                return _raw(*a)
        self._cache[name] = fn
        return fn


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().hd_status_str(status).decode()
        raise HdError(f"{what} failed: {msg} ({status})")
