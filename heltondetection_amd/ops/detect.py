"""Detection ops: anchors, pairwise IoU + assignment, delta codec, NMS, RPN proposals, RoIAlign / RoIPool, YOLOv5 decode /
build-targets / loss / detect, CIoU, WBF, Mosaic-4 (csrc/anchors.hip, iou.hip, codec.hip, nms.hip, rpn.hip, roi_align.hip,
yolo.hip, wbf.hip, mosaic.hip)."""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from .. import _lib
from .._lib import RoiLevels, Rows2, YoloLevels, YoloLossCfg, check
from ._core import *  # noqa: F401,F403
from ._core import _F32, _I32, _TIMED, _ktimed, _need, _note_bytes, _ptr, _stream, _timed, _ws  # noqa: F401
from .backbone import _cl_row_stride, head_conv_bwd_rows, head_conv_pitch, head_conv_supported  # the loss runs the Detect backward

# ------------------------------------------------------------------------------- anchors
def anchor_base(sizes: Sequence[float], ratios: Sequence[float]) -> torch.Tensor:
    """Host helper: [len(ratios)*len(sizes), 4] base anchors (CPU tensor)."""
    lib = _lib.load()
    s = (C.c_float * len(sizes))(*sizes)
    r = (C.c_float * len(ratios))(*ratios)
    out = (C.c_float * (4 * len(sizes) * len(ratios)))()
    check(lib.hd_anchor_base(s, len(sizes), r, len(ratios), out), "hd_anchor_base")
    return torch.tensor(list(out), dtype=_F32).view(-1, 4)


def anchor_grid(feat_hw: Sequence[Tuple[int, int]], strides: Sequence[float], base_anchors: torch.Tensor,
                offset: float = 0.5, device="cuda") -> torch.Tensor:
    """base_anchors: CPU [L, A, 4]. Returns device [sum H*W*A, 4] (level, y, x, anchor order)."""
    lib = _lib.load()
    L, A = base_anchors.shape[0], base_anchors.shape[1]
    fh = (C.c_int * L)(*[int(h) for h, _ in feat_hw])
    fw = (C.c_int * L)(*[int(w) for _, w in feat_hw])
    st = (C.c_float * L)(*[float(s) for s in strides])
    base = base_anchors.to(_F32).contiguous().cpu()
    n = sum(int(h) * int(w) for h, w in feat_hw) * A
    out = torch.empty((n, 4), dtype=_F32, device=device)
    check(lib.hd_anchor_grid(L, fh, fw, st, C.c_void_p(base.data_ptr()), A, float(offset), _ptr(out), _stream(out)),
          "hd_anchor_grid")
    return out


# ------------------------------------------------------------------------------- IoU
def pairwise_iou(a: torch.Tensor, b: torch.Tensor, m_valid: Optional[torch.Tensor] = None) -> torch.Tensor:
    """a [B,N,4] or [N,4] (shared across the batch), b [B,M,4] -> [B,N,M]; padded columns = -1."""
    lib = _lib.load()
    b = _need(b, _F32, "b")
    a = _need(a, _F32, "a")
    B, M = b.shape[0], b.shape[1]
    shared = a.dim() == 2
    N = a.shape[-2]
    if m_valid is not None:
        m_valid = _need(m_valid, _I32, "m_valid")
    out = torch.empty((B, N, M), dtype=_F32, device=b.device)
    with _ktimed("pairwise_iou"):
        check(lib.hd_pairwise_iou(_ptr(a), 0 if shared else N * 4, _ptr(b), _ptr(m_valid), B, N, M, _ptr(out),
                                  _stream(b)), "hd_pairwise_iou")
    return out


def iou_assign(a: torch.Tensor, b: torch.Tensor, m_valid: Optional[torch.Tensor] = None, with_gt: bool = True):
    """Fused max/argmax. Returns (max_iou [B,N], argmax [B,N], gt_max_iou [B,M], gt_argmax [B,M])."""
    lib = _lib.load()
    b = _need(b, _F32, "b")
    a = _need(a, _F32, "a")
    B, M = b.shape[0], b.shape[1]
    shared = a.dim() == 2
    N = a.shape[-2]
    if m_valid is not None:
        m_valid = _need(m_valid, _I32, "m_valid")
    dev = b.device
    mx = torch.empty((B, N), dtype=_F32, device=dev)
    am = torch.empty((B, N), dtype=_I32, device=dev)
    gmx = torch.empty((B, M), dtype=_F32, device=dev) if with_gt else None
    gam = torch.empty((B, M), dtype=_I32, device=dev) if with_gt else None
    ws, wsb = _ws(lib.hd_iou_assign_workspace_bytes(B, N, M) if with_gt else 0, dev)
    check(lib.hd_iou_assign(_ptr(a), 0 if shared else N * 4, _ptr(b), _ptr(m_valid), B, N, M, _ptr(mx), _ptr(am),
                            _ptr(gmx), _ptr(gam), _ptr(ws), wsb, _stream(b)), "hd_iou_assign")
    return mx, am, gmx, gam


# ------------------------------------------------------------------------------- codec
def box_encode(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    src, dst = _need(src, _F32, "src"), _need(dst, _F32, "dst")
    out = torch.empty_like(src)
    check(lib.hd_box_encode(_ptr(src), _ptr(dst), src.numel() // 4, _ptr(out), _stream(src)), "hd_box_encode")
    return out


def box_decode(anchors: torch.Tensor, deltas: torch.Tensor, img_h: float = 0.0, img_w: float = 0.0) -> torch.Tensor:
    lib = _lib.load()
    anchors, deltas = _need(anchors, _F32, "anchors"), _need(deltas, _F32, "deltas")
    out = torch.empty_like(deltas)
    check(lib.hd_box_decode(_ptr(anchors), _ptr(deltas), deltas.numel() // 4, float(img_h), float(img_w), _ptr(out),
                            _stream(deltas)), "hd_box_decode")
    return out


# ------------------------------------------------------------------------------- NMS
def nms_batched(boxes: torch.Tensor, scores: torch.Tensor, cls: Optional[torch.Tensor] = None,
                n_valid: Optional[torch.Tensor] = None, score_thr: float = float("-inf"), iou_thr: float = 0.5,
                max_candidates: int = 0, max_out: Optional[int] = None):
    """Exact greedy NMS per image. Returns (keep [B,max_out] int32 padded -1, count [B] int32)."""
    lib = _lib.load()
    boxes, scores = _need(boxes, _F32, "boxes"), _need(scores, _F32, "scores")
    B, N = scores.shape
    max_out = N if max_out is None else int(max_out)
    if cls is not None:
        cls = _need(cls, _I32, "cls")
    if n_valid is not None:
        n_valid = _need(n_valid, _I32, "n_valid")
    dev = boxes.device
    keep = torch.empty((B, max_out), dtype=_I32, device=dev)
    cnt = torch.empty((B,), dtype=_I32, device=dev)
    ws, wsb = _ws(lib.hd_nms_workspace_bytes(B, N, max_candidates), dev)
    check(lib.hd_nms_batched(_ptr(boxes), _ptr(scores), _ptr(cls), _ptr(n_valid), B, N, float(score_thr),
                             float(iou_thr), int(max_candidates), max_out, _ptr(keep), _ptr(cnt), _ptr(ws), wsb,
                             _stream(boxes)), "hd_nms_batched")
    return keep, cnt


def rpn_proposals(logits: torch.Tensor, deltas: torch.Tensor, anchors: torch.Tensor, level_sizes: Sequence[int],
                  pre_nms_topk: int, post_nms_topk: int, iou_thr: float, min_size: float, img_h: float, img_w: float,
                  return_index: bool = False):
    """Returns (rois [B,post,4], scores [B,post], count [B]) and, with return_index, the anchor index of every
    kept proposal [B,post] int32 (-1 padded) — the NMS keep set itself."""
    lib = _lib.load()
    logits, deltas, anchors = _need(logits, _F32, "logits"), _need(deltas, _F32, "deltas"), _need(anchors, _F32, "anchors")
    B, NA = logits.shape
    L = len(level_sizes)
    ls = (C.c_int * L)(*[int(v) for v in level_sizes])
    dev = logits.device
    rois = torch.empty((B, post_nms_topk, 4), dtype=_F32, device=dev)
    sc = torch.empty((B, post_nms_topk), dtype=_F32, device=dev)
    cnt = torch.empty((B,), dtype=_I32, device=dev)
    idx = torch.empty((B, post_nms_topk), dtype=_I32, device=dev) if return_index else None
    ws, wsb = _ws(lib.hd_rpn_proposals_workspace_bytes(B, NA, L, pre_nms_topk, post_nms_topk), dev)
    with _timed("rpn_proposals"):
        check(lib.hd_rpn_proposals(_ptr(logits), _ptr(deltas), _ptr(anchors), B, NA, L, ls, int(pre_nms_topk),
                                   int(post_nms_topk), float(iou_thr), float(min_size), float(img_h), float(img_w),
                                   _ptr(rois), _ptr(sc), _ptr(idx), _ptr(cnt), _ptr(ws), wsb, _stream(logits)),
              "hd_rpn_proposals")
    return (rois, sc, cnt, idx) if return_index else (rois, sc, cnt)


# ------------------------------------------------------------------------------- RoIAlign
def _roi_levels(feats: Sequence[torch.Tensor], scales: Sequence[float], k_min: int, canonical_scale: float,
                canonical_level: int, allow_bf16: bool = False) -> RoiLevels:
    st = RoiLevels()
    st.n_levels = len(feats)
    dt = feats[0].dtype
    ok = (_F32, torch.bfloat16) if allow_bf16 else (_F32,)
    for l, f in enumerate(feats):
        if not f.is_cuda or f.dtype not in ok or f.dtype != dt or f.dim() != 4:
            raise _lib.HdError("roi levels: features must be [B,C,H,W] tensors of one dtype (float32%s) on a HIP device"
                               % (" or bfloat16" if allow_bf16 else ""))
        st.feat[l] = f.data_ptr()
        st.h[l], st.w[l] = f.shape[2], f.shape[3]
        st.spatial_scale[l] = float(scales[l])
        st.sb[l], st.sc[l], st.sh[l], st.sw[l] = f.stride(0), f.stride(1), f.stride(2), f.stride(3)
    st.k_min = int(k_min)
    st.canonical_scale = float(canonical_scale)
    st.canonical_level = int(canonical_level)
    st.dtype = 0 if dt == _F32 else 1
    return st


def roi_level_map(rois: torch.Tensor, k_min: int, n_levels: int, canonical_scale: float = 224.0,
                  canonical_level: int = 4) -> torch.Tensor:
    lib = _lib.load()
    rois = _need(rois, _F32, "rois")
    out = torch.empty((rois.shape[0],), dtype=_I32, device=rois.device)
    check(lib.hd_roi_level_map(_ptr(rois), rois.shape[0], k_min, n_levels, float(canonical_scale), canonical_level,
                               _ptr(out), _stream(rois)), "hd_roi_level_map")
    return out


class RoiBwdSwitch:
    """Which RoIAlign backward runs: the gather (csrc/roi_align.hip roi_align_bwd_gather_kernel: no atomics, every cell written
    once in the maps' dtype, bit-reproducible) or the float-atomic scatter.  HD_ROI_BWD_GATHER=0 | 1 overrides the default."""
    gather = os.environ.get("HD_ROI_BWD_GATHER", "1") != "0"


class _RoIAlignFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, P, scales, sampling_ratio, aligned, k_min, canonical_scale, canonical_level,
                channels_last_out, out_dtype, *feats):
        lib = _lib.load()
        rois = _need(rois.detach(), _F32, "rois")
        feats_d = [f.detach() for f in feats]
        st = _roi_levels(feats_d, scales, k_min, canonical_scale, canonical_level, allow_bf16=True)
        R, Cc = rois.shape[0], feats[0].shape[1]
        out = torch.empty((R, Cc, P, P), dtype=out_dtype, device=rois.device,
                          memory_format=torch.channels_last if channels_last_out else torch.contiguous_format)
        with _ktimed("roi_align_fwd"):
            check(lib.hd_roi_align_fwd(C.byref(st), _ptr(rois), R, Cc, P, int(sampling_ratio), int(bool(aligned)),
                                       _ptr(out), 0 if out_dtype == _F32 else 1, out.stride(0), out.stride(1),
                                       out.stride(2), out.stride(3), _stream(rois)), "hd_roi_align_fwd")
        ctx.save_for_backward(rois)
        ctx.meta = (P, tuple(scales), sampling_ratio, aligned, k_min, canonical_scale, canonical_level,
                    [(tuple(f.shape), f.is_contiguous(memory_format=torch.channels_last), f.dtype) for f in feats])
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        (rois,) = ctx.saved_tensors
        P, scales, sampling_ratio, aligned, k_min, cs, cl, shapes = ctx.meta
        if grad_out.dtype not in (_F32, torch.bfloat16):
            grad_out = grad_out.float()
        R, Cc = rois.shape[0], shapes[0][0][1]
        # gather form: every cell of the gradient maps written once in the features' own dtype, no atomics, no zero-fill, no
        # cast pass, bit-reproducible (csrc/roi_align.hip: roi_align_bwd_gather_kernel)
        # (832^2 bs16 pyramid, C = 256, R = 2048: 0.37 ms against 0.99 ms for zero-fill + atomics + cast with bf16 maps, 0.66
        # against 0.75 ms with float32 maps: tools/roi_bwd_probe.py)
        if RoiBwdSwitch.gather and all(cl_fmt for _, cl_fmt, _ in shapes) and len({dt for _, _, dt in shapes}) == 1:
            go = grad_out if grad_out.stride(1) == 1 else grad_out.contiguous(memory_format=torch.channels_last)
            maps = [torch.empty(shape, dtype=dt, device=rois.device, memory_format=torch.channels_last) for shape, _, dt in shapes]
            stg = _roi_levels(maps, scales, k_min, cs, cl, allow_bf16=True)
            gdt = 0 if go.dtype == _F32 else 1
            if lib.hd_roi_align_bwd_gather_supported(C.byref(stg), R, Cc, P, int(sampling_ratio), gdt, go.stride(1)):
                ws = torch.empty(lib.hd_roi_align_bwd_gather_workspace_bytes(R, shapes[0][0][0]), dtype=torch.uint8, device=rois.device)
                with _ktimed("roi_align_bwd"):
                    check(lib.hd_roi_align_bwd_gather(C.byref(stg), shapes[0][0][0], _ptr(rois), R, Cc, P, int(sampling_ratio),
                                                      int(bool(aligned)), _ptr(go), gdt, go.stride(0), go.stride(1), go.stride(2),
                                                      go.stride(3), _ptr(ws), ws.numel(), _stream(rois)), "hd_roi_align_bwd_gather")
                return (None,) * 10 + tuple(maps)
        # float32 accumulators for the atomics whatever the feature dtype (a bf16 running sum would round on
        # every add); cast once at the end for bf16 features
        grads = [torch.empty(shape, dtype=_F32, device=rois.device,
                             memory_format=torch.channels_last if cl_fmt else torch.contiguous_format).zero_()
                 for shape, cl_fmt, _ in shapes]
        st = _roi_levels(grads, scales, k_min, cs, cl)
        with _ktimed("roi_align_bwd"):
            check(lib.hd_roi_align_bwd(C.byref(st), _ptr(rois), R, Cc, P, int(sampling_ratio), int(bool(aligned)),
                                       _ptr(grad_out), 0 if grad_out.dtype == _F32 else 1, grad_out.stride(0),
                                       grad_out.stride(1), grad_out.stride(2), grad_out.stride(3), _stream(rois)),
                  "hd_roi_align_bwd")
        grads = [g if dt == _F32 else g.to(dt) for g, (_, _, dt) in zip(grads, shapes)]
        return (None,) * 10 + tuple(grads)


def roi_align(feats: Sequence[torch.Tensor], rois: torch.Tensor, P: int, scales: Sequence[float],
              sampling_ratio: int = 2, aligned: bool = True, k_min: int = 2, canonical_scale: float = 224.0,
              canonical_level: int = 4, channels_last_out: bool = False, out_dtype=None) -> torch.Tensor:
    """feats: list of [B,C,H,W] float32 or bfloat16 maps (any strides; channels-last with sampling_ratio 1 or 2 is
    the fast path); rois [R,5] -> [R,C,P,P] in `out_dtype` (default: the features' dtype), accumulated in float32."""
    out_dtype = feats[0].dtype if out_dtype is None else out_dtype
    if out_dtype not in (_F32, torch.bfloat16):
        raise _lib.HdError("roi_align: out_dtype must be float32 or bfloat16")
    return _RoIAlignFn.apply(rois, int(P), tuple(scales), int(sampling_ratio), bool(aligned), int(k_min),
                             float(canonical_scale), int(canonical_level), bool(channels_last_out), out_dtype, *feats)


class _RoIPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, P, scales, k_min, canonical_scale, canonical_level, channels_last_out, *feats):
        lib = _lib.load()
        rois = _need(rois.detach(), _F32, "rois")
        feats_d = [f.detach() for f in feats]
        st = _roi_levels(feats_d, scales, k_min, canonical_scale, canonical_level)
        R, Cc = rois.shape[0], feats[0].shape[1]
        fmt = torch.channels_last if channels_last_out else torch.contiguous_format
        out = torch.empty((R, Cc, P, P), dtype=_F32, device=rois.device, memory_format=fmt)
        arg = torch.empty((R, Cc, P, P), dtype=_I32, device=rois.device, memory_format=fmt)
        check(lib.hd_roi_pool_fwd(C.byref(st), _ptr(rois), R, Cc, P, _ptr(out), _ptr(arg), out.stride(0),
                                  out.stride(1), out.stride(2), out.stride(3), _stream(rois)), "hd_roi_pool_fwd")
        ctx.save_for_backward(rois, arg)
        ctx.meta = (P, tuple(scales), k_min, canonical_scale, canonical_level,
                    [(tuple(f.shape), f.is_contiguous(memory_format=torch.channels_last)) for f in feats])
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _grad_arg):
        lib = _lib.load()
        rois, arg = ctx.saved_tensors
        P, scales, k_min, cs, cl, shapes = ctx.meta
        # the gradient must share argmax's strides: one index walks both
        go = torch.empty_like(arg, dtype=_F32)
        go.copy_(grad_out)
        grads = [torch.empty(shape, dtype=_F32, device=rois.device,
                             memory_format=torch.channels_last if cl_fmt else torch.contiguous_format).zero_()
                 for shape, cl_fmt in shapes]
        st = _roi_levels(grads, scales, k_min, cs, cl)
        R, Cc = rois.shape[0], shapes[0][0][1]
        check(lib.hd_roi_pool_bwd(C.byref(st), _ptr(rois), R, Cc, P, _ptr(go), _ptr(arg), go.stride(0),
                                  go.stride(1), go.stride(2), go.stride(3), _stream(rois)), "hd_roi_pool_bwd")
        return (None,) * 7 + tuple(grads)


def roi_pool(feats: Sequence[torch.Tensor], rois: torch.Tensor, P: int, scales: Sequence[float], k_min: int = 2,
             canonical_scale: float = 224.0, canonical_level: int = 4, channels_last_out: bool = False,
             return_argmax: bool = False):
    """Quantised RoI max-pooling over pyramid levels (README.md:73 "-RoIPooling" variant).
    feats: list of [B,C,H,W] float32 (channels-last is the fast layout); rois [R,5] -> [R,C,P,P]."""
    out, arg = _RoIPoolFn.apply(rois, int(P), tuple(scales), int(k_min), float(canonical_scale),
                                int(canonical_level), bool(channels_last_out), *feats)
    return (out, arg) if return_argmax else out


# ------------------------------------------------------------------------------- YOLOv5
def _yolo_levels(preds: Sequence[torch.Tensor], strides: Sequence[float], anchors) -> YoloLevels:
    """preds: list of [B, na, H, W, no] views (any strides). anchors: [L][na][2] pixels."""
    st = YoloLevels()
    st.n_levels = len(preds)
    st.na = preds[0].shape[1]
    st.no = preds[0].shape[4]
    for l, p in enumerate(preds):
        if not p.is_cuda or p.dtype != _F32 or p.dim() != 5:
            raise _lib.HdError("yolo ops: predictions must be float32 [B,na,H,W,no] tensors on a HIP device")
        st.pred[l] = p.data_ptr()
        st.h[l], st.w[l] = p.shape[2], p.shape[3]
        st.stride[l] = float(strides[l])
        for a in range(st.na):
            st.anchor_w[l][a] = float(anchors[l][a][0])
            st.anchor_h[l][a] = float(anchors[l][a][1])
        st.sb[l], st.sa[l], st.sy[l], st.sx[l], st.so[l] = p.stride()
    return st


def yolo_decode(preds: Sequence[torch.Tensor], strides, anchors) -> torch.Tensor:
    lib = _lib.load()
    st = _yolo_levels(preds, strides, anchors)
    B = preds[0].shape[0]
    n = sum(p.shape[1] * p.shape[2] * p.shape[3] for p in preds)
    out = torch.empty((B, n, st.no), dtype=_F32, device=preds[0].device)
    with _timed("yolo_decode"):
        check(lib.hd_yolo_decode(C.byref(st), B, _ptr(out), _stream(out)), "hd_yolo_decode")
    return out


def yolo_assign(shapes: Sequence[Tuple[int, int, int, int]], strides, anchors, gt_boxes: torch.Tensor,
                gt_cls: torch.Tensor, m_valid: torch.Tensor, anchor_t: float = 4.0):
    """shapes: per level (na, H, W, no). Returns slot_cell [L,S] i32, slot_tbox [L,S,4] f32, slot_cls [L,S] i32."""
    lib = _lib.load()
    gt_boxes, gt_cls, m_valid = _need(gt_boxes, _F32, "gt_boxes"), _need(gt_cls, _I32, "gt_cls"), _need(m_valid, _I32, "m_valid")
    B, M = gt_cls.shape
    st = YoloLevels()
    st.n_levels, st.na, st.no = len(shapes), shapes[0][0], shapes[0][3]
    for l, (na, H, W, no) in enumerate(shapes):
        st.h[l], st.w[l] = H, W
        st.stride[l] = float(strides[l])
        for a in range(na):
            st.anchor_w[l][a] = float(anchors[l][a][0])
            st.anchor_h[l][a] = float(anchors[l][a][1])
    L, S = len(shapes), 5 * shapes[0][0] * B * M
    dev = gt_boxes.device
    cell = torch.empty((L, S), dtype=_I32, device=dev)
    tbox = torch.empty((L, S, 4), dtype=_F32, device=dev)
    cls = torch.empty((L, S), dtype=_I32, device=dev)
    check(lib.hd_yolo_assign(C.byref(st), _ptr(gt_boxes), _ptr(gt_cls), _ptr(m_valid), B, M, float(anchor_t),
                             _ptr(cell), _ptr(tbox), _ptr(cls), _stream(gt_boxes)), "hd_yolo_assign")
    return cell, tbox, cls


_FOCAL_TERM_BITS = {"cls": 1, "obj": 2}


def _focal_tuple(focal):
    """None | dict(gamma=, alpha=, terms=) | (gamma, alpha, terms) -> hashable (gamma, alpha, bits) or None."""
    if focal is None:
        return None
    if isinstance(focal, dict):
        focal = (focal.get("gamma", 1.5), focal.get("alpha", 0.25), focal.get("terms", ("cls", "obj")))
    gamma, alpha, terms = focal
    bits = terms if isinstance(terms, int) else sum(_FOCAL_TERM_BITS[t] for t in terms)
    if float(gamma) <= 0.0 or bits == 0:
        return None
    return (float(gamma), float(alpha), int(bits))


def _loss_cfg(box_gain, obj_gain, cls_gain, balance, focal=None) -> YoloLossCfg:
    cfg = YoloLossCfg()
    cfg.box_gain, cfg.obj_gain, cfg.cls_gain = float(box_gain), float(obj_gain), float(cls_gain)
    for l, v in enumerate(balance):
        cfg.balance[l] = float(v)
    if focal is not None:
        cfg.focal_gamma, cfg.focal_alpha, cfg.focal_terms = focal
    return cfg


class _YoloLossFn(torch.autograd.Function):
    """total loss (scalar) + detached components; gradients flow to the raw head outputs and, when
    the Detect-conv biases are passed along, straight to those biases (sum over cells computed
    inside the HIP kernels, so autograd never reduces the dense gradient for the bias)."""

    @staticmethod
    def forward(ctx, strides, anchors, slots, M, gains, balance, n_pred, n_root, n_head, *tensors):
        lib = _lib.load()
        preds, roots = tensors[:n_pred], tensors[n_pred:n_pred + n_root]
        heads, biases = tensors[n_pred + n_root:n_pred + n_root + 2 * n_head], tensors[n_pred + n_root + 2 * n_head:]
        preds_d = [p.detach() for p in preds]
        st = _yolo_levels(preds_d, strides, anchors)
        B = preds[0].shape[0]
        slot_cell, slot_tbox, slot_cls = slots
        cfg = _loss_cfg(gains[0], gains[1], gains[2], balance, gains[3] if len(gains) > 3 else None)
        dev = preds[0].device
        out = torch.empty((4,), dtype=_F32, device=dev)
        ws, wsb = _ws(lib.hd_yolo_loss_workspace_bytes(C.byref(st), B), dev)
        with _timed("yolo_loss_fwd"):
            check(lib.hd_yolo_loss_fwd(C.byref(st), B, int(M), _ptr(slot_cell), _ptr(slot_tbox), _ptr(slot_cls),
                                       C.byref(cfg), _ptr(out), _ptr(ws), wsb, _stream(out)), "hd_yolo_loss_fwd")
        # heads: (x, weight) of the Detect convolution that produced each level's padded rows (ops.head_conv): backward then
        # writes the logit gradient as BF16 padded rows and runs the convolutions' backward itself (see yolo_loss)
        ctx.save_for_backward(slot_cell, slot_tbox, slot_cls, ws, *preds_d, *heads)
        # roots: the tensors autograd differentiates when the predictions are (detached) strided views of padded rows
        # (ops.head_conv): gradient buffers then take the root's geometry and the views are re-derived on them
        root_meta = [(tuple(r.shape), tuple(r.stride()), p.storage_offset() - r.storage_offset()) for r, p in zip(roots, preds)]
        ctx.meta = (strides, anchors, int(M), gains, balance, wsb, n_pred,
                    [(tuple(b.shape), b.dtype) for b in biases], root_meta, n_head)
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, grad_total, _grad_parts):
        lib = _lib.load()
        slot_cell, slot_tbox, slot_cls, ws, *saved = ctx.saved_tensors
        strides, anchors, M, gains, balance, wsb, n_pred, bias_meta, root_meta, n_head = ctx.meta
        preds, heads = saved[:n_pred], saved[n_pred:]
        st = _yolo_levels(preds, strides, anchors)
        if n_head:
            return _YoloLossFn._backward_through_heads(ctx, grad_total, st, preds, heads, slot_cell, slot_tbox, slot_cls, ws)
        # gradient tensors keep the producer's memory order (dense channels-last fast path)
        groots = None
        if root_meta:
            # padded rows: the dense kernel writes whole rows, padding included, when they are 16-byte words (yolo.hip,
            # is_dense_nhwc kind 2); any other geometry gets a zeroed buffer
            groots = []
            for sh, sd, _ in root_meta:
                g = torch.empty_strided(sh, sd, dtype=_F32, device=preds[0].device)
                groots.append(g if (sd[1] == 1 and sh[1] % 4 == 0) else g.zero_())
            grads = [g.as_strided(p.shape, p.stride(), off) for g, p, (_, _, off) in zip(groots, preds, root_meta)]
        else:
            grads = [torch.empty_strided(p.shape, p.stride(), dtype=_F32, device=p.device) for p in preds]
        gst = _yolo_levels(grads, strides, anchors)
        B = preds[0].shape[0]
        cfg = _loss_cfg(gains[0], gains[1], gains[2], balance, gains[3] if len(gains) > 3 else None)
        gs = grad_total.detach().to(_F32).reshape(1).contiguous()
        gbias = None
        if bias_meta:
            gbias = torch.zeros((len(preds), st.na * st.no), dtype=_F32, device=gs.device)
        with _timed("yolo_loss_bwd_dense"):
            check(lib.hd_yolo_loss_bwd_dense(C.byref(st), C.byref(gst), B, C.byref(cfg), _ptr(gs), _ptr(gbias),
                                             _ptr(ws), wsb, _stream(gs)), "hd_yolo_loss_bwd_dense")
        check(lib.hd_yolo_loss_bwd_pos(C.byref(st), C.byref(gst), B, M, _ptr(slot_cell), _ptr(slot_tbox),
                                       _ptr(slot_cls), C.byref(cfg), _ptr(gs), _ptr(gbias), _ptr(ws), wsb,
                                       _stream(gs)), "hd_yolo_loss_bwd_pos")
        bias_grads = tuple(gbias[l].view(shape).to(dtype) for l, (shape, dtype) in enumerate(bias_meta))
        if groots is not None:
            return (None,) * 9 + (None,) * n_pred + tuple(groots) + bias_grads
        return (None,) * 9 + tuple(grads) + bias_grads

    @staticmethod
    def _backward_through_heads(ctx, grad_total, st, preds, heads, slot_cell, slot_tbox, slot_cls, ws):
        """The logit gradient as bf16 padded rows (hd_yolo_loss_bwd_*_bf16) and, from them, the Detect convolutions' data and
        weight gradients (hd_head_conv_bwd): no float32 gradient tensor, no cast."""
        lib = _lib.load()
        strides, anchors, M, gains, balance, wsb, n_pred, bias_meta, root_meta, n_head = ctx.meta
        dev = preds[0].device
        B = preds[0].shape[0]
        dys = []
        gst = YoloLevels()
        C.memmove(C.byref(gst), C.byref(st), C.sizeof(YoloLevels))
        for l, p in enumerate(preds):
            _, na, H, W, no = p.shape
            P = p.stride(3)
            dy = torch.empty((B, P, H, W), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
            dys.append(dy)
            gst.pred[l] = dy.data_ptr()
        cfg = _loss_cfg(gains[0], gains[1], gains[2], balance, gains[3] if len(gains) > 3 else None)
        gs = grad_total.detach().to(_F32).reshape(1).contiguous()
        gbias = torch.zeros((len(preds), st.na * st.no), dtype=_F32, device=dev) if bias_meta else None
        with _timed("yolo_loss_bwd_dense"):
            check(lib.hd_yolo_loss_bwd_dense_bf16(C.byref(st), C.byref(gst), B, C.byref(cfg), _ptr(gs), _ptr(gbias), _ptr(ws),
                                                  wsb, _stream(gs)), "hd_yolo_loss_bwd_dense_bf16")
        check(lib.hd_yolo_loss_bwd_pos_bf16(C.byref(st), C.byref(gst), B, M, _ptr(slot_cell), _ptr(slot_tbox), _ptr(slot_cls),
                                            C.byref(cfg), _ptr(gs), _ptr(gbias), _ptr(ws), wsb, _stream(gs)),
              "hd_yolo_loss_bwd_pos_bf16")
        head_grads = []
        for l, dy in enumerate(dys):
            dx, dw = head_conv_bwd_rows(dy, heads[2 * l], heads[2 * l + 1])
            head_grads += [dx, dw]
        bias_grads = tuple(gbias[l].view(shape).to(dtype) for l, (shape, dtype) in enumerate(bias_meta))
        return (None,) * 9 + (None,) * n_pred + tuple(head_grads) + bias_grads


def yolo_loss(preds: Sequence[torch.Tensor], strides, anchors, slots, M: int, box_gain: float = 0.05,
              obj_gain: float = 1.0, cls_gain: float = 0.5, balance=(4.0, 1.0, 0.4),
              biases: Optional[Sequence[torch.Tensor]] = None, focal=None,
              roots: Optional[Sequence[torch.Tensor]] = None, heads: Optional[Sequence[Tuple[torch.Tensor, torch.Tensor]]] = None):
    """Returns (total, parts[4] = (total, lbox, lobj, lcls) detached).

    ``focal``: None, or dict(gamma=1.5, alpha=0.25, terms=("cls", "obj")) / the same as a tuple — the
    "focalloss" variants of README.md:118-154 (alpha <= 0 drops the class-balance factor).

    ``biases``: optionally the per-level Detect-conv bias parameters ([na*no] each) whose values are
    already folded into ``preds`` (the conv ran with ``bias.detach()``); their gradients are then
    produced by the loss kernels.

    ``roots``: one tensor per level of which ``preds[l]`` is a DETACHED strided view (the padded float32 rows of
    ops.head_conv: [B, pitch, H, W] channels-last, viewed as [B, na, H, W, no]); the gradient then flows to the roots,
    allocated with their geometry and written through the same views, padding columns zero.

    ``heads``: instead of ``roots``, per level the (x, weight) the padded rows were computed from by ops.head_conv
    (x [B, Cin, H, W] bf16 channels-last rows, weight the bf16 [Cout, Cin, 1, 1] leaf).  The loss then differentiates the
    Detect convolutions itself: its backward writes the logit gradient as BF16 padded rows (rounded once; cells matched by
    several targets accumulate in bf16 — hd_yolo_loss_bwd_*_bf16) and hands them to hd_head_conv_bwd, so the gradient flows
    to x and weight without a float32 gradient tensor or a cast.  ``preds`` must be the padded-row views of those outputs."""
    biases = tuple(biases) if biases is not None else ()
    roots = tuple(roots) if roots is not None else ()
    heads = tuple(heads) if heads is not None else ()
    if roots and len(roots) != len(preds):
        raise _lib.HdError("yolo_loss: `roots` needs one tensor per prediction level")
    if heads:
        if roots or len(heads) != len(preds):
            raise _lib.HdError("yolo_loss: `heads` needs one (x, weight) pair per prediction level and excludes `roots`")
        for p, (x, w) in zip(preds, heads):
            P = p.stride(3)
            if (p.dim() != 5 or p.stride(4) != 1 or P != head_conv_pitch(int(w.shape[0])) or P % 8
                    or p.shape[1] * p.shape[4] != w.shape[0] or not head_conv_supported(x, w)):
                raise _lib.HdError("yolo_loss: `heads` needs predictions that are views of ops.head_conv's padded rows")
    flat_heads = [t for pair in heads for t in pair]
    detached = bool(roots) or bool(heads)
    return _YoloLossFn.apply(tuple(strides), tuple(tuple(tuple(a) for a in lv) for lv in anchors), slots, int(M),
                             (box_gain, obj_gain, cls_gain, _focal_tuple(focal)), tuple(balance), len(preds), len(roots),
                             len(heads), *[p.detach() for p in preds] if detached else preds, *roots, *flat_heads, *biases)


def time_yolo_loss_bwd_dense(preds: Sequence[torch.Tensor], strides, anchors, slots, M: int, gains, balance,
                             iters: int = 20, bf16_rows: bool = False) -> List[float]:
    """Measurement helper for bench.py: runs the loss forward once on `preds`, then launches the dense gradient pass `iters`
    times on the current stream, each launch stamping its own HIP event pair with the kernel's begin / end
    (hd_profile_time_next_launch).  Returns the per-launch milliseconds.  bf16_rows: the form the training step runs —
    the predictions re-laid as float32 rows padded to head_conv_pitch, the gradient as bf16 padded rows
    (hd_yolo_loss_bwd_dense_bf16); otherwise float32 gradient tensors with the predictions' own strides."""
    lib = _lib.load()
    preds = [p.detach() for p in preds]
    dev = preds[0].device
    if bf16_rows:
        padded = []
        for p in preds:
            B, na, H, W, no = p.shape
            P = head_conv_pitch(na * no)
            root = torch.zeros((B, H, W, P), dtype=_F32, device=dev)
            root[..., : na * no] = p.permute(0, 2, 3, 1, 4).reshape(B, H, W, na * no)
            padded.append(root.as_strided((B, na, H, W, no), (H * W * P, no, W * P, P, 1)))
        preds = padded
    st = _yolo_levels(preds, strides, anchors)
    B = preds[0].shape[0]
    cfg = _loss_cfg(gains[0], gains[1], gains[2], balance)
    out = torch.empty((4,), dtype=_F32, device=dev)
    ws, wsb = _ws(lib.hd_yolo_loss_workspace_bytes(C.byref(st), B), dev)
    s = _stream(out)
    check(lib.hd_yolo_loss_fwd(C.byref(st), B, int(M), _ptr(slots[0]), _ptr(slots[1]), _ptr(slots[2]), C.byref(cfg),
                               _ptr(out), _ptr(ws), wsb, s), "hd_yolo_loss_fwd")
    if bf16_rows:
        grads = [torch.empty((B, p.stride(3), p.shape[2], p.shape[3]), dtype=torch.bfloat16, device=dev,
                             memory_format=torch.channels_last) for p in preds]
        gst = YoloLevels()
        C.memmove(C.byref(gst), C.byref(st), C.sizeof(YoloLevels))
        for l, g in enumerate(grads):
            gst.pred[l] = g.data_ptr()
        fn, name = lib.hd_yolo_loss_bwd_dense_bf16, "hd_yolo_loss_bwd_dense_bf16"
    else:
        grads = [torch.empty_strided(p.shape, p.stride(), dtype=_F32, device=dev) for p in preds]
        gst = _yolo_levels(grads, strides, anchors)
        fn, name = lib.hd_yolo_loss_bwd_dense, "hd_yolo_loss_bwd_dense"
    gs = torch.ones(1, dtype=_F32, device=dev)
    gbias = torch.zeros((len(preds), st.na * st.no), dtype=_F32, device=dev)
    evs = []
    for i in range(iters + 3):
        pair = _arm_kernel_timing()
        check(fn(C.byref(st), C.byref(gst), B, C.byref(cfg), _ptr(gs), _ptr(gbias), _ptr(ws), wsb, s), name)
        if i >= 3:
            evs.append(pair)
    torch.cuda.synchronize(dev)
    return [a.elapsed_time(b) for a, b in evs]


def yolo_detect(preds: Sequence[torch.Tensor], strides, anchors, conf_thr: float = 0.25, iou_thr: float = 0.45,
                max_nms: int = 30000, max_det: int = 300):
    """`yolo_nms(yolo_decode(preds))` without the decoded tensor (csrc/nms.hip: yolo_raw_cand_kernel): the raw head
    outputs [B,na,H,W,no] are gated on sigmoid(objectness) and only passing rows are decoded.  Identical output:
    det [B,max_det,6] = (x1,y1,x2,y2,conf,cls), count [B]."""
    lib = _lib.load()
    st = _yolo_levels([p.detach() for p in preds], strides, anchors)
    B = preds[0].shape[0]
    dev = preds[0].device
    det = torch.empty((B, max_det, 6), dtype=_F32, device=dev)
    cnt = torch.empty((B,), dtype=_I32, device=dev)
    ws, wsb = _ws(lib.hd_yolo_detect_workspace_bytes(C.byref(st), B, int(max_nms), int(max_det)), dev)
    check(lib.hd_yolo_detect(C.byref(st), B, float(conf_thr), float(iou_thr), int(max_nms), int(max_det), _ptr(det),
                             _ptr(cnt), _ptr(ws), wsb, _stream(det)), "hd_yolo_detect")
    return det, cnt


class _CIoUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        lib = _lib.load()
        p, t = _need(pred.detach(), _F32, "pred"), _need(target.detach(), _F32, "target")
        out = torch.empty((p.shape[0],), dtype=_F32, device=p.device)
        check(lib.hd_ciou_fwd(_ptr(p), _ptr(t), p.shape[0], _ptr(out), _stream(p)), "hd_ciou_fwd")
        ctx.save_for_backward(p, t)
        return out

    @staticmethod
    def backward(ctx, grad):
        lib = _lib.load()
        p, t = ctx.saved_tensors
        g = _need(grad, _F32, "grad")
        out = torch.empty_like(p)
        check(lib.hd_ciou_bwd(_ptr(p), _ptr(t), _ptr(g), p.shape[0], _ptr(out), _stream(p)), "hd_ciou_bwd")
        return out, None


def ciou(pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """CIoU of matched xyxy pairs [P,4] -> [P]; differentiable w.r.t. pred."""
    return _CIoUFn.apply(pred, target)


def yolo_nms(dec: torch.Tensor, conf_thr: float = 0.25, iou_thr: float = 0.45, max_nms: int = 30000,
             max_det: int = 300):
    """dec [B,n,no] decoded rows -> (det [B,max_det,6], count [B])."""
    lib = _lib.load()
    dec = _need(dec, _F32, "dec")
    B, n, no = dec.shape
    dev = dec.device
    det = torch.empty((B, max_det, 6), dtype=_F32, device=dev)
    cnt = torch.empty((B,), dtype=_I32, device=dev)
    ws, wsb = _ws(lib.hd_yolo_nms_workspace_bytes(B, n, int(max_nms), int(max_det)), dev)
    check(lib.hd_yolo_nms(_ptr(dec), B, n, no, float(conf_thr), float(iou_thr), int(max_nms), int(max_det), _ptr(det),
                          _ptr(cnt), _ptr(ws), wsb, _stream(dec)), "hd_yolo_nms")
    return det, cnt


# ------------------------------------------------------------------------------- WBF
def wbf(boxes: torch.Tensor, scores: torch.Tensor, labels: torch.Tensor, model_weight: torch.Tensor,
        k_valid: Optional[torch.Tensor], weight_sum: float, iou_thr: float = 0.55, skip_thr: float = 0.0):
    """boxes [B,K,4], scores/labels/model_weight [B,K] -> (out [B,K,6], count [B])."""
    lib = _lib.load()
    boxes, scores = _need(boxes, _F32, "boxes"), _need(scores, _F32, "scores")
    labels, mw = _need(labels, _I32, "labels"), _need(model_weight, _F32, "model_weight")
    if k_valid is not None:
        k_valid = _need(k_valid, _I32, "k_valid")
    B, K = scores.shape
    dev = boxes.device
    out = torch.empty((B, K, 6), dtype=_F32, device=dev)
    cnt = torch.empty((B,), dtype=_I32, device=dev)
    ws, wsb = _ws(lib.hd_wbf_workspace_bytes(B, K), dev)
    check(lib.hd_wbf(_ptr(boxes), _ptr(scores), _ptr(labels), _ptr(mw), _ptr(k_valid), B, K, float(weight_sum),
                     float(iou_thr), float(skip_thr), _ptr(out), _ptr(cnt), _ptr(ws), wsb, _stream(boxes)), "hd_wbf")
    return out, cnt



# ------------------------------------------------------------------------------- Mosaic-4
def mosaic4(images: torch.Tensor, gt_boxes: torch.Tensor, gt_cls: torch.Tensor, m_valid: torch.Tensor,
            src: torch.Tensor, center: torch.Tensor, apply: torch.Tensor, fill: float = 114.0 / 255.0,
            min_wh: float = 2.0, min_area_ratio: float = 0.1):
    """Mosaic-4 on device (csrc/mosaic.hip; convention in DESIGN.md section 2).  images [B,3,S,S] float32 (any
    strides; the output keeps the input's memory format), gt_boxes [B,M,4], gt_cls [B,M] int32, m_valid [B] int32,
    src [B,4] / center [B,2] / apply [B] int32.  -> (images', boxes [B,4M,4], cls [B,4M], count [B])."""
    lib = _lib.load()
    if not images.is_cuda or images.dtype != _F32 or images.dim() != 4 or images.shape[1] != 3 or \
            images.shape[2] != images.shape[3]:
        raise _lib.HdError("mosaic4: images must be a float32 [B,3,S,S] tensor on a HIP device")
    gt_boxes, gt_cls, m_valid = _need(gt_boxes, _F32, "gt_boxes"), _need(gt_cls, _I32, "gt_cls"), _need(m_valid, _I32, "m_valid")
    src, center, apply = _need(src, _I32, "src"), _need(center, _I32, "center"), _need(apply, _I32, "apply")
    B, _, S, _ = images.shape
    M = gt_boxes.shape[1]
    out = torch.empty_like(images)  # preserves channels-last / planar strides
    st = _stream(images)
    check(lib.hd_mosaic4(_ptr(images), *images.stride(), B, S, _ptr(src), _ptr(center), _ptr(apply), float(fill),
                         _ptr(out), *out.stride(), st), "hd_mosaic4")
    ob = torch.empty((B, 4 * M, 4), dtype=_F32, device=images.device)
    oc = torch.empty((B, 4 * M), dtype=_I32, device=images.device)
    cnt = torch.empty((B,), dtype=_I32, device=images.device)
    check(lib.hd_mosaic4_boxes(_ptr(gt_boxes), _ptr(gt_cls), _ptr(m_valid), B, M, S, _ptr(src), _ptr(center),
                               _ptr(apply), float(min_wh), float(min_area_ratio), _ptr(ob), _ptr(oc), _ptr(cnt), st),
          "hd_mosaic4_boxes")
    return out, ob, oc, cnt


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]   # private helpers included: the sibling modules and the tests use them
