"""Tensor-level front end of the HIP C-ABI (include/hd_ops.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every op below
hands raw pointers to libhd_ops.so.  There is NO CPU path: tensors must live on a HIP device
and the library must have been built (``__graft_entry__.build()``), otherwise these raise.

The op set mirrors what a detector framework of the reference's kind plugs into its
heads/assigners/post-processing (reference README.md:8-9,19,65): anchors, pairwise IoU +
assignment, delta codec, NMS, RPN proposals, RoIAlign, YOLOv5 decode / build-targets / loss,
CIoU, per-class NMS, WBF.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import RoiLevels, Rows2, YoloLevels, YoloLossCfg, check

_F32, _I32 = torch.float32, torch.int32


def _stream(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need(t: torch.Tensor, dtype, name: str, contiguous: bool = True) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.HdError(f"{name}: expected a tensor on a HIP device (there is no CPU path)")
    if t.dtype != dtype:
        raise _lib.HdError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        t = t.contiguous()
    return t


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


# ---- optional HIP-event timing of individual C-ABI launches (bench.py's roofline leg) ---------
# Events are recorded on the stream the kernel is launched on (torch's current stream, the one
# handed to the C ABI), immediately around the call; nothing is synchronised here.
_TIMED: dict = {}
_TIMED_BYTES: dict = {}
_TIMED_TAGS: dict = {}


def enable_timing(*names: str) -> None:
    _TIMED.clear()
    _TIMED_BYTES.clear()
    _TIMED_TAGS.clear()
    for n in names:
        _TIMED[n] = []
        _TIMED_BYTES[n] = []
        _TIMED_TAGS[n] = []


def disable_timing() -> None:
    _TIMED.clear()
    _TIMED_BYTES.clear()
    _TIMED_TAGS.clear()


def _note_bytes(name: str, nbytes: int, tag=None) -> None:
    """Algorithmic bytes (and a shape tag) of a timed launch, noted by ops that know their own operand sizes, in launch
    order — bench.py's step-weighted rooflines pair them with the launch's event times."""
    if name in _TIMED_BYTES:
        _TIMED_BYTES[name].append(int(nbytes))
        _TIMED_TAGS[name].append(tag)


def collect_bytes() -> dict:
    return {n: list(v) for n, v in _TIMED_BYTES.items()}


def collect_tags() -> dict:
    return {n: list(v) for n, v in _TIMED_TAGS.items()}


def collect_timing() -> dict:
    """After a device synchronise: {name: [ms, ...]} for every timed launch since enable_timing()."""
    return {n: [a.elapsed_time(b) for a, b in ev] for n, ev in _TIMED.items()}


class _timed:
    """Event pair around a whole C-ABI call (ops that launch several kernels)."""
    __slots__ = ("name", "start")

    def __init__(self, name: str):
        self.name = name if name in _TIMED else None

    def __enter__(self):
        if self.name is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.name is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            _TIMED[self.name].append((self.start, end))
        return False


def _arm_kernel_timing():
    """Arms libhd_ops so that the NEXT launch of an hd_launch() kernel stamps a fresh event pair with the kernel's own
    begin / end (hd_profile_time_next_launch: the dispatch's timestamps, what a profiler reports).  Returns the pair."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()  # creates the hipEvent_t handles (torch creates them lazily at the first record)
    b.record()
    check(_lib.load().hd_profile_time_next_launch(C.c_void_p(a.cuda_event), C.c_void_p(b.cuda_event)), "hd_profile_time_next_launch")
    return a, b


class _ktimed:
    """Kernel-only timing of a single-kernel C-ABI call: the event pair is stamped by the launch itself."""
    __slots__ = ("name", "pair")

    def __init__(self, name: str):
        self.name = name if name in _TIMED else None

    def __enter__(self):
        if self.name is not None:
            self.pair = _arm_kernel_timing()
        return self

    def __exit__(self, *exc):
        if self.name is not None:
            _lib.load().hd_profile_time_next_launch(None, None)  # in case the call did not launch (empty input)
            _TIMED[self.name].append(self.pair)
        return False


def _ws(nbytes: int, device) -> Tuple[Optional[torch.Tensor], int]:
    if nbytes <= 0:
        return None, 0
    t = torch.empty((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=device)
    return t, t.numel()


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


# ------------------------------------------------------------------------------- SPPF max-pool
class _OutSlot:
    """Non-tensor wrapper (so autograd ignores it) around the tensor a custom Function writes its result into."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        self.t = t


class CatBuffer:
    """Destination of a channel concatenation whose inputs are written IN PLACE by the kernels that produce them
    (fused BN+act, SPPF max-pool) instead of being copied there by torch.cat.

        buf = ops.CatBuffer(like, [c0, c1, ...])      # one channels-last allocation [B, sum(c), H, W]
        y0 = bn_act(..., out=buf.slot(0)); y1 = ...   # producers take the slot as `out`
        z = buf.join([y0, y1, ...])                   # parts already in their slot cost nothing; others are copied in

    slot(i) and join() hand out FRESH tensor objects over the buffer's storage (Tensor.set_ on the storage — not
    views), so autograd sees ordinary independent tensors and no view / version-counter bookkeeping is involved; the
    gradient of join() is the channel slices of the incoming gradient (views; the BN backward kernels read a strided
    dy in place)."""

    def __init__(self, like: torch.Tensor, channels: Sequence[int]):
        B, _, H, W = like.shape
        self.channels = [int(c) for c in channels]
        self.offsets = [0]
        for c in self.channels:
            self.offsets.append(self.offsets[-1] + c)
        self.buf = torch.empty((B, self.offsets[-1], H, W), dtype=like.dtype, device=like.device,
                               memory_format=torch.channels_last)

    def _alias(self, off: int, c: int) -> torch.Tensor:
        b = self.buf
        B, _, H, W = b.shape
        t = torch.empty(0, dtype=b.dtype, device=b.device)
        t.set_(b.untyped_storage(), b.storage_offset() + off, (B, c, H, W), b.stride())
        return t

    def slot(self, i: int) -> torch.Tensor:
        return self._alias(self.offsets[i], self.channels[i])

    def in_place(self, i: int, t: torch.Tensor) -> bool:
        return (t.data_ptr() == self.buf.data_ptr() + self.offsets[i] * self.buf.element_size()
                and t.shape[1] == self.channels[i] and t.stride() == self.buf.stride() and t.dtype == self.buf.dtype)

    def join(self, parts: Sequence[torch.Tensor]) -> torch.Tensor:
        if len(parts) != len(self.channels):
            raise _lib.HdError("CatBuffer.join: one tensor per slot")
        return _CatJoinFn.apply(self, *parts)


class _CatJoinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cb: "CatBuffer", *parts):
        for i, p in enumerate(parts):
            if not cb.in_place(i, p):
                cb.slot(i).copy_(p)  # a producer that could not write in place (or a foreign tensor): torch.cat's copy
        ctx.offsets = cb.offsets
        return cb._alias(0, cb.offsets[-1])

    @staticmethod
    def backward(ctx, g):
        o = ctx.offsets
        return (None,) + tuple(g[:, o[i]:o[i + 1]] for i in range(len(o) - 1))


class _MaxPool5Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_slot):
        lib = _lib.load()
        B, Cc, H, W = x.shape
        dt = {torch.float32: 0, torch.bfloat16: 1}[x.dtype]
        vec = 4 if dt == 0 else 8
        ld_in = _cl_row_stride(x, vec)
        out = out_slot.t if out_slot is not None else torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device,
                                                                  memory_format=torch.channels_last)
        ld_out = _cl_row_stride(out, vec)
        if ld_in is None or ld_out is None:
            raise _lib.HdError("maxpool5: input / output must be (channel slices of) channels-last tensors")
        idx = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=x.device)
        check(lib.hd_maxpool5_fwd(_ptr(x), int(ld_in), dt, B, H, W, Cc, _ptr(out), int(ld_out), _ptr(idx), _stream(x)),
              "hd_maxpool5_fwd")
        ctx.save_for_backward(idx)
        ctx.meta = (dt, B, H, W, Cc)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        dt, B, H, W, Cc = ctx.meta
        ld = _cl_row_stride(g, 4 if dt == 0 else 8)
        if ld is None or g.dtype != (torch.float32 if dt == 0 else torch.bfloat16):
            g = g.to(torch.float32 if dt == 0 else torch.bfloat16).contiguous(memory_format=torch.channels_last)
            ld = Cc
        gin = torch.empty((B, Cc, H, W), dtype=g.dtype, device=g.device, memory_format=torch.channels_last)
        check(lib.hd_maxpool5_bwd(_ptr(g), int(ld), _ptr(idx), dt, B, H, W, Cc, _ptr(gin), _stream(g)), "hd_maxpool5_bwd")
        return gin, None


def maxpool5_supported(x: torch.Tensor) -> bool:
    """HIP path: HIP tensor, channels-last (dense, or a channel slice of a wider channels-last tensor), float32
    (C % 4 == 0) or bfloat16 (C % 8 == 0)."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    vec = 4 if x.dtype == torch.float32 else 8
    return x.shape[1] % vec == 0 and _cl_row_stride(x, vec) is not None


def maxpool5(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """5x5 / stride 1 / pad 2 max-pool on a channels-last HIP tensor (csrc/pool.hip): forward records the winning
    tap, backward is a gather (no atomics).  out: a CatBuffer slot to write the result into."""
    if not maxpool5_supported(x):
        raise _lib.HdError("maxpool5: needs a channels-last float32/bfloat16 HIP tensor")
    return _MaxPool5Fn.apply(x, _OutSlot(out) if out is not None else None)


class _MaxPool3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        B, Cc, H, W = x.shape
        dt = {torch.float32: 0, torch.bfloat16: 1}[x.dtype]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty((B, Cc, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((B, Ho, Wo, Cc), dtype=torch.uint8, device=x.device)
        check(lib.hd_maxpool3s2_fwd(_ptr(x), dt, B, H, W, Cc, _ptr(out), _ptr(idx), _stream(x)), "hd_maxpool3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.meta = (dt, B, H, W, Cc)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        dt, B, H, W, Cc = ctx.meta
        want = torch.float32 if dt == 0 else torch.bfloat16
        if g.dtype != want or not g.is_contiguous(memory_format=torch.channels_last):
            g = g.to(want).contiguous(memory_format=torch.channels_last)
        gin = torch.empty((B, Cc, H, W), dtype=g.dtype, device=g.device, memory_format=torch.channels_last)
        check(lib.hd_maxpool3s2_bwd(_ptr(g), _ptr(idx), dt, B, H, W, Cc, _ptr(gin), _stream(g)), "hd_maxpool3s2_bwd")
        return gin


def maxpool3s2_supported(x: torch.Tensor) -> bool:
    """HIP path of the 3x3 / stride 2 / pad 1 max-pool: a dense channels-last float32 (C % 4 == 0) / bfloat16 (C % 8 == 0) tensor."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    vec = 4 if x.dtype == torch.float32 else 8
    return x.shape[1] % vec == 0 and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0


def maxpool3s2(x: torch.Tensor) -> torch.Tensor:
    """3x3 / stride 2 / pad 1 max-pool (the ResNet stem) on a dense channels-last HIP tensor (csrc/pool.hip): one byte of tap
    index per output element, gather backward without atomics; values and gradients equal F.max_pool2d(x, 3, 2, 1)."""
    if not maxpool3s2_supported(x):
        raise _lib.HdError("maxpool3s2: needs a dense channels-last float32/bfloat16 HIP tensor")
    return _MaxPool3s2Fn.apply(x)


# ------------------------------------------------------------------------------- fused BN + act
_ACT_CODE = {None: 0, "none": 0, "silu": 1, "relu": 2}


class _StatArena:
    """One zero-initialised fp32 arena per device for the per-channel sums the BN kernels accumulate
    with atomics.  ``reset()`` clears every slot with ONE kernel (the Trainer calls it once per
    step); a layer that runs twice without a reset in between clears its own slot."""

    def __init__(self, device):
        self.device = device
        self.chunks: List[torch.Tensor] = []
        self.used = 0
        self.gen = 0
        self.chunk_elems = 1 << 20

    def alloc(self, n: int) -> "_StatSlot":
        n = (n + 63) // 64 * 64
        if not self.chunks or self.used + n > self.chunk_elems:
            self.chunks.append(torch.zeros(max(self.chunk_elems, n), dtype=_F32, device=self.device))
            self.used = 0
        buf = self.chunks[-1][self.used:self.used + n]
        self.used += n
        return _StatSlot(self, buf)

    def reset(self):
        self.gen += 1
        for c in self.chunks:
            c.zero_()


class _StatSlot:
    __slots__ = ("arena", "buf", "fwd_gen", "bwd_gen")

    def __init__(self, arena, buf):
        self.arena, self.buf, self.fwd_gen, self.bwd_gen = arena, buf, -1, -1


_ARENAS: dict = {}


def _arena(device) -> _StatArena:
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _ARENAS:
        _ARENAS[key] = _StatArena(device)
    return _ARENAS[key]


def bn_arena_reset(device) -> None:
    """Zero every BN statistics slot on `device` (one kernel).  Call once per training step."""
    _arena(torch.device(device)).reset()


def bn_act_supported(x: torch.Tensor) -> bool:
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    v = 4 if x.dtype == torch.float32 else 8
    C = x.shape[1]
    return C % v == 0 and C // v <= 256 and x.is_contiguous(memory_format=torch.channels_last)


def _cl_row_stride(t: torch.Tensor, vec: int):
    """Row stride (elements) of a [B,C,H,W] tensor whose memory is channels-last rows of >= C channels —
    dense channels-last, or a channel slice of a wider channels-last tensor (a torch.cat input's gradient).
    None when the layout is anything else or the 16-byte vector accesses would be misaligned."""
    B, Cc, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if Cc > 1 and sc != 1:
        return None
    ld = sw if W > 1 else (sh if H > 1 else (sb if B > 1 else Cc))
    if ld < Cc or ld % vec != 0:
        return None
    if (W > 1 and sw != ld) or (H > 1 and sh != W * ld) or (B > 1 and sb != H * W * ld):
        return None
    if (t.data_ptr() % 16) != 0:
        return None
    return ld


# ------------------------------------------------------------------------------- upsample 2x + concat
class _UpCatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, small, big):
        lib = _lib.load()
        B, Cs, Hs, Ws = small.shape
        Cb, H, W = big.shape[1], big.shape[2], big.shape[3]
        dt = 0 if small.dtype == torch.float32 else 1
        vec = 4 if dt == 0 else 8
        out = torch.empty((B, Cs + Cb, H, W), dtype=small.dtype, device=small.device, memory_format=torch.channels_last)
        check(lib.hd_upsample2x_cat_fwd(_ptr(small), int(_cl_row_stride(small, vec)), _ptr(big), int(_cl_row_stride(big, vec)), dt,
                                        B, H, W, Cs, Cb, _ptr(out), Cs + Cb, _stream(small)), "hd_upsample2x_cat_fwd")
        ctx.meta = (dt, B, H, W, Cs)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        dt, B, H, W, Cs = ctx.meta
        vec = 4 if dt == 0 else 8
        want = torch.float32 if dt == 0 else torch.bfloat16
        if g.dtype != want or _cl_row_stride(g, vec) is None:
            g = g.to(want).contiguous(memory_format=torch.channels_last)
        d_small = None
        if ctx.needs_input_grad[0]:
            d_small = torch.empty((B, Cs, H // 2, W // 2), dtype=want, device=g.device, memory_format=torch.channels_last)
            check(lib.hd_upsample2x_cat_bwd(_ptr(g), int(_cl_row_stride(g, vec)), dt, B, H, W, Cs, _ptr(d_small), Cs, _stream(g)),
                  "hd_upsample2x_cat_bwd")
        return d_small, (g[:, Cs:] if ctx.needs_input_grad[1] else None)


def upsample2x_cat_supported(small: torch.Tensor, big: torch.Tensor) -> bool:
    if not (small.is_cuda and big.is_cuda and small.dim() == 4 and big.dim() == 4 and small.dtype == big.dtype
            and small.dtype in (torch.float32, torch.bfloat16)):
        return False
    vec = 4 if small.dtype == torch.float32 else 8
    return (big.shape[0] == small.shape[0] and big.shape[2] == 2 * small.shape[2] and big.shape[3] == 2 * small.shape[3]
            and small.shape[1] % vec == 0 and big.shape[1] % vec == 0
            and _cl_row_stride(small, vec) is not None and _cl_row_stride(big, vec) is not None)


def upsample2x_cat(small: torch.Tensor, big: torch.Tensor) -> torch.Tensor:
    """cat(nearest-upsample-2x(small), big) along channels, one kernel, channels-last (csrc/upcat.hip)."""
    if not upsample2x_cat_supported(small, big):
        raise _lib.HdError("upsample2x_cat: needs same-dtype channels-last HIP tensors with big = 2x small spatially")
    return _UpCatFn.apply(small, big)


# ------------------------------------------------------------------------------- 1x1 convolution (MFMA GEMM)
def pointwise_conv_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """x: [B, Cin, H, W] bf16 channels-last rows (dense or a channel slice); weight: [Cout, Cin, 1, 1] bf16 dense."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and weight.dim() == 4 and weight.shape[2:] == (1, 1) and weight.shape[1] == x.shape[1]):
        return False
    B, Cin, H, W = x.shape
    if not _lib.load().hd_pointwise_conv_supported(B * H * W, Cin, weight.shape[0]):
        return False
    return _cl_row_stride(x, 8) is not None and weight.is_contiguous() and weight.data_ptr() % 16 == 0


def pointwise_conv_rows(x: torch.Tensor, w2: torch.Tensor, stats: Optional[torch.Tensor] = None,
                        out: Optional[torch.Tensor] = None, name: str = "pointwise_conv") -> torch.Tensor:
    """y[B, N, H, W] = 1x1 convolution of x[B, K, H, W] (channels-last rows) with w2[N, K] (bf16, dense), no autograd.
    stats: float32 [replicas * 2 * N], zeroed by the caller — the fused BatchNorm's (sum, sum of squares) are added."""
    lib = _lib.load()
    B, K, H, W = x.shape
    N = w2.shape[0]
    ld_x = _cl_row_stride(x, 8)
    if ld_x is None or x.dtype != torch.bfloat16 or w2.dtype != torch.bfloat16 or w2.shape[1] != K or not w2.is_contiguous():
        raise _lib.HdError("pointwise_conv: needs bf16 channels-last rows and a dense bf16 [Cout, Cin] weight")
    if out is None:
        out = torch.empty((B, N, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    ld_y = _cl_row_stride(out, 8)
    if ld_y is None or out.shape != (B, N, H, W) or out.dtype != torch.bfloat16:
        raise _lib.HdError("pointwise_conv: `out` must be a bf16 channels-last tensor (or channel slice) [B, Cout, H, W]")
    _note_bytes(name, 2 * (B * H * W * (K + N) + N * K), {"flops": 2 * B * H * W * K * N, "shape": (B * H * W, K, N)})
    with _ktimed(name):
        check(lib.hd_pointwise_conv(_ptr(x), int(ld_x), _ptr(w2), B * H * W, K, N, _ptr(out), int(ld_y), _ptr(stats), _stream(x)),
              "hd_pointwise_conv")
    return out


def _fwd_sums(slot: "_StatSlot", channels: int) -> torch.Tensor:
    """The slot's forward accumulators [replicas, 2, C], zero: the arena reset zeroed them once per step; a second
    forward of the same layer since then clears its own."""
    sums = slot.buf[: _lib.load().hd_bn_act_replicas() * 2 * channels]
    if slot.fwd_gen == slot.arena.gen:
        sums.zero_()
    slot.fwd_gen = slot.arena.gen
    return sums


class _WgradChain:
    """Weight gradients are computed as row-sliced float32 partial products (a slab) plus their sum over the slices.  Inside
    a training step (`enabled`, set by train.Trainer around backward) the sum of layer L does not get a launch of its own:
    it rides in extra workgroups of the NEXT weight-gradient launch (hd_wgrad_job `piggyback`), and only the last one of a
    backward pass is launched by itself (`flush`, called by dist.GradReducer before it reads any gradient and by the Trainer
    after backward).  Outside a step every weight gradient is summed at once, as before.

    Slabs: two per device, used alternately — the pending job's slab must survive until the next launch has consumed it, and
    that launch writes the other one.  A slab that has to grow is replaced, never released (a captured hipGraph may still
    point at the old one: ADVICE r02), so the superseded buffers stay referenced for the life of the process."""

    enabled = False
    _slabs: dict = {}      # device key -> [buf0, buf1]
    _retired: list = []    # superseded slabs, kept alive
    _pending: dict = {}    # device key -> (WgradJob, keep-alive tensors)
    _handed: list = []     # (weakref to the weight leaf, dw.data_ptr()) of the chained gradients since the last flush

    @staticmethod
    def _key(device):
        return (device.type, device.index if device.index is not None else torch.cuda.current_device())

    @classmethod
    def slab(cls, device, nbytes: int) -> torch.Tensor:
        """A slab of >= nbytes that is NOT the one the pending job's partial products live in."""
        key = cls._key(device)
        pair = cls._slabs.setdefault(key, [None, None])
        pend = cls._pending.get(key)
        busy = pend[1][0] if pend is not None else None
        t = 1 if (busy is not None and pair[0] is busy) else 0
        buf = pair[t]
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                cls._retired.append(buf)
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            pair[t] = buf
        return buf

    @staticmethod
    def _reduce_now(job, device) -> None:
        """The slice sum as a launch of its own (noted for bench.py's per-launch rooflines: slab in, bf16 gradient out)."""
        _note_bytes("wgrad_reduce", int(job.slices) * int(job.count) * 4 + int(job.count) * 2, {"flops": int(job.slices) * int(job.count)})
        with _ktimed("wgrad_reduce"):
            check(_lib.load().hd_wgrad_reduce(C.byref(job), C.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                  "hd_wgrad_reduce")

    @classmethod
    def piggyback(cls, device):
        """The pending job of `device` (to ride in the launch being prepared), or None."""
        pend = cls._pending.pop(cls._key(device), None)
        return pend

    @classmethod
    def submit(cls, device, job, keep, leaf=None) -> None:
        """`job` (a filled WgradJob) has been launched: chain it, or sum it right away.  Chained only inside a training step
        AND when the gradient goes to an autograd LEAF (the bf16 weight shadow): AccumulateGrad takes a leaf's first gradient
        as it is and nothing reads it before the bucket is packed; a non-leaf weight (autocast's cast of a float32 master)
        has its gradient cast back by the very next autograd node, which must see the finished sum."""
        if cls.enabled and leaf is not None and leaf.is_leaf:
            cls._pending[cls._key(device)] = (job, keep)
            if leaf is not None:
                import weakref
                cls._handed.append((weakref.ref(leaf), int(job.dw)))
        else:
            cls._reduce_now(job, device)

    @classmethod
    def flush_device(cls, device) -> None:
        """Sum the pending job of one device now (a launch that cannot carry it is about to use the stream)."""
        pend = cls._pending.pop(cls._key(device), None)
        if pend is not None:
            cls._reduce_now(pend[0], device)

    @classmethod
    def reset(cls) -> None:
        """Forget the pending slice sum and the hand-over records WITHOUT launching anything: for a step that ended in an
        exception (a failed hipGraph capture above all — the pending job's gradient pointer then lies in the destroyed capture's
        private pool, and a later launch that carried it would write through freed memory).  The gradients of the
        aborted step are lost, which is what an aborted step means."""
        cls._pending.clear()
        cls._handed.clear()

    @classmethod
    def flush(cls) -> None:
        for key, (job, keep) in list(cls._pending.items()):
            cls._reduce_now(job, torch.device(key[0], key[1]))
        cls._pending.clear()
        # a chained gradient is complete only after the launch that carried its sum: autograd must have taken the returned
        # tensor as it is (AccumulateGrad steals a fresh dense gradient); a copy taken earlier would hold unsummed bytes
        for ref, ptr in cls._handed:
            leaf = ref()
            if leaf is not None and leaf.grad is not None and leaf.grad.data_ptr() != ptr:
                cls._handed.clear()
                raise _lib.HdError("a chained weight gradient was copied before its slices were summed (autograd did not take "
                                   "the tensor as returned); run with ops._WgradChain.enabled = False")
        cls._handed.clear()


def flush_wgrad() -> None:
    """Sum the partial products of the last weight-gradient launch (no-op when nothing is pending): after this call every
    weight gradient handed out so far is complete for the current stream."""
    _WgradChain.flush()


def _wgrad_workspace(device, nbytes: int) -> torch.Tensor:
    """One of the device's two float32 slabs (see _WgradChain), at least `nbytes` long."""
    return _WgradChain.slab(device, nbytes)


def pointwise_dgrad_rows(dy: torch.Tensor, w2: torch.Tensor, out: Optional[torch.Tensor] = None,
                         add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx[B, Cin, H, W] = dy[B, Cout, H, W] . w2[Cout, Cin] on channels-last rows, no autograd; Cout in {32, 64, 128, 256}.
    add: bf16 channels-last rows [B, Cin, H, W] summed into the result in the kernel's epilogue (the gradient that reaches
    the same input through a shortcut) — float32 add of the two bf16 values, one rounding, like a separate bf16 add."""
    lib = _lib.load()
    B, _, H, W = dy.shape
    N = w2.shape[0]
    K = w2.shape[1]
    ld_dy = _cl_row_stride(dy, 8)
    if ld_dy is None or dy.dtype != torch.bfloat16 or w2.dtype != torch.bfloat16 or not w2.is_contiguous() \
            or dy.shape[1] != N:
        raise _lib.HdError("pointwise_dgrad: needs bf16 channels-last rows and the dense bf16 [Cout, Cin] forward weight")
    if out is None:
        out = torch.empty((B, K, H, W), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    ld_dx = _cl_row_stride(out, 8)
    if ld_dx is None or out.shape != (B, K, H, W) or out.dtype != torch.bfloat16:
        raise _lib.HdError("pointwise_dgrad: `out` must be a bf16 channels-last tensor (or channel slice) [B, Cin, H, W]")
    _note_bytes("pointwise_dgrad", 2 * (B * H * W * (K + N) + N * K),
                {"flops": 2 * B * H * W * K * N, "shape": (B * H * W, N, K)})
    if add is not None:
        ld_add = _cl_row_stride(add, 8)
        if ld_add is None or add.dtype != torch.bfloat16 or tuple(add.shape) != (B, K, H, W):
            raise _lib.HdError("pointwise_dgrad: `add` must be bf16 channels-last rows of the result's shape")
        with _ktimed("pointwise_dgrad"):
            check(lib.hd_pointwise_conv_dgrad_add(_ptr(dy), int(ld_dy), _ptr(w2), B * H * W, K, N, _ptr(add), int(ld_add), _ptr(out),
                                                  int(ld_dx), _stream(dy)), "hd_pointwise_conv_dgrad_add")
        return out
    with _ktimed("pointwise_dgrad"):
        check(lib.hd_pointwise_conv_dgrad(_ptr(dy), int(ld_dy), _ptr(w2), B * H * W, K, N, _ptr(out), int(ld_dx), _stream(dy)),
              "hd_pointwise_conv_dgrad")
    return out


def pointwise_wgrad_supported(R: int, cin: int, cout: int) -> bool:
    return bool(_lib.load().hd_pointwise_wgrad_supported(int(R), int(cin), int(cout)))


def pointwise_wgrad_rows(dy: torch.Tensor, x: torch.Tensor, leaf: Optional[torch.Tensor] = None,
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dw[Cout, Cin] (bf16) = sum over rows of dy[B, Cout, H, W]^T x[B, Cin, H, W], channels-last rows, no autograd.
    leaf: the tensor autograd will hand the result to (checked by the chain, see _WgradChain).  Inside a training step the
    returned tensor is complete only after the next weight-gradient launch or flush_wgrad()."""
    lib = _lib.load()
    B, K, H, W = x.shape
    N = int(dy.shape[1])
    ld_dy, ld_x = _cl_row_stride(dy, 8), _cl_row_stride(x, 8)
    if (ld_dy is None or ld_x is None or dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16
            or x.shape[0] != dy.shape[0] or x.shape[2:] != dy.shape[2:]):
        raise _lib.HdError("pointwise_wgrad: needs bf16 channels-last rows of equal batch and spatial size")
    R = B * H * W
    nbytes = lib.hd_pointwise_wgrad_workspace_bytes(R, K, N)
    if nbytes == 0:
        raise _lib.HdError("pointwise_wgrad: Cin and Cout must be multiples of 64, or multiples of 32 up to 256")
    ws = _wgrad_workspace(dy.device, nbytes)
    dw = out if out is not None else torch.empty((N, K), dtype=torch.bfloat16, device=dy.device)
    if dw.dtype != torch.bfloat16 or dw.numel() != N * K or not dw.is_contiguous():
        raise _lib.HdError("pointwise_wgrad: `out` must be a dense bf16 tensor of Cout x Cin elements")
    pend = _WgradChain.piggyback(dy.device)
    job = _lib.WgradJob()
    _note_bytes("pointwise_wgrad", 2 * (R * (K + N) + N * K), {"flops": 2 * R * K * N, "shape": (R, K, N)})
    with _ktimed("pointwise_wgrad"):
        check(lib.hd_pointwise_wgrad_partial(_ptr(dy), int(ld_dy), _ptr(x), int(ld_x), R, K, N, _ptr(dw),
                                             _ptr(ws), ws.numel(), C.byref(pend[0]) if pend is not None else None, C.byref(job),
                                             _stream(dy)), "hd_pointwise_wgrad_partial")
    _WgradChain.submit(dy.device, job, (ws,), leaf)  # not dw: a second reference would make autograd copy it
    return dw


class PwBwdSwitch:
    """The one-launch backward of the 1x1 layers (csrc/pointwise_bwd.hip: data gradient + weight-gradient partials from one
    staging of dy).  HD_FUSED_PW_BWD=0 goes back to the pair hd_pointwise_conv_dgrad + hd_pointwise_wgrad_partial."""
    enabled = os.environ.get("HD_FUSED_PW_BWD", "1") != "0"


def pointwise_bwd_supported(R: int, cin: int, cout: int) -> bool:
    return PwBwdSwitch.enabled and bool(_lib.load().hd_pointwise_bwd_supported(int(R), int(cin), int(cout)))


def pointwise_bwd_rows(dy: torch.Tensor, w2: torch.Tensor, x: torch.Tensor, add: Optional[torch.Tensor] = None,
                       leaf: Optional[torch.Tensor] = None, out_dx: Optional[torch.Tensor] = None,
                       out_dw: Optional[torch.Tensor] = None):
    """(dx, dw) of a 1x1 convolution y = x . w2^T on channels-last rows, ONE launch, no autograd: dx[B, Cin, H, W] = dy . w2
    (+ add), dw[Cout, Cin] = dy^T . x (bf16; complete after the next weight-gradient launch or flush_wgrad(), see
    _WgradChain).  dy [B, Cout, H, W], x [B, Cin, H, W] bf16 channels-last rows; w2 the dense bf16 [Cout, Cin] forward weight."""
    lib = _lib.load()
    B, N, H, W = dy.shape
    K = w2.shape[1]
    R = B * H * W
    ld_dy, ld_x = _cl_row_stride(dy, 8), _cl_row_stride(x, 8)
    if (ld_dy is None or ld_x is None or dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or w2.dtype != torch.bfloat16
            or not w2.is_contiguous() or tuple(w2.shape) != (N, K) or tuple(x.shape) != (B, K, H, W)):
        raise _lib.HdError("pointwise_bwd: needs bf16 channels-last rows dy [B, Cout, H, W], x [B, Cin, H, W] and the dense bf16 "
                           "[Cout, Cin] forward weight")
    nbytes = lib.hd_pointwise_bwd_workspace_bytes(R, K, N)
    if nbytes == 0:
        raise _lib.HdError("pointwise_bwd: Cout must be 32 / 64 / 128 / 256 and Cin a multiple of 32")
    dx = out_dx if out_dx is not None else torch.empty((B, K, H, W), dtype=torch.bfloat16, device=dy.device,
                                                       memory_format=torch.channels_last)
    ld_dx = _cl_row_stride(dx, 8)
    if ld_dx is None or tuple(dx.shape) != (B, K, H, W) or dx.dtype != torch.bfloat16:
        raise _lib.HdError("pointwise_bwd: `out_dx` must be a bf16 channels-last tensor (or channel slice) [B, Cin, H, W]")
    ld_add = 0
    if add is not None:
        ld_add = _cl_row_stride(add, 8)
        if ld_add is None or add.dtype != torch.bfloat16 or tuple(add.shape) != (B, K, H, W):
            raise _lib.HdError("pointwise_bwd: `add` must be bf16 channels-last rows of dx's shape")
    dw = out_dw if out_dw is not None else torch.empty((N, K), dtype=torch.bfloat16, device=dy.device)
    if dw.dtype != torch.bfloat16 or dw.numel() != N * K or not dw.is_contiguous():
        raise _lib.HdError("pointwise_bwd: `out_dw` must be a dense bf16 tensor of Cout x Cin elements")
    ws = _wgrad_workspace(dy.device, nbytes)
    pend = _WgradChain.piggyback(dy.device)
    job = _lib.WgradJob()
    # each operand once: dy, x in; dx out (+ the addend in); w and the bf16 dw
    _note_bytes("pointwise_bwd", 2 * (R * (N + 2 * K) + 2 * N * K) + (2 * R * K if add is not None else 0),
                {"flops": 4 * R * K * N, "shape": (R, K, N)})
    with _ktimed("pointwise_bwd"):
        check(lib.hd_pointwise_bwd(_ptr(dy), int(ld_dy), _ptr(w2), _ptr(x), int(ld_x), R, K, N, _ptr(add), int(ld_add), _ptr(dx),
                                   int(ld_dx), _ptr(dw), _ptr(ws), ws.numel(), C.byref(pend[0]) if pend is not None else None,
                                   C.byref(job), _stream(dy)), "hd_pointwise_bwd")
    _WgradChain.submit(dy.device, job, (ws,), leaf)
    return dx, dw


# ------------------------------------------------------------------------------- Detect convolution (YOLOv5 head)
class HeadConvSwitch:
    """The Detect convolutions on the HIP kernels (hd_head_conv_fwd / hd_head_conv_bwd, padded float32 rows).  HD_HEAD_CONV=0:
    the library's biased convolution + cast, as in rounds 1-3."""
    enabled = os.environ.get("HD_HEAD_CONV", "1") != "0"
    # the loss writes its logit gradient as bf16 padded rows and runs the Detect convolutions' backward itself (yolo_loss(heads=))
    bf16_grad = os.environ.get("HD_HEAD_GRAD_BF16", "1") != "0"


def head_conv_pitch(cout: int) -> int:
    """Columns per row of the padded head output (the next of 32 / 64 / 128 / 256 >= cout; 0 = unsupported)."""
    return int(_lib.load().hd_head_conv_pitch(int(cout)))


def head_conv_supported(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> bool:
    """bf16 channels-last rows, a dense bf16 [Cout, Cin, 1, 1] weight, Cout <= 256 and a backward the fused kernel takes;
    `bias` (when given): float32, dense, on x's device, one entry per output channel — what hd_head_conv_fwd reads.
    (A forward the streaming kernel does not take — Cin not 32/64/128/256k — runs on the library with padded weights.)"""
    if bias is not None and not (bias.dtype == _F32 and bias.device == x.device and bias.is_contiguous()
                                 and bias.numel() == weight.shape[0]):
        return False
    if not (HeadConvSwitch.enabled and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16
            and weight.dtype == torch.bfloat16 and weight.dim() == 4 and tuple(weight.shape[2:]) == (1, 1)
            and weight.shape[1] == x.shape[1] and weight.is_contiguous() and _cl_row_stride(x, 8) is not None):
        return False
    B, K, H, W = x.shape
    return bool(_lib.load().hd_head_conv_bwd_supported(B * H * W, K, int(weight.shape[0])))


def head_conv_rows(x: torch.Tensor, w2: torch.Tensor, bias: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y [B, pitch, H, W] float32 channels-last = x . w2^T + bias in the first Cout channels, zeros in the rest; no autograd."""
    lib = _lib.load()
    B, K, H, W = x.shape
    N = w2.shape[0]
    P = head_conv_pitch(N)
    ld_x = _cl_row_stride(x, 8)
    if (ld_x is None or x.dtype != torch.bfloat16 or w2.dtype != torch.bfloat16 or tuple(w2.shape) != (N, K) or not w2.is_contiguous()
            or bias.dtype != _F32 or bias.numel() != N or not bias.is_contiguous() or P == 0):
        raise _lib.HdError("head_conv: needs bf16 channels-last rows, a dense bf16 [Cout <= 256, Cin] weight and a float32 bias")
    if out is None:
        out = torch.empty((B, P, H, W), dtype=_F32, device=x.device, memory_format=torch.channels_last)
    ld_y = _cl_row_stride(out, 4)
    if ld_y is None or tuple(out.shape) != (B, P, H, W) or out.dtype != _F32:
        raise _lib.HdError("head_conv: `out` must be a float32 channels-last tensor [B, pitch, H, W]")
    R = B * H * W
    if K <= 256:   # one launch: timed and priced per launch (bench.py); longer reductions run as K / 256 launches, left unpriced
        _note_bytes("head_conv", 2 * R * K + 4 * R * P + 2 * N * K, {"flops": 2 * R * K * P, "shape": (R, K, N)})
    with (_ktimed("head_conv") if K <= 256 else contextlib.nullcontext()):
        check(lib.hd_head_conv_fwd(_ptr(x), int(ld_x), _ptr(w2), _ptr(bias), R, K, N, _ptr(out), int(ld_y), _stream(x)),
              "hd_head_conv_fwd")
    return out


class _HeadConvFn(torch.autograd.Function):
    """Detect convolution: x [B, Cin, H, W] bf16 rows -> y [B, pitch, H, W] FLOAT32 rows (accumulator + bias, unrounded; padding
    channels zero).  backward takes the float32 gradient of y whose padding channels are zero (the loss kernels write them),
    rounds it to bf16 once — what `.float()`'s backward did — and runs hd_head_conv_bwd: dx and the weight-gradient partials
    from one staging.  The bias takes no gradient here: ops.yolo_loss(biases=...) produces it."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        N, K = weight.shape[0], weight.shape[1]
        B, _, H, W = x.shape
        if bool(_lib.load().hd_head_conv_fwd_supported(B * H * W, K, N)):
            y = head_conv_rows(x, weight.view(N, K), bias.detach().contiguous())
        else:  # the library's convolution on the padded weights: the same rows, rounded to bf16 before the bias
            P = head_conv_pitch(N)
            wp = torch.cat([weight, weight.new_zeros((P - N, K, 1, 1))], 0) if P > N else weight
            bp = torch.cat([bias.detach(), bias.new_zeros(P - N)], 0) if P > N else bias.detach()
            xd = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            y = torch.nn.functional.conv2d(xd, wp).float().add_(bp.view(1, P, 1, 1))
            if not y.is_contiguous(memory_format=torch.channels_last):
                y = y.contiguous(memory_format=torch.channels_last)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        dy = gy.to(torch.bfloat16)
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx, dw = head_conv_bwd_rows(dy, x, weight)
        return dx, dw, None


def head_conv_bwd_rows(dy: torch.Tensor, x: torch.Tensor, weight: torch.Tensor):
    """(dx [B, Cin, H, W] bf16, dw like `weight` bf16) of the Detect convolution from dy [B, pitch, H, W] bf16 channels-last
    rows whose padding channels are zero; one launch, the weight gradient's slice sum chained as for every other layer
    (_WgradChain; complete after the next weight-gradient launch or flush_wgrad())."""
    lib = _lib.load()
    N, K = weight.shape[0], weight.shape[1]
    B, _, H, W = x.shape
    R = B * H * W
    ld_dy, ld_x = _cl_row_stride(dy, 8), _cl_row_stride(x, 8)
    if (ld_dy is None or ld_x is None or dy.dtype != torch.bfloat16 or dy.shape[1] != head_conv_pitch(N)
            or tuple(dy.shape[2:]) != (H, W) or dy.shape[0] != B):
        raise _lib.HdError("head_conv backward: the gradient must have the padded bf16 channels-last rows of the output")
    dx = torch.empty((B, K, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    dw = torch.empty((N, K), dtype=torch.bfloat16, device=x.device)
    nbytes = lib.hd_head_conv_bwd_workspace_bytes(R, K, N)
    ws = _wgrad_workspace(x.device, nbytes)
    pend = _WgradChain.piggyback(x.device)
    job = _lib.WgradJob()
    P = dy.shape[1]
    _note_bytes("head_conv_bwd", 2 * (R * (P + 2 * K) + 2 * N * K), {"flops": 4 * R * K * P, "shape": (R, K, N)})
    with _ktimed("head_conv_bwd"):
        check(lib.hd_head_conv_bwd(_ptr(dy), int(ld_dy), _ptr(weight), _ptr(x), int(ld_x), R, K, N, _ptr(dx), K, _ptr(dw),
                                   _ptr(ws), ws.numel(), C.byref(pend[0]) if pend is not None else None, C.byref(job),
                                   _stream(x)), "hd_head_conv_bwd")
    _WgradChain.submit(x.device, job, (ws,), weight)
    return dx, dw.view(weight.shape)


def head_conv(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """Differentiable (x, weight) Detect convolution -> padded float32 rows [B, head_conv_pitch(Cout), H, W]; see _HeadConvFn."""
    return _HeadConvFn.apply(x, weight, bias)


def conv3x3_wgrad_supported(x: torch.Tensor, cout: int, stride: int = 1) -> bool:
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and _cl_row_stride(x, 8) is not None):
        return False
    B, K, H, W = x.shape
    if stride not in (1, 2) or H % stride or W % stride:
        return False
    return bool(_lib.load().hd_conv3x3_wgrad_supported(B, H // stride, W // stride, K, int(cout), int(stride)))


def conv3x3_wgrad_rows(dy: torch.Tensor, x: torch.Tensor, stride: int = 1, out: Optional[torch.Tensor] = None,
                       leaf: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight gradient of a 3x3 / pad 1 convolution of stride 1 or 2: dy [B, Cout, H, W], x [B, Cin, stride H, stride W]
    bf16 channels-last rows -> dw [Cout, Cin, 3, 3] bf16 in channels-last memory ([Cout][ky][kx][Cin]); no autograd."""
    lib = _lib.load()
    B, N, H, W = dy.shape
    K = x.shape[1]
    ld_dy, ld_x = _cl_row_stride(dy, 8), _cl_row_stride(x, 8)
    if (ld_dy is None or ld_x is None or dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16
            or x.shape[0] != B or x.shape[2] != stride * H or x.shape[3] != stride * W):
        raise _lib.HdError("conv3x3_wgrad: needs bf16 channels-last rows, x spatially `stride` times dy")
    nbytes = lib.hd_conv3x3_wgrad_workspace_bytes(B, H, W, K, N, int(stride))
    if nbytes == 0:
        raise _lib.HdError("conv3x3_wgrad: channel counts must be multiples of 32 (<= 1024), stride 1 or 2, and one row block must fit LDS")
    ws = _wgrad_workspace(dy.device, nbytes)   # the other slab of the pair: a pending chained job stays intact
    dw = out if out is not None else torch.empty((N, K, 3, 3), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    if dw.dtype != torch.bfloat16 or tuple(dw.shape) != (N, K, 3, 3) or not dw.is_contiguous(memory_format=torch.channels_last):
        raise _lib.HdError("conv3x3_wgrad: `out` must be a channels-last bf16 [Cout, Cin, 3, 3] tensor")
    pend = _WgradChain.piggyback(dy.device)
    job = _lib.WgradJob()
    _note_bytes("conv3x3_wgrad", 2 * (B * H * W * (stride * stride * K + N) + 9 * N * K),
                {"flops": 18 * B * H * W * K * N, "shape": (B, H, W, K, N, stride)})
    with _ktimed("conv3x3_wgrad"):
        check(lib.hd_conv3x3_wgrad_partial(_ptr(dy), int(ld_dy), _ptr(x), int(ld_x), B, H, W, K, N, int(stride), _ptr(dw), _ptr(ws),
                                           ws.numel(), C.byref(pend[0]) if pend is not None else None, C.byref(job), _stream(dy)),
              "hd_conv3x3_wgrad_partial")
    _WgradChain.submit(dy.device, job, (ws,), leaf)
    return dw


def conv3x3_fwd_supported(x: torch.Tensor, cout: int) -> bool:
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and _cl_row_stride(x, 8) is not None):
        return False
    B, K, H, W = x.shape
    return bool(_lib.load().hd_conv3x3_fwd_supported(B, H, W, K, int(cout)))


def conv3x3_fwd_rows(x: torch.Tensor, w: torch.Tensor, stats: Optional[torch.Tensor] = None, name: str = "conv3x3_fwd") -> torch.Tensor:
    """y = 3x3 / stride 1 / pad 1 convolution of x [B, Cin, H, W] (bf16 channels-last rows) with w [Cout, Cin, 3, 3] (bf16,
    channels-last memory [Cout][3][3][Cin]); stats as for pointwise_conv_rows; no autograd."""
    lib = _lib.load()
    B, K, H, W = x.shape
    N = w.shape[0]
    ld_x = _cl_row_stride(x, 8)
    if (ld_x is None or x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or tuple(w.shape) != (N, K, 3, 3)
            or not w.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.HdError("conv3x3_fwd: needs bf16 channels-last rows and a channels-last bf16 [Cout, Cin, 3, 3] weight")
    y = torch.empty((B, N, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    _note_bytes(name, 2 * (B * H * W * (K + N) + 9 * N * K), {"flops": 18 * B * H * W * K * N, "shape": (B, H, W, K, N, 1)})
    with _ktimed(name):
        check(lib.hd_conv3x3_fwd(_ptr(x), int(ld_x), _ptr(w), B, H, W, K, N, _ptr(y), N, _ptr(stats), _stream(x)), "hd_conv3x3_fwd")
    return y


class FlippedWeights:
    """The flipped, transposed copies hd_conv3x3_fwd needs as a data gradient (wt[c][ky][kx][n] = w[n][2-ky][2-kx][c]), rebuilt
    for every registered weight by ONE launch at the head of a training step (hd_conv3x3_flip_weights) instead of a flip + a
    permuting copy per layer in backward.  One instance per train.Trainer (it owns the buffers for as long as the trainer —
    and any hipGraph captured from it — lives): built from the bf16 weight shadows of the model's 3x3 / stride 1 layers,
    `refresh()` inside the step before forward (so a captured graph replays it), and `with flipped:` around forward +
    backward makes it the one _Conv3x3Fn consults; outside that window the copy is built per call, so a stale one can never
    be used."""

    _current = None

    def __init__(self, weights):
        self.items = []
        self.by_ptr = {}
        for w in weights:
            if not (w.is_cuda and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3) and w.dtype == torch.bfloat16
                    and w.is_contiguous(memory_format=torch.channels_last)):
                continue
            if len(self.items) >= 32:   # HD_MAX_FLIP
                break
            N, K = w.shape[0], w.shape[1]
            wt = torch.empty((K, N, 3, 3), dtype=torch.bfloat16, device=w.device, memory_format=torch.channels_last)
            self.items.append((w, wt))
            self.by_ptr[w.data_ptr()] = wt

    def __len__(self):
        return len(self.items)

    def refresh(self) -> None:
        if not self.items:
            return
        n = len(self.items)
        src = (C.c_void_p * n)(*[w.data_ptr() for w, _ in self.items])
        dst = (C.c_void_p * n)(*[t.data_ptr() for _, t in self.items])
        cout = (C.c_int * n)(*[int(w.shape[0]) for w, _ in self.items])
        cin = (C.c_int * n)(*[int(w.shape[1]) for w, _ in self.items])
        dev = self.items[0][0].device
        check(_lib.load().hd_conv3x3_flip_weights(n, src, dst, cout, cin, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
              "hd_conv3x3_flip_weights")

    def __enter__(self):
        self._prev, FlippedWeights._current = FlippedWeights._current, self
        return self

    def __exit__(self, *exc):
        FlippedWeights._current = self._prev
        return False

    @classmethod
    def get(cls, weight: torch.Tensor):
        cur = cls._current
        return cur.by_ptr.get(weight.data_ptr()) if cur is not None else None


class _Conv3x3Fn(torch.autograd.Function):
    """3x3 / pad 1 convolution of stride 1 or 2.  Stride 1 with a supported shape: forward on hd_conv3x3_fwd (BatchNorm
    statistics into `slot` from its epilogue) and the data gradient on the same kernel with the flipped, transposed weights;
    otherwise those two stay with MIOpen.  The weight gradient always runs on hd_conv3x3_wgrad."""

    @staticmethod
    def forward(ctx, x, weight, stride, slot, ours_fwd):
        ctx.save_for_backward(x, weight)
        ctx.stride = int(stride)
        N = weight.shape[0]
        ctx.ours = (bool(ours_fwd) or slot is not None) and ctx.stride == 1 and conv3x3_fwd_supported(x, N) \
            and weight.is_contiguous(memory_format=torch.channels_last)
        if ctx.ours:
            return conv3x3_fwd_rows(x, weight, _fwd_sums(slot, N) if slot is not None else None)
        if slot is not None:
            raise _lib.HdError("conv3x3: statistics in the epilogue need the hand-written forward (stride 1, supported shape)")
        return torch.nn.functional.conv2d(x, weight, None, ctx.stride, 1)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        st = ctx.stride
        dx = dw = None
        g = dy if (dy.dtype == torch.bfloat16 and _cl_row_stride(dy, 8) is not None) else \
            dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        if ctx.needs_input_grad[0]:
            if ctx.ours and conv3x3_fwd_supported(g, x.shape[1]):
                # dx = conv3x3(dy, wt) with wt[c][ky][kx][n] = w[n][2 - ky][2 - kx][c]
                wt = FlippedWeights.get(weight)   # rebuilt for all layers by one launch at the head of the step
                if wt is None:
                    wt = weight.flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)
                dx = conv3x3_fwd_rows(g, wt, None, name="conv3x3_dgrad")
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, weight, None, (st, st), (1, 1), (1, 1), False, (0, 0), 1,
                                                         (True, False, False))[0]
        if ctx.needs_input_grad[1]:
            dw = conv3x3_wgrad_rows(g, x, st, leaf=weight)
        return dx, dw, None, None, None


def conv3x3(x: torch.Tensor, weight: torch.Tensor, stride: int = 1, slot: Optional["_StatSlot"] = None,
            ours_forward: bool = False) -> torch.Tensor:
    """3x3 / pad 1 convolution whose weight gradient runs on hd_conv3x3_wgrad.  ours_forward (or a BatchNorm `slot` to fill):
    forward and data gradient on hd_conv3x3_fwd too (stride 1, supported shapes), else MIOpen's."""
    return _Conv3x3Fn.apply(x, weight, stride, slot, ours_forward)


def conv3x3_takes_stats(x: torch.Tensor, weight: torch.Tensor, stride: int) -> bool:
    """True when conv3x3(x, weight, stride, slot) runs the hand-written forward (and so can fill a BatchNorm slot)."""
    return (stride == 1 and weight.dtype == torch.bfloat16 and weight.is_contiguous(memory_format=torch.channels_last)
            and conv3x3_fwd_supported(x, weight.shape[0]))


# ------------------------------------------------------------------------------- the 6x6 / stride 2 stem (3 input channels)
def stem_conv_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """x: [B, 3, H, W] float32 or bf16, dense channels-last; weight: [Cout, 3, 6, 6] bf16 in channels-last memory."""
    if not (x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and x.dtype in (torch.float32, torch.bfloat16)
            and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0):
        return False
    if not (weight.dim() == 4 and tuple(weight.shape[1:]) == (3, 6, 6) and weight.dtype == torch.bfloat16
            and weight.is_contiguous(memory_format=torch.channels_last)):
        return False
    B, _, H, W = x.shape
    return bool(_lib.load().hd_stem_conv_supported(B, H, W, int(weight.shape[0]), 0 if x.dtype == torch.float32 else 1))


def stem_conv_rows(x: torch.Tensor, weight: torch.Tensor, stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y [B, Cout, H/2, W/2] bf16 channels-last = Conv2d(3, Cout, 6, 2, 2)(x) on hd_stem_conv_fwd, no autograd; stats as for
    pointwise_conv_rows."""
    lib = _lib.load()
    if not stem_conv_supported(x, weight):
        raise _lib.HdError("stem_conv: needs dense channels-last [B, 3, H, W] images (H even, W % 4 == 0) and a channels-last bf16 "
                           "[Cout in {32, 64}, 3, 6, 6] weight")
    B, _, H, W = x.shape
    N = weight.shape[0]
    y = torch.empty((B, N, H // 2, W // 2), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    _note_bytes("stem_conv_fwd", x.numel() * x.element_size() + y.numel() * 2 + weight.numel() * 2,
                {"flops": 2 * B * (H // 2) * (W // 2) * N * 108})
    with _ktimed("stem_conv_fwd"):
        check(lib.hd_stem_conv_fwd(_ptr(x), 0 if x.dtype == torch.float32 else 1, _ptr(weight), B, H, W, N, _ptr(y), N, _ptr(stats),
                                   _stream(x)), "hd_stem_conv_fwd")
    return y


def stem_conv_wgrad_rows(dy: torch.Tensor, x: torch.Tensor, leaf: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dw [Cout, 3, 6, 6] bf16 (channels-last memory) of the stem from dy [B, Cout, H/2, W/2] bf16 channels-last rows and the
    images x; completes like every chained weight gradient (_WgradChain)."""
    lib = _lib.load()
    B, _, H, W = x.shape
    N = dy.shape[1]
    ld_dy = _cl_row_stride(dy, 8)
    if ld_dy is None or dy.dtype != torch.bfloat16 or tuple(dy.shape) != (B, N, H // 2, W // 2):
        raise _lib.HdError("stem_conv_wgrad: dy must be bf16 channels-last rows [B, Cout, H/2, W/2]")
    nbytes = lib.hd_stem_conv_wgrad_workspace_bytes(B, H, W, N)
    if nbytes == 0:
        raise _lib.HdError("stem_conv_wgrad: unsupported shape")
    _WgradChain.flush_device(dy.device)   # this launch carries no piggyback: sum what is pending first
    ws = _wgrad_workspace(dy.device, nbytes)
    dw = torch.empty((N, 3, 6, 6), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    job = _lib.WgradJob()
    _note_bytes("stem_conv_wgrad", x.numel() * x.element_size() + dy.numel() * 2, {"flops": 2 * B * (H // 2) * (W // 2) * N * 108})
    with _ktimed("stem_conv_wgrad"):
        check(lib.hd_stem_conv_wgrad_partial(_ptr(dy), int(ld_dy), _ptr(x), 0 if x.dtype == torch.float32 else 1, B, H, W, N,
                                             _ptr(dw), _ptr(ws), ws.numel(), C.byref(job), _stream(dy)),
              "hd_stem_conv_wgrad_partial")
    _WgradChain.submit(dy.device, job, (ws,), leaf)  # not dw: a second reference would make autograd copy it
    return dw


class _StemConvFn(torch.autograd.Function):
    """The stem convolution on hd_stem_conv_fwd (BatchNorm statistics into `slot` from its epilogue) / hd_stem_conv_wgrad_partial;
    the images get no gradient."""

    @staticmethod
    def forward(ctx, x, weight, slot):
        ctx.save_for_backward(x, weight)
        return stem_conv_rows(x, weight, _fwd_sums(slot, weight.shape[0]) if slot is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise _lib.HdError("stem_conv: the input images must not require a gradient")
        dw = None
        if ctx.needs_input_grad[1]:
            g = dy if (dy.dtype == torch.bfloat16 and _cl_row_stride(dy, 8) is not None) else \
                dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            dw = stem_conv_wgrad_rows(g, x, weight)
        return None, dw, None


def stem_conv(x: torch.Tensor, weight: torch.Tensor, slot: Optional["_StatSlot"] = None) -> torch.Tensor:
    return _StemConvFn.apply(x, weight, slot)


POINTWISE_MAX_REDUCTION = 256  # hd_pointwise_conv's streaming form (weights resident in LDS); longer reductions -> MIOpen


class _PointwiseConvFn(torch.autograd.Function):
    """1x1 convolution (stride 1, no bias) of channels-last bf16 rows on the MFMA kernel of csrc/pointwise_conv.hip.
    forward: y = x . w^T, and the following fused BatchNorm's batch statistics go into `slot` from the epilogue.
    backward: the data gradient is the same kernel with the transposed weights (when Cout is short enough to be the
    reduction); the weight gradient is hd_pointwise_wgrad (channel counts that are multiples of 64), else MIOpen's."""

    @staticmethod
    def forward(ctx, x, weight, slot):
        N, K = weight.shape[0], weight.shape[1]
        if slot is not None or pointwise_conv_usable(x, weight):
            sums = _fwd_sums(slot, N) if slot is not None else None
            y = pointwise_conv_rows(x, weight.view(N, K), sums)
        else:  # a reduction too long for the resident-weight kernel: MIOpen's forward, our backward
            xd = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            y = torch.nn.functional.conv2d(xd, weight)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        N, K = weight.shape[0], weight.shape[1]
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if dy.dtype != torch.bfloat16 or _cl_row_stride(dy, 8) is None:
            dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dx = dw = None
        R = x.shape[0] * x.shape[2] * x.shape[3]
        ours = need_dx and bool(_lib.load().hd_pointwise_conv_dgrad_supported(R, K, N))
        ours_w = need_dw and pointwise_wgrad_supported(R, K, N) and _cl_row_stride(x, 8) is not None
        if ours and ours_w and pointwise_bwd_supported(R, K, N):   # both gradients from one staging of dy
            dx, dw = pointwise_bwd_rows(dy, weight.view(N, K), x, leaf=weight)
            return dx, dw.view(N, K, 1, 1), None
        if ours:
            dx = pointwise_dgrad_rows(dy, weight.view(N, K))
        if ours_w:
            dw = pointwise_wgrad_rows(dy, x, leaf=weight).view(N, K, 1, 1)
        lib_w, lib_x = need_dw and not ours_w, need_dx and not ours
        if lib_w or lib_x:
            xd = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            dyd = dy if dy.is_contiguous(memory_format=torch.channels_last) else dy.contiguous(memory_format=torch.channels_last)
            gi, gw, _ = torch.ops.aten.convolution_backward(dyd, xd, weight, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                            (lib_x, lib_w, False))
            if lib_w:
                dw = gw
            if lib_x:
                dx = gi
        return dx, dw, None


def pointwise_conv(x: torch.Tensor, weight: torch.Tensor, slot: Optional["_StatSlot"] = None) -> torch.Tensor:
    """Differentiable 1x1 convolution; with `slot` (a bn_act_slot of Cout channels) the BatchNorm statistics of the
    result are accumulated there, and the bn_act that follows is called with stats_done=True."""
    return _PointwiseConvFn.apply(x, weight, slot)


def pointwise_conv_usable(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """The forward kernel is preferred to the library where it runs in its streaming form (Cin <= 256, <= 8 column tiles)."""
    if not pointwise_conv_supported(x, weight):
        return False
    return bool(_lib.load().hd_pointwise_conv_resident(x.shape[0] * x.shape[2] * x.shape[3], weight.shape[1], weight.shape[0]))


def pointwise_backward_usable(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """The backward kernels alone are worth taking (weight gradient without zero-fill / cast launches) even where the
    forward stays with the library: channel counts that are multiples of 64, bf16 channels-last rows."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.dim() == 4
            and weight.shape[2:] == (1, 1) and weight.shape[1] == x.shape[1] and weight.is_contiguous()):
        return False
    B, K, H, W = x.shape
    return pointwise_wgrad_supported(B * H * W, K, weight.shape[0]) and _cl_row_stride(x, 8) is not None


class _BNState:
    """What the fused BatchNorm + activation forward leaves for its backward (shared by _BNActFn and _ConvBNActFn)."""
    __slots__ = ("dt", "R", "C", "act", "slot", "sync_group", "stat_rows", "res_mode", "split", "tag", "w", "b", "save", "res")


def _bn_forward(x, weight, bias, running_mean, running_var, eps, momentum, act, slot, sync_group, res, res_mode, out_slot,
                stats_done):
    """Statistics (unless the producer of x already left them in the slot) + apply.  -> (y, y2 | None, _BNState)."""
    lib = _lib.load()
    if res is not None:
        if res.shape != x.shape or res.dtype != x.dtype or res.stride() != x.stride():
            res = res.to(x.dtype).contiguous(memory_format=torch.channels_last)
            if res.shape != x.shape:
                raise _lib.HdError("bn_act: the residual must have the shape of the normalised tensor")
    B, Cc, H, W = x.shape
    R = B * H * W
    dt = 0 if x.dtype == torch.float32 else 1
    sums = slot.buf[: lib.hd_bn_act_replicas() * 2 * Cc] if stats_done else _fwd_sums(slot, Cc)
    vec = 4 if dt == 0 else 8
    split = 0
    y2, tail = None, None
    if out_slot is not None and isinstance(out_slot.t, tuple):
        # two destinations (merged C3 branch convolutions): channels [0, split) -> y, [split, C) -> y2
        y, y2 = out_slot.t
        split = y.shape[1]
        ld_y, ld_y2 = _cl_row_stride(y, vec), _cl_row_stride(y2, vec)
        if (ld_y is None or ld_y2 is None or y.shape[1] + y2.shape[1] != Cc or y.dtype != x.dtype or y2.dtype != x.dtype
                or split % vec or y.shape[0] != B or y.shape[2:] != x.shape[2:] or y2.shape[2:] != x.shape[2:]):
            raise _lib.HdError("bn_act: split outputs must be channels-last tensors / channel slices covering x's channels")
        tail = Rows2(y2.data_ptr(), int(ld_y2), int(split))
    else:
        y = out_slot.t if out_slot is not None else torch.empty_like(x)
        ld_y = _cl_row_stride(y, vec)
        if ld_y is None or y.shape != x.shape or y.dtype != x.dtype:
            raise _lib.HdError("bn_act: `out` must be a channels-last tensor (or channel slice) of x's shape and dtype")
    save = torch.empty((2, Cc), dtype=_F32, device=x.device)
    st = _stream(x)
    w, b = weight.detach().float(), bias.detach().float()
    tb = x.numel() * x.element_size()
    tag = (tuple(x.shape), int(act), int(res_mode) if res is not None else 0, bool(stats_done))
    if not stats_done:
        _note_bytes("bn_stats_kernel", tb, tag)
        with _ktimed("bn_stats_kernel"):
            check(lib.hd_bn_act_stats(_ptr(x), dt, R, Cc, _ptr(sums), st), "hd_bn_act_stats")
    stat_rows = 0
    if sync_group is not None:
        # synchronised BatchNorm: the per-channel (sum, sum of squares) of every rank are added between the
        # statistics kernel and the apply kernel; ranks hold equal shards, so the sums cover R * world rows
        import torch.distributed as dist
        dist.all_reduce(sums, group=sync_group)
        stat_rows = R * dist.get_world_size(sync_group)
    _note_bytes("bn_apply_kernel", (3 if res is not None else 2) * tb, tag)
    with _ktimed("bn_apply_kernel"):
        check(lib.hd_bn_act_apply(_ptr(x), dt, R, stat_rows, Cc, _ptr(sums), _ptr(w), _ptr(b), float(eps), float(momentum),
                                  _ptr(running_mean), _ptr(running_var), _ptr(save[0]), _ptr(save[1]), int(act),
                                  _ptr(res), int(res_mode) if res is not None else 0, _ptr(y), int(ld_y),
                                  C.byref(tail) if tail is not None else None, st), "hd_bn_act_apply")
    s_ = _BNState()
    s_.dt, s_.R, s_.C, s_.act, s_.slot, s_.sync_group, s_.stat_rows = dt, R, Cc, int(act), slot, sync_group, stat_rows
    s_.res_mode, s_.split, s_.tag = (0 if res is None else int(res_mode)), split, tag
    # the backward kernels need the residual only when it was added BEFORE the activation (mode 1)
    s_.w, s_.b, s_.save, s_.res = w, b, save, (res if (res is not None and int(res_mode) == 1) else None)
    return y, y2, s_


class _BNGrad:
    """The incoming gradient of a fused BatchNorm + activation as the kernels take it: (a channel slice of) a channels-last
    tensor, possibly in two places (the merged C3 branch convolutions)."""
    __slots__ = ("dy", "ld", "dy2", "tail", "dsums")


def _bn_backward_reduce(st: "_BNState", x, dy, dy2) -> "_BNGrad":
    """Per-channel sums of the backward (bn_bwd_reduce_kernel) into the slot's accumulators."""
    lib = _lib.load()
    vec = 4 if st.dt == 0 else 8
    Cc = st.C

    def prep(g, channels):
        """-> (tensor kept alive, row stride): a (channel slice of a) channels-last tensor, copied once otherwise."""
        if g is None:  # an unused half: zero gradient
            g = torch.zeros((x.shape[0], channels, x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device,
                            memory_format=torch.channels_last)
        if g.dtype != x.dtype:
            g = g.to(x.dtype)
        ld_ = _cl_row_stride(g, vec)
        if ld_ is None:
            g = g.contiguous(memory_format=torch.channels_last)
            ld_ = channels
        return g, ld_

    g_ = _BNGrad()
    g_.tail, g_.dy2 = None, None
    if st.split:
        g_.dy, g_.ld = prep(dy, st.split)
        g_.dy2, ld2 = prep(dy2, Cc - st.split)
        g_.tail = Rows2(g_.dy2.data_ptr(), int(ld2), int(st.split))
    else:
        g_.dy, g_.ld = prep(dy, Cc)
    tail_ref = C.byref(g_.tail) if g_.tail is not None else None
    rep = lib.hd_bn_act_replicas()
    slot = st.slot
    dsums = slot.buf[rep * 2 * Cc: rep * 4 * Cc]
    if slot.bwd_gen == slot.arena.gen:
        dsums.zero_()
    slot.bwd_gen = slot.arena.gen
    g_.dsums = dsums
    tb = x.numel() * x.element_size()
    _note_bytes("bn_bwd_reduce_kernel", (3 if st.res is not None else 2) * tb, st.tag)  # x, dy (+ the pre-activation residual)
    with _ktimed("bn_bwd_reduce_kernel"):
        check(lib.hd_bn_act_bwd_reduce(_ptr(x), _ptr(g_.dy), int(g_.ld), st.dt, st.R, Cc, _ptr(st.save[0]), _ptr(st.save[1]),
                                       _ptr(st.w), _ptr(st.b), st.act, _ptr(st.res), _ptr(dsums), tail_ref, _stream(x)),
              "hd_bn_act_bwd_reduce")
    return g_


def _bn_backward_dx(st: "_BNState", x, g_: "_BNGrad"):
    """The elementwise half (bn_bwd_dx_kernel): -> (dx, dgamma, dbeta, d(residual) | None)."""
    lib = _lib.load()
    Cc = st.C
    rep = lib.hd_bn_act_replicas()
    local = None
    if st.sync_group is not None:
        # dgamma / dbeta stay this rank's own sums (the gradient all-reduce averages them like every other
        # parameter gradient); dx needs the sums over the whole synchronised batch
        import torch.distributed as dist
        local = g_.dsums.view(rep, 2, Cc).sum(0)
        dist.all_reduce(g_.dsums, group=st.sync_group)
    dx = torch.empty_like(x)
    dg = torch.empty((2, Cc), dtype=_F32, device=x.device)
    dres = torch.empty_like(x) if st.res is not None else None
    tb = x.numel() * x.element_size()
    _note_bytes("bn_bwd_dx_kernel", (5 if st.res is not None else 3) * tb, st.tag)  # x, dy -> dx (+ residual in, d(res) out)
    with _ktimed("bn_bwd_dx_kernel"):
        check(lib.hd_bn_act_bwd_dx(_ptr(x), _ptr(g_.dy), int(g_.ld), st.dt, st.R, st.stat_rows, Cc, _ptr(st.save[0]),
                                   _ptr(st.save[1]), _ptr(st.w), _ptr(st.b), st.act, _ptr(st.res), _ptr(g_.dsums), _ptr(dx),
                                   _ptr(dres), _ptr(dg[0]), _ptr(dg[1]), C.byref(g_.tail) if g_.tail is not None else None,
                                   _stream(x)), "hd_bn_act_bwd_dx")
    # d(residual): dz (written by the dx kernel) in mode 1, the incoming gradient itself in mode 2
    gres = dres if st.res_mode == 1 else (g_.dy if st.res_mode == 2 else None)
    if local is not None:
        return dx, local[1], local[0], gres
    return dx, dg[0], dg[1], gres


class _BNActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, act, slot, sync_group=None, res=None,
                res_mode=0, out_slot=None, stats_done=False):
        """stats_done: the producer of x (pointwise_conv) already added x's per-channel sums into the slot."""
        y, y2, st = _bn_forward(x, weight, bias, running_mean, running_var, eps, momentum, act, slot, sync_group, res, res_mode,
                                out_slot, stats_done)
        ctx.save_for_backward(x)
        ctx.st = st
        return y if y2 is None else (y, y2)

    @staticmethod
    def backward(ctx, dy, dy2=None):
        (x,) = ctx.saved_tensors
        st = ctx.st
        g_ = _bn_backward_reduce(st, x, dy, dy2)
        dx, dgamma, dbeta, gres = _bn_backward_dx(st, x, g_)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, gres, None, None, None


class _PwConvBNActFn(torch.autograd.Function):
    """1x1 convolution -> BatchNorm (batch statistics) -> activation [-> + residual] as ONE autograd node: its backward runs
    bn_bwd_reduce, bn_bwd_dx and then the convolution's data + weight gradient in one launch (hd_pointwise_bwd), with a
    shortcut's gradient added in that kernel's epilogue (`fork`).  (Folding bn_bwd_dx into the gradient kernels' operand load
    was built in round 3 — bit-identical, slower — and removed in round 5: tools/experiments/r05_removed_paths.patch.)"""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, eps, momentum, act, slot, res, res_mode, out_slot, ours_fwd,
                fork):
        """fork: also return an alias of x (last output).  A shortcut that takes THAT tensor instead of x sends its gradient
        through this node, whose data-gradient kernel adds it in its epilogue (hd_pointwise_conv_dgrad_add) — otherwise the
        autograd engine sums the two gradients of x with a launch of its own."""
        N, K = weight.shape[0], weight.shape[1]
        if ours_fwd:   # the epilogue leaves the batch statistics in the slot
            y_raw = pointwise_conv_rows(x, weight.view(N, K), _fwd_sums(slot, N))
        else:          # a reduction too long for the resident-weight kernel: the library's forward
            xd = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            y_raw = torch.nn.functional.conv2d(xd, weight)
        y, y2, st = _bn_forward(y_raw, gamma, beta, running_mean, running_var, eps, momentum, act, slot, None, res, res_mode,
                                out_slot, bool(ours_fwd))
        ctx.save_for_backward(x, weight, y_raw)
        ctx.st = st
        ctx.fork = bool(fork)
        outs = (y,) if y2 is None else (y, y2)
        if fork:
            outs = outs + (x.view_as(x),)
        return outs[0] if len(outs) == 1 else outs

    @staticmethod
    def backward(ctx, *grads):
        x, weight, y_raw = ctx.saved_tensors
        st = ctx.st
        d_skip = None
        if ctx.fork:
            grads, d_skip = grads[:-1], grads[-1]
        dout, dout2 = grads[0], (grads[1] if len(grads) > 1 else None)
        if d_skip is not None and (d_skip.dtype != torch.bfloat16 or _cl_row_stride(d_skip, 8) is None or d_skip.shape != x.shape):
            d_skip = d_skip.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        N, K = weight.shape[0], weight.shape[1]
        R = x.shape[0] * x.shape[2] * x.shape[3]
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        lib = _lib.load()
        g_ = _bn_backward_reduce(st, y_raw, dout, dout2)
        ours_dx = bool(lib.hd_pointwise_conv_dgrad_supported(R, K, N))
        ours_dw = pointwise_wgrad_supported(R, K, N) and _cl_row_stride(x, 8) is not None
        dy, dgamma, dbeta, gres = _bn_backward_dx(st, y_raw, g_)
        dx = dw = None
        if need_dw and ours_dw and need_dx and ours_dx and pointwise_bwd_supported(R, K, N):
            # one launch: data gradient (+ the shortcut's gradient in its epilogue) and the weight-gradient partials
            dx, dw = pointwise_bwd_rows(dy, weight.view(N, K), x, add=d_skip, leaf=weight)
            dw = dw.view(N, K, 1, 1)
            d_skip = None
        if need_dw and ours_dw and dw is None:
            dw = pointwise_wgrad_rows(dy, x, leaf=weight).view(N, K, 1, 1)
        if need_dx and ours_dx and dx is None:
            dx = pointwise_dgrad_rows(dy, weight.view(N, K), add=d_skip)   # the shortcut's gradient rides in the epilogue
            d_skip = None
        lib_w, lib_x = need_dw and dw is None, need_dx and dx is None
        if lib_w or lib_x:
            xd = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            gi, gw, _ = torch.ops.aten.convolution_backward(dy, xd, weight, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                            (lib_x, lib_w, False))
            dw = gw if lib_w else dw
            dx = gi if lib_x else dx
        if dx is not None and d_skip is not None:
            dx = dx + d_skip
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None, gres, None, None, None, None


def pointwise_conv_bn_act(x, weight, gamma, beta, running_mean, running_var, eps: float, momentum: float, act: str,
                          slot: "_StatSlot", residual=None, residual_after_act: bool = False, out=None, ours_forward: bool = True,
                          fork: bool = False):
    """conv1x1(x, weight) -> BatchNorm2d (training) -> activation [+ residual] on the HIP kernels, one autograd node (see
    _PwConvBNActFn).  x: bf16 channels-last rows, weight: bf16 [Cout, Cin, 1, 1]; out as for bn_act.
    fork: returns (result, x_alias) — hand x_alias to the shortcut that bypasses this layer."""
    return _PwConvBNActFn.apply(x, weight, gamma, beta, running_mean, running_var, eps, momentum, _ACT_CODE[act], slot, residual,
                                (2 if residual_after_act else 1) if residual is not None else 0,
                                _OutSlot(out) if out is not None else None, bool(ours_forward), bool(fork))


class _BiasActFn(torch.autograd.Function):
    """y = act(x + bias[c]) for a biased convolution that was run without its bias (csrc/bn_act.hip: hd_bias_act_fwd / _bwd).
    Backward works from the output: dx = dy * (y > 0) for ReLU (dy itself for no activation) and the bias gradient is the
    per-channel sum of dx taken in the same pass."""

    @staticmethod
    def forward(ctx, x, bias, code: int, inplace: bool):
        lib = _lib.load()
        B, Cc, H, W = x.shape
        y = x if inplace else torch.empty_like(x, memory_format=torch.channels_last)
        b32 = bias if bias.dtype == torch.float32 else bias.float()
        check(lib.hd_bias_act_fwd(_ptr(x), 0 if x.dtype == torch.float32 else 1, B * H * W, Cc, _ptr(b32), code, _ptr(y),
                                  _stream(x)), "hd_bias_act_fwd")
        if inplace:
            ctx.mark_dirty(x)
        ctx.code, ctx.bias_dtype = code, bias.dtype
        ctx.save_for_backward(y if code != 0 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        B, Cc, H, W = dy.shape
        rep = lib.hd_bn_act_replicas()
        dsums = torch.zeros(rep * 2 * Cc, dtype=torch.float32, device=dy.device)
        dx = torch.empty_like(dy, memory_format=torch.channels_last) if ctx.code != 0 else None
        check(lib.hd_bias_act_bwd(_ptr(y), _ptr(dy), 0 if dy.dtype == torch.float32 else 1, B * H * W, Cc, ctx.code, _ptr(dx),
                                  _ptr(dsums), _stream(dy)), "hd_bias_act_bwd")
        db = dsums.view(rep, 2, Cc)[:, 0].sum(0)
        return (dx if dx is not None else dy), (db if ctx.bias_dtype == torch.float32 else db.to(ctx.bias_dtype)), None, None


def bias_act_supported(x: torch.Tensor, act) -> bool:
    return act in (None, "none", "relu") and bn_act_supported(x)


def bias_act(x: torch.Tensor, bias: torch.Tensor, act=None, inplace: bool = False) -> torch.Tensor:
    """act(x + bias[None, :, None, None]) on a dense channels-last float32 / bfloat16 HIP tensor, act None or "relu"; bias float32
    (or x's dtype).  inplace: write over x (x must not be needed by the backward of whatever produced it — a convolution's
    output is not)."""
    if not bias_act_supported(x, act):
        raise _lib.HdError("bias_act: needs a dense channels-last float32/bfloat16 HIP tensor with C % vec == 0, act None or relu")
    if bias.dim() != 1 or bias.shape[0] != x.shape[1] or not bias.is_cuda:
        raise _lib.HdError("bias_act: bias must be a HIP vector of C elements")
    return _BiasActFn.apply(x, bias, _ACT_CODE[act], bool(inplace))


def bn_act_slot(device, channels: int) -> _StatSlot:
    # forward sums + backward sums, each [replicas, 2, C]
    return _arena(torch.device(device)).alloc(4 * channels * _lib.load().hd_bn_act_replicas())


def bn_act(x, weight, bias, running_mean, running_var, eps: float, momentum: float, act: str, training: bool,
           slot: Optional[_StatSlot] = None, sync_group=None, residual: Optional[torch.Tensor] = None,
           residual_after_act: bool = False, out: Optional[torch.Tensor] = None, stats_done: bool = False) -> torch.Tensor:
    """Fused BatchNorm2d + activation on a dense channels-last HIP tensor (see csrc/bn_act.hip).
    sync_group: a torch.distributed process group -> statistics (and the backward sums) are taken over the
    batches of all its ranks (SyncBN; ranks must hold equal shards).  Training mode only, not graph-capturable.
    residual: a tensor shaped like x that is added inside the apply kernel — before the activation
    (`act(bn(x) + r)`, the ResNet bottleneck tail) or after it (`act(bn(x)) + r`, the YOLOv5 shortcut).
    out: a tensor to write the result into, e.g. a CatBuffer slot (a channel slice of a channels-last buffer); or a
    pair (y_first, y_second) of such tensors that split the channels between them (returns the pair)."""
    if not bn_act_supported(x):
        raise _lib.HdError("bn_act: needs a dense channels-last float32/bfloat16 HIP tensor with C % vec == 0")
    code = _ACT_CODE[act]
    if training:
        if slot is None:
            slot = bn_act_slot(x.device, x.shape[1])
        return _BNActFn.apply(x, weight, bias, running_mean, running_var, eps, momentum, code, slot, sync_group, residual,
                              (2 if residual_after_act else 1) if residual is not None else 0,
                              _OutSlot(out) if out is not None else None, bool(stats_done))
    lib = _lib.load()
    B, Cc, H, W = x.shape
    vec = 4 if x.dtype == torch.float32 else 8
    tail, y2 = None, None
    if isinstance(out, tuple):  # two destinations (merged C3 branch convolutions); no residual in that layer
        if residual is not None:
            raise _lib.HdError("bn_act: split outputs and a residual are not combined")
        y, y2 = out
        ld_y, ld_y2 = _cl_row_stride(y, vec), _cl_row_stride(y2, vec)
        if ld_y is None or ld_y2 is None or y.shape[1] + y2.shape[1] != Cc or y.shape[1] % vec:
            raise _lib.HdError("bn_act: split outputs must be channels-last tensors / channel slices covering x's channels")
        tail = Rows2(y2.data_ptr(), int(ld_y2), int(y.shape[1]))
    else:
        direct = out is not None and residual is None  # inference with a residual finishes with torch ops below
        y = out if direct else torch.empty_like(x)
        ld_y = _cl_row_stride(y, vec)
        if ld_y is None:
            raise _lib.HdError("bn_act: `out` must be a channels-last tensor or channel slice")
    fused_act = code if (residual is None or residual_after_act) else 0
    check(lib.hd_bn_act_infer(_ptr(x), 0 if x.dtype == torch.float32 else 1, B * H * W, Cc, _ptr(weight.detach().float()),
                              _ptr(bias.detach().float()), _ptr(running_mean), _ptr(running_var), float(eps), fused_act,
                              _ptr(y), int(ld_y), C.byref(tail) if tail is not None else None, _stream(x)), "hd_bn_act_infer")
    if y2 is not None:
        return y, y2
    if residual is not None:  # inference: the add (and a pre-add activation) stay plain torch ops
        y = y + residual
        if not residual_after_act:
            y = {1: torch.nn.functional.silu, 2: torch.relu, 0: (lambda t: t)}[code](y)
        if out is not None:
            out.copy_(y)
            y = out
    return y


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


def time_bn_act_kernels(shape, dtype=torch.bfloat16, act: str = "silu", iters: int = 20, device=None,
                        residual_after_act: bool = False):
    """Measurement helper for bench.py: the four fused-BN kernels on one channels-last tensor of `shape`
    ([B,C,H,W]), each launched `iters` times on the current stream, every launch stamping its own HIP event pair
    with the kernel's begin / end (hd_profile_time_next_launch), i.e. the kernel alone, without host launch gaps.
    Returns {kernel: (median ms, algorithmic bytes)}.
    residual_after_act: the apply kernel also reads a residual tensor (y = act(bn(x)) + r, the YOLOv5 shortcut;
    the backward kernels are the plain ones in that mode)."""
    lib = _lib.load()
    device = torch.device(device or "cuda")
    x = torch.randn(shape, device=device).to(dtype).contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x)
    B, Cc, H, W = shape
    R = B * H * W
    dt = 0 if dtype == torch.float32 else 1
    rep = lib.hd_bn_act_replicas()
    sums = torch.zeros(rep * 4 * Cc, dtype=_F32, device=device)
    dsums = sums[rep * 2 * Cc:]
    w = torch.ones(Cc, dtype=_F32, device=device)
    b = torch.zeros(Cc, dtype=_F32, device=device)
    save = torch.zeros((2, Cc), dtype=_F32, device=device)
    y = torch.empty_like(x)
    res = torch.randn_like(x) if residual_after_act else None
    dg = torch.empty((2, Cc), dtype=_F32, device=device)
    st = _stream(x)
    code = _ACT_CODE[act]
    tb = x.numel() * x.element_size()

    def run(fn):
        for _ in range(3):
            fn()
        evs = []
        for _ in range(iters):
            evs.append(_arm_kernel_timing())  # the launch stamps the pair with the kernel's own begin / end
            fn()
        torch.cuda.synchronize(device)
        ts = sorted(a.elapsed_time(e) for a, e in evs)
        return ts[len(ts) // 2]

    out = {}
    out["bn_stats_kernel"] = (run(lambda: check(lib.hd_bn_act_stats(_ptr(x), dt, R, Cc, _ptr(sums), st), "stats")), tb)
    sums.zero_()
    check(lib.hd_bn_act_stats(_ptr(x), dt, R, Cc, _ptr(sums), st), "stats")
    out["bn_apply_kernel"] = (run(lambda: check(lib.hd_bn_act_apply(
        _ptr(x), dt, R, 0, Cc, _ptr(sums), _ptr(w), _ptr(b), 1e-3, 0.03, None, None, _ptr(save[0]), _ptr(save[1]), code,
        _ptr(res), 2 if res is not None else 0, _ptr(y), 0, None, st), "apply")), (3 if res is not None else 2) * tb)
    out["bn_bwd_reduce_kernel"] = (run(lambda: check(lib.hd_bn_act_bwd_reduce(
        _ptr(x), _ptr(dy), 0, dt, R, Cc, _ptr(save[0]), _ptr(save[1]), _ptr(w), _ptr(b), code, None, _ptr(dsums), None, st), "reduce")), 2 * tb)
    out["bn_bwd_dx_kernel"] = (run(lambda: check(lib.hd_bn_act_bwd_dx(
        _ptr(x), _ptr(dy), 0, dt, R, 0, Cc, _ptr(save[0]), _ptr(save[1]), _ptr(w), _ptr(b), code, None, _ptr(dsums), _ptr(y),
        None, _ptr(dg[0]), _ptr(dg[1]), None, st), "dx")), 3 * tb)
    return out


# ------------------------------------------------------------------------------- hipGraph hygiene
def graph_replace_memsets(raw_graph) -> Tuple[int, int]:
    """raw_graph: the hipGraph_t of a captured but not yet instantiated graph
    (``torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()``).  Returns (nodes, memsets replaced)."""
    lib = _lib.load()
    n, r = C.c_int(0), C.c_int(0)
    ptr = raw_graph if isinstance(raw_graph, int) else int(raw_graph)
    check(lib.hd_graph_replace_memsets(C.c_void_p(ptr), C.byref(n), C.byref(r)), "hd_graph_replace_memsets")
    return n.value, r.value


def graph_mark(slot: torch.Tensor, index: int) -> None:
    """Marks the current point of a stream capture (one-lane kernel storing `index` into the int32 tensor `slot`)."""
    check(_lib.load().hd_graph_mark(_ptr(slot), int(index), _stream(slot)), "hd_graph_mark")


class GraphSegments:
    """Executable segments of a captured graph cut at its markers (hd_graph_split_instantiate)."""

    def __init__(self, raw_graph, slot: torch.Tensor, max_segments: int):
        lib = _lib.load()
        execs = (C.c_void_p * max_segments)()
        marks = (C.c_int * max_segments)()
        n, nn = C.c_int(0), C.c_int(0)
        ptr = raw_graph if isinstance(raw_graph, int) else int(raw_graph)
        check(lib.hd_graph_split_instantiate(C.c_void_p(ptr), _ptr(slot), int(max_segments), execs, marks, C.byref(n),
                                             C.byref(nn)), "hd_graph_split_instantiate")
        self.execs = [execs[i] for i in range(n.value)]
        self.nodes = nn.value
        self._slot = slot  # keeps the marker slot alive for the replays

        self.marker_of = [marks[i] for i in range(n.value)]  # entry k valid for the marker segments only

    def launch(self, k: int, like: torch.Tensor) -> None:
        check(_lib.load().hd_graph_exec_launch(C.c_void_p(self.execs[k]), _stream(like)), "hd_graph_exec_launch")

    def __len__(self):
        return len(self.execs)

    def __del__(self):
        try:
            lib = _lib.load()
            for e in self.execs:
                lib.hd_graph_exec_destroy(C.c_void_p(e))
        except Exception:
            pass
