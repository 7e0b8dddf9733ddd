"""ctypes front-end of oracle/hd_oracle.c.  TEST INFRASTRUCTURE ONLY (parity unpinned).

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product package ``heltondetection_amd`` never does.  All arguments and
results are numpy arrays on the host.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhd_oracle.so")
MAX_LEVELS = 8
MAX_ANCHORS = 16


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "hd_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libhd_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


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
    ]


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


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------- anchors
def anchor_base(sizes, ratios):
    sizes, ratios = _f(sizes), _f(ratios)
    out = np.empty((len(ratios) * len(sizes), 4), np.float32)
    lib().orc_anchor_base(_p(sizes), C.c_int(len(sizes)), _p(ratios), C.c_int(len(ratios)), _p(out))
    return out


def anchor_grid(feat_h, feat_w, strides, base_anchors, offset=0.5):
    fh, fw, st = _i(feat_h), _i(feat_w), _f(strides)
    base = _f(base_anchors)  # [L, A, 4]
    L, A = base.shape[0], base.shape[1]
    n = int(sum(int(h) * int(w) for h, w in zip(fh, fw))) * A
    out = np.empty((n, 4), np.float32)
    lib().orc_anchor_grid(C.c_int(L), _p(fh), _p(fw), _p(st), _p(base), C.c_int(A), C.c_float(offset), _p(out))
    return out


# ---------------------------------------------------------------------------- IoU
def pairwise_iou(a, b, m_valid=None):
    a, b = _f(a), _f(b)
    B, N, M = a.shape[0], a.shape[1], b.shape[1]
    mv = None if m_valid is None else _i(m_valid)
    out = np.empty((B, N, M), np.float32)
    lib().orc_pairwise_iou(_p(a), _p(b), _p(mv), C.c_int(B), C.c_int(N), C.c_int(M), _p(out))
    return out


def iou_assign(a, b, m_valid=None, with_gt=True):
    a, b = _f(a), _f(b)
    B, N, M = a.shape[0], a.shape[1], b.shape[1]
    mv = None if m_valid is None else _i(m_valid)
    mx, am = np.empty((B, N), np.float32), np.empty((B, N), np.int32)
    gmx = np.empty((B, M), np.float32) if with_gt else None
    gam = np.empty((B, M), np.int32) if with_gt else None
    lib().orc_iou_assign(_p(a), _p(b), _p(mv), C.c_int(B), C.c_int(N), C.c_int(M), _p(mx), _p(am), _p(gmx), _p(gam))
    return mx, am, gmx, gam


# ---------------------------------------------------------------------------- codec
def box_encode(src, dst):
    src, dst = _f(src), _f(dst)
    out = np.empty_like(src)
    lib().orc_box_encode(_p(src), _p(dst), C.c_int64(src.shape[0]), _p(out))
    return out


def box_decode(anchors, deltas, img_h=0.0, img_w=0.0):
    anchors, deltas = _f(anchors), _f(deltas)
    out = np.empty_like(anchors)
    lib().orc_box_decode(_p(anchors), _p(deltas), C.c_int64(anchors.shape[0]), C.c_float(img_h), C.c_float(img_w), _p(out))
    return out


# ---------------------------------------------------------------------------- NMS
def nms_batched(boxes, scores, cls=None, n_valid=None, score_thr=-np.inf, iou_thr=0.5,
                max_candidates=0, max_out=None):
    boxes, scores = _f(boxes), _f(scores)
    B, N = scores.shape
    max_out = N if max_out is None else max_out
    c = None if cls is None else _i(cls)
    nv = None if n_valid is None else _i(n_valid)
    keep = np.empty((B, max_out), np.int32)
    cnt = np.empty((B,), np.int32)
    lib().orc_nms_batched(_p(boxes), _p(scores), _p(c), _p(nv), C.c_int(B), C.c_int(N), C.c_float(score_thr),
                          C.c_float(iou_thr), C.c_int(max_candidates), C.c_int(max_out), _p(keep), _p(cnt))
    return keep, cnt


def rpn_proposals(logits, deltas, anchors, level_sizes, pre_nms_topk, post_nms_topk, iou_thr,
                  min_size, img_h, img_w, return_index=False):
    logits, deltas, anchors = _f(logits), _f(deltas), _f(anchors)
    B, NA = logits.shape
    ls = np.ascontiguousarray(level_sizes, dtype=np.int32)
    rois = np.empty((B, post_nms_topk, 4), np.float32)
    sc = np.empty((B, post_nms_topk), np.float32)
    cnt = np.empty((B,), np.int32)
    idx = np.empty((B, post_nms_topk), np.int32)
    lib().orc_rpn_proposals(_p(logits), _p(deltas), _p(anchors), C.c_int(B), C.c_int(NA), C.c_int(len(ls)), _p(ls),
                            C.c_int(pre_nms_topk), C.c_int(post_nms_topk), C.c_float(iou_thr), C.c_float(min_size),
                            C.c_float(img_h), C.c_float(img_w), _p(rois), _p(sc), _p(idx), _p(cnt))
    return (rois, sc, cnt, idx) if return_index else (rois, sc, cnt)


# ---------------------------------------------------------------------------- RoIAlign
def _roi_levels(feats, scales, k_min, canonical_scale, canonical_level, layout):
    """feats: list of [B,C,H,W] float32 arrays (logical NCHW; `layout` selects memory order)."""
    st = RoiLevels()
    st.n_levels = len(feats)
    keep = []
    for l, f in enumerate(feats):
        B, Cc, H, W = f.shape
        if layout == "nhwc":
            mem = np.ascontiguousarray(f.transpose(0, 2, 3, 1))
            sb, sh, sw, sc = H * W * Cc, W * Cc, Cc, 1
        else:
            mem = np.ascontiguousarray(f)
            sb, sc, sh, sw = Cc * H * W, H * W, W, 1
        keep.append(mem)
        st.feat[l] = mem.ctypes.data
        st.h[l], st.w[l] = H, W
        st.spatial_scale[l] = scales[l]
        st.sb[l], st.sc[l], st.sh[l], st.sw[l] = sb, sc, sh, sw
    st.k_min = k_min
    st.canonical_scale = canonical_scale
    st.canonical_level = canonical_level
    return st, keep


def roi_level_map(rois, k_min, n_levels, canonical_scale=224.0, canonical_level=4):
    rois = _f(rois)
    out = np.empty((rois.shape[0],), np.int32)
    lib().orc_roi_level_map(_p(rois), C.c_int(rois.shape[0]), C.c_int(k_min), C.c_int(n_levels),
                            C.c_float(canonical_scale), C.c_int(canonical_level), _p(out))
    return out


def roi_align_fwd(feats, rois, P, scales, sampling_ratio=2, aligned=True, k_min=2,
                  canonical_scale=224.0, canonical_level=4, layout="nchw"):
    feats = [_f(f) for f in feats]
    rois = _f(rois)
    st, keep = _roi_levels(feats, scales, k_min, canonical_scale, canonical_level, layout)
    R, Cc = rois.shape[0], feats[0].shape[1]
    out = np.zeros((R, Cc, P, P), np.float32)
    lib().orc_roi_align_fwd(C.byref(st), _p(rois), C.c_int(R), C.c_int(Cc), C.c_int(P), C.c_int(sampling_ratio),
                            C.c_int(int(aligned)), _p(out), C.c_int64(Cc * P * P), C.c_int64(P * P), C.c_int64(P), C.c_int64(1))
    del keep
    return out


def roi_align_bwd(feat_shapes, rois, grad_out, scales, sampling_ratio=2, aligned=True, k_min=2,
                  canonical_scale=224.0, canonical_level=4):
    grads = [np.zeros(s, np.float32) for s in feat_shapes]
    rois, grad_out = _f(rois), _f(grad_out)
    st, keep = _roi_levels(grads, scales, k_min, canonical_scale, canonical_level, "nchw")
    R, Cc, P = grad_out.shape[0], grad_out.shape[1], grad_out.shape[2]
    lib().orc_roi_align_bwd(C.byref(st), _p(rois), C.c_int(R), C.c_int(Cc), C.c_int(P), C.c_int(sampling_ratio),
                            C.c_int(int(aligned)), _p(grad_out), C.c_int64(Cc * P * P), C.c_int64(P * P), C.c_int64(P), C.c_int64(1))
    return keep  # the (contiguous NCHW) gradient arrays


def roi_pool_fwd(feats, rois, P, scales, k_min=2, canonical_scale=224.0, canonical_level=4, layout="nchw"):
    """-> (out [R,C,P,P] f32, argmax [R,C,P,P] i32 = y*W+x inside the RoI's level, -1 for empty bins)."""
    feats = [_f(f) for f in feats]
    rois = _f(rois)
    st, keep = _roi_levels(feats, scales, k_min, canonical_scale, canonical_level, layout)
    R, Cc = rois.shape[0], feats[0].shape[1]
    out = np.zeros((R, Cc, P, P), np.float32)
    arg = np.full((R, Cc, P, P), -1, np.int32)
    lib().orc_roi_pool_fwd(C.byref(st), _p(rois), C.c_int(R), C.c_int(Cc), C.c_int(P), _p(out), _p(arg),
                           C.c_int64(Cc * P * P), C.c_int64(P * P), C.c_int64(P), C.c_int64(1))
    del keep
    return out, arg


def roi_pool_bwd(feat_shapes, rois, grad_out, argmax, scales, k_min=2, canonical_scale=224.0, canonical_level=4):
    grads = [np.zeros(s, np.float32) for s in feat_shapes]
    rois, grad_out = _f(rois), _f(grad_out)
    argmax = np.ascontiguousarray(argmax, np.int32)
    st, keep = _roi_levels(grads, scales, k_min, canonical_scale, canonical_level, "nchw")
    R, Cc, P = grad_out.shape[0], grad_out.shape[1], grad_out.shape[2]
    lib().orc_roi_pool_bwd(C.byref(st), _p(rois), C.c_int(R), C.c_int(Cc), C.c_int(P), _p(grad_out), _p(argmax),
                           C.c_int64(Cc * P * P), C.c_int64(P * P), C.c_int64(P), C.c_int64(1))
    return keep


# ---------------------------------------------------------------------------- YOLOv5
def _yolo_levels(preds, strides, anchors):
    """preds: list of [B,na,H,W,no] float32 arrays (contiguous); anchors: [L,na,2] pixels."""
    st = YoloLevels()
    st.n_levels = len(preds)
    st.na = preds[0].shape[1]
    st.no = preds[0].shape[4]
    keep = []
    for l, p in enumerate(preds):
        p = _f(p)
        keep.append(p)
        B, na, H, W, no = p.shape
        st.pred[l] = p.ctypes.data
        st.h[l], st.w[l] = H, W
        st.stride[l] = strides[l]
        for a in range(na):
            st.anchor_w[l][a] = float(anchors[l][a][0])
            st.anchor_h[l][a] = float(anchors[l][a][1])
        st.sb[l], st.sa[l], st.sy[l], st.sx[l], st.so[l] = na * H * W * no, H * W * no, W * no, no, 1
    return st, keep


def yolo_decode(preds, strides, anchors):
    st, keep = _yolo_levels(preds, strides, anchors)
    B = keep[0].shape[0]
    n = sum(p.shape[1] * p.shape[2] * p.shape[3] for p in keep)
    out = np.empty((B, n, st.no), np.float32)
    lib().orc_yolo_decode(C.byref(st), C.c_int(B), _p(out))
    return out


def yolo_assign(shapes, strides, anchors, gt_boxes, gt_cls, m_valid, anchor_t=4.0):
    """shapes: list of (na,H,W,no). Returns slot_cell [L,S], slot_tbox [L,S,4], slot_cls [L,S]."""
    gt_boxes, gt_cls, m_valid = _f(gt_boxes), _i(gt_cls), _i(m_valid)
    B, M = gt_cls.shape
    dummy = [np.zeros((1, s[0], s[1], s[2], s[3]), np.float32) for s in shapes]
    st, keep = _yolo_levels(dummy, strides, anchors)
    L, na = len(shapes), shapes[0][0]
    S = 5 * na * B * M
    cell = np.empty((L, S), np.int32)
    tbox = np.empty((L, S, 4), np.float32)
    cls = np.empty((L, S), np.int32)
    lib().orc_yolo_assign(C.byref(st), _p(gt_boxes), _p(gt_cls), _p(m_valid), C.c_int(B), C.c_int(M),
                          C.c_float(anchor_t), _p(cell), _p(tbox), _p(cls))
    return cell, tbox, cls


FOCAL_TERM_BITS = {"cls": 1, "obj": 2}


def _loss_cfg(box_gain, obj_gain, cls_gain, balance, focal=None):
    """focal: None or (gamma, alpha, terms) with terms a subset of ("cls", "obj")."""
    cfg = YoloLossCfg()
    cfg.box_gain, cfg.obj_gain, cfg.cls_gain = box_gain, obj_gain, cls_gain
    for l, b in enumerate(balance):
        cfg.balance[l] = b
    if focal is not None:
        cfg.focal_gamma, cfg.focal_alpha = float(focal[0]), float(focal[1])
        cfg.focal_terms = sum(FOCAL_TERM_BITS[t] for t in focal[2])
    return cfg


def yolo_loss(preds, strides, anchors, slot_cell, slot_tbox, slot_cls, M, box_gain=0.05, obj_gain=1.0,
              cls_gain=0.5, balance=(4.0, 1.0, 0.4), grad_scale=1.0, want_grad=True, focal=None):
    """Returns loss[4], grads (list of [B,na,H,W,no]) or None.  focal = (gamma, alpha, terms) or None."""
    st, keep = _yolo_levels(preds, strides, anchors)
    B = keep[0].shape[0]
    cells = sum(p.shape[0] * p.shape[1] * p.shape[2] * p.shape[3] for p in keep)
    tobj = np.zeros((cells,), np.float32)
    acc = np.zeros((len(keep) * 4,), np.float32)
    loss = np.zeros((4,), np.float32)
    cfg = _loss_cfg(box_gain, obj_gain, cls_gain, balance, focal)
    sc, tb, scl = _i(slot_cell), _f(slot_tbox), _i(slot_cls)
    lib().orc_yolo_loss_fwd(C.byref(st), C.c_int(B), C.c_int(M), _p(sc), _p(tb), _p(scl), C.byref(cfg), _p(loss), _p(tobj), _p(acc))
    grads = None
    if want_grad:
        grads = [np.zeros_like(p) for p in keep]
        gst, gkeep = _yolo_levels(grads, strides, anchors)
        lib().orc_yolo_loss_bwd(C.byref(st), C.byref(gst), C.c_int(B), C.c_int(M), _p(sc), _p(tb), _p(scl), C.byref(cfg),
                                C.c_float(grad_scale), _p(tobj), _p(acc))
        grads = gkeep
    return loss, grads, tobj, acc


def ciou_fwd(pred, target):
    pred, target = _f(pred), _f(target)
    out = np.empty((pred.shape[0],), np.float32)
    lib().orc_ciou_fwd(_p(pred), _p(target), C.c_int64(pred.shape[0]), _p(out))
    return out


def ciou_bwd(pred, target, grad_ciou):
    pred, target, grad_ciou = _f(pred), _f(target), _f(grad_ciou)
    out = np.empty_like(pred)
    lib().orc_ciou_bwd(_p(pred), _p(target), _p(grad_ciou), C.c_int64(pred.shape[0]), _p(out))
    return out


def yolo_nms(dec, conf_thr=0.25, iou_thr=0.45, max_nms=30000, max_det=300):
    dec = _f(dec)
    B, n, no = dec.shape
    det = np.empty((B, max_det, 6), np.float32)
    cnt = np.empty((B,), np.int32)
    lib().orc_yolo_nms(_p(dec), C.c_int(B), C.c_int(n), C.c_int(no), C.c_float(conf_thr), C.c_float(iou_thr),
                       C.c_int(max_nms), C.c_int(max_det), _p(det), _p(cnt))
    return det, cnt


# ---------------------------------------------------------------------------- WBF
def wbf(boxes, scores, labels, model_weight, k_valid, weight_sum, iou_thr=0.55, skip_thr=0.0):
    boxes, scores, labels, mw = _f(boxes), _f(scores), _i(labels), _f(model_weight)
    B, K = scores.shape
    kv = None if k_valid is None else _i(k_valid)
    out = np.empty((B, K, 6), np.float32)
    cnt = np.empty((B,), np.int32)
    lib().orc_wbf(_p(boxes), _p(scores), _p(labels), _p(mw), _p(kv), C.c_int(B), C.c_int(K), C.c_float(weight_sum),
                  C.c_float(iou_thr), C.c_float(skip_thr), _p(out), _p(cnt))
    return out, cnt


# ---------------------------------------------------------------------------- Mosaic-4
def mosaic4(images, gt_boxes, gt_cls, m_valid, src, center, apply, fill=114.0 / 255.0, min_wh=2.0, min_area_ratio=0.1):
    """images [B,3,S,S] f32 (logical NCHW); src [B,4], center [B,2], apply [B] int32.
    -> (out images [B,3,S,S], boxes [B,4M,4], cls [B,4M], count [B])."""
    images = _f(images)
    gt_boxes, gt_cls, m_valid = _f(gt_boxes), _i(gt_cls), _i(m_valid)
    src, center, apply = _i(src), _i(center), _i(apply)
    B, _, S, _ = images.shape
    M = gt_boxes.shape[1]
    out = np.empty_like(images)
    lib().orc_mosaic4(_p(images), C.c_int64(3 * S * S), C.c_int64(S * S), C.c_int64(S), C.c_int64(1), C.c_int(B), C.c_int(S),
                      _p(src), _p(center), _p(apply), C.c_float(fill), _p(out), C.c_int64(3 * S * S), C.c_int64(S * S),
                      C.c_int64(S), C.c_int64(1))
    ob = np.empty((B, 4 * M, 4), np.float32)
    oc = np.empty((B, 4 * M), np.int32)
    cnt = np.empty((B,), np.int32)
    lib().orc_mosaic4_boxes(_p(gt_boxes), _p(gt_cls), _p(m_valid), C.c_int(B), C.c_int(M), C.c_int(S), _p(src), _p(center),
                            _p(apply), C.c_float(min_wh), C.c_float(min_area_ratio), _p(ob), _p(oc), _p(cnt))
    return out, ob, oc, cnt
