"""Second, independent CPU restatement of the detection hot path in vectorised PyTorch.

TEST INFRASTRUCTURE ONLY — PARITY UNPINNED (the reference snapshot has no source; SURVEY.md
§0, §8c).  This file restates the same published algorithms as oracle/hd_oracle.c but is
written separately (tensor ops + autograd instead of serial loops), so the two cross-check
each other; gradients here come from autograd, in the C file they are hand-derived.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg
("our PyTorch-CPU restatement").  The product package never imports it.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import torch

EPS = 1e-7
DELTA_CLAMP = math.log(1000.0 / 16.0)


# ------------------------------------------------------------------------------- IoU
def box_iou(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a [...,N,4], b [...,M,4] xyxy -> [...,N,M]; 0 where the union is <= 0."""
    area_a = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area_b = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    lt = torch.maximum(a[..., :, None, :2], b[..., None, :, :2])
    rb = torch.minimum(a[..., :, None, 2:], b[..., None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    union = (area_a[..., :, None] + area_b[..., None, :]) - inter
    return torch.where(union > 0, inter / union, torch.zeros_like(inter))


def pairwise_iou(a, b, m_valid=None):
    out = box_iou(a, b)
    if m_valid is not None:
        M = b.shape[1]
        pad = torch.arange(M)[None, :] >= m_valid[:, None].long()
        out = torch.where(pad[:, None, :], torch.full_like(out, -1.0), out)
    return out


def iou_assign(a, b, m_valid=None):
    iou = pairwise_iou(a, b, m_valid)  # [B,N,M], padded = -1
    B, N, M = iou.shape
    if M == 0:
        return (torch.full((B, N), -1.0), torch.full((B, N), -1, dtype=torch.int32),
                torch.zeros((B, 0)), torch.zeros((B, 0), dtype=torch.int32))
    mx, am = iou.max(dim=2)
    # first index of the max (torch.max may return any index on ties)
    am = (iou == mx[..., None]).to(torch.int32).argmax(dim=2)
    none = mx < 0
    am = torch.where(none, torch.full_like(am, -1), am)
    gmx, gam = iou.max(dim=1)
    gam = (iou == gmx[:, None, :]).to(torch.int32).argmax(dim=1)
    gnone = gmx < 0
    gam = torch.where(gnone, torch.full_like(gam, -1), gam)
    return mx, am.to(torch.int32), gmx, gam.to(torch.int32)


# ------------------------------------------------------------------------------- anchors
def anchor_base(sizes: Sequence[float], ratios: Sequence[float]) -> torch.Tensor:
    sizes_t = torch.tensor(sizes, dtype=torch.float32)
    hr = torch.sqrt(torch.tensor(ratios, dtype=torch.float32))
    w = (sizes_t[None, :] / hr[:, None]).reshape(-1)
    h = (sizes_t[None, :] * hr[:, None]).reshape(-1)
    return torch.stack([-w * 0.5, -h * 0.5, w * 0.5, h * 0.5], dim=1)


def anchor_grid(feat_h, feat_w, strides, base_anchors, offset=0.5) -> torch.Tensor:
    out = []
    for l, (H, W, s) in enumerate(zip(feat_h, feat_w, strides)):
        base = torch.as_tensor(base_anchors[l], dtype=torch.float32)
        sx = (torch.arange(W, dtype=torch.float32) + offset) * s
        sy = (torch.arange(H, dtype=torch.float32) + offset) * s
        yy, xx = torch.meshgrid(sy, sx, indexing="ij")
        shift = torch.stack([xx, yy, xx, yy], dim=-1)  # [H,W,4]
        out.append((shift[:, :, None, :] + base[None, None, :, :]).reshape(-1, 4))
    return torch.cat(out, 0)


# ------------------------------------------------------------------------------- codec
def box_encode(src, dst):
    sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
    scx, scy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
    dw, dh = dst[:, 2] - dst[:, 0], dst[:, 3] - dst[:, 1]
    dcx, dcy = dst[:, 0] + 0.5 * dw, dst[:, 1] + 0.5 * dh
    return torch.stack([(dcx - scx) / sw, (dcy - scy) / sh, torch.log(dw / sw), torch.log(dh / sh)], 1)


def box_decode(anchors, deltas, img_h=0.0, img_w=0.0):
    aw, ah = anchors[..., 2] - anchors[..., 0], anchors[..., 3] - anchors[..., 1]
    acx, acy = anchors[..., 0] + 0.5 * aw, anchors[..., 1] + 0.5 * ah
    dw, dh = deltas[..., 2].clamp(max=DELTA_CLAMP), deltas[..., 3].clamp(max=DELTA_CLAMP)
    cx, cy = acx + deltas[..., 0] * aw, acy + deltas[..., 1] * ah
    w, h = aw * torch.exp(dw), ah * torch.exp(dh)
    x1, y1, x2, y2 = cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h
    if img_w > 0:
        x1, x2 = x1.clamp(0, img_w), x2.clamp(0, img_w)
        y1, y2 = y1.clamp(0, img_h), y2.clamp(0, img_h)
    return torch.stack([x1, y1, x2, y2], -1)


# ------------------------------------------------------------------------------- NMS
def _order_desc(scores: torch.Tensor) -> torch.Tensor:
    """descending, ties -> lower index (stable sort of the negated key)."""
    return torch.sort(scores, descending=True, stable=True).indices


def nms_single(boxes, scores, cls=None, score_thr=-float("inf"), iou_thr=0.5, max_candidates=0,
               max_out=None) -> torch.Tensor:
    idx = torch.nonzero(scores > score_thr).flatten()
    idx = idx[_order_desc(scores[idx])]
    if max_candidates > 0:
        idx = idx[:max_candidates]
    max_out = len(idx) if max_out is None else max_out
    if len(idx) == 0:
        return idx
    iou = box_iou(boxes[idx], boxes[idx])
    sup = iou > iou_thr
    if cls is not None:
        c = cls[idx]
        sup &= c[:, None] == c[None, :]
    alive = torch.ones(len(idx), dtype=torch.bool)
    keep = []
    for i in range(len(idx)):
        if not alive[i]:
            continue
        keep.append(i)
        if len(keep) >= max_out:
            break
        alive &= ~sup[i]
        alive[i] = False
    return idx[torch.tensor(keep, dtype=torch.long)]


def nms_batched(boxes, scores, cls=None, n_valid=None, score_thr=-float("inf"), iou_thr=0.5,
                max_candidates=0, max_out=None):
    B, N = scores.shape
    max_out = N if max_out is None else max_out
    keep = torch.full((B, max_out), -1, dtype=torch.int32)
    cnt = torch.zeros((B,), dtype=torch.int32)
    for b in range(B):
        nv = N if n_valid is None else int(n_valid[b])
        k = nms_single(boxes[b, :nv], scores[b, :nv], None if cls is None else cls[b, :nv], score_thr,
                       iou_thr, max_candidates, max_out)
        keep[b, : len(k)] = k.to(torch.int32)
        cnt[b] = len(k)
    return keep, cnt


def rpn_proposals(logits, deltas, anchors, level_sizes, pre_nms_topk, post_nms_topk, iou_thr, min_size,
                  img_h, img_w):
    """-> (rois, scores, count, index): index [B,post] = anchor index of every kept proposal, -1 padded."""
    B, NA = logits.shape
    rois = torch.zeros((B, post_nms_topk, 4))
    sc = torch.zeros((B, post_nms_topk))
    cnt = torch.zeros((B,), dtype=torch.int32)
    idx = torch.full((B, post_nms_topk), -1, dtype=torch.int32)
    for b in range(B):
        cb, cs, cl, ci = [], [], [], []
        off = 0
        for l, n in enumerate(level_sizes):
            lg = logits[b, off:off + n]
            order = _order_desc(lg)[: min(n, pre_nms_topk)] + off
            bx = box_decode(anchors[order], deltas[b, order], img_h, img_w)
            ok = ((bx[:, 2] - bx[:, 0]) >= min_size) & ((bx[:, 3] - bx[:, 1]) >= min_size)
            cb.append(bx)
            cs.append(torch.where(ok, logits[b, order], torch.full_like(logits[b, order], -float("inf"))))
            cl.append(torch.full((len(order),), l, dtype=torch.int32))
            ci.append(order.to(torch.int32))
            off += n
        cb, cs, cl, ci = torch.cat(cb), torch.cat(cs), torch.cat(cl), torch.cat(ci)
        k = nms_single(cb, cs, cl, -float("inf"), iou_thr, 0, post_nms_topk)
        cnt[b] = len(k)
        rois[b, : len(k)] = cb[k]
        sc[b, : len(k)] = torch.sigmoid(cs[k])
        idx[b, : len(k)] = ci[k]
    return rois, sc, cnt, idx


# ------------------------------------------------------------------------------- RoIAlign
def roi_level_map(rois, k_min, n_levels, canonical_scale=224.0, canonical_level=4):
    area = (rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2])
    lvl = torch.zeros(len(rois), dtype=torch.int32)
    for k in range(k_min + 1, k_min + n_levels):
        t = canonical_scale * (2.0 ** (k - canonical_level))
        lvl += (area >= torch.tensor(t * t, dtype=torch.float32)).to(torch.int32)
    return lvl


def _bilinear(feat: torch.Tensor, y: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """feat [C,H,W]; y,x [...] sample coordinates -> [C,...] (public RoIAlign boundary rules)."""
    C, H, W = feat.shape
    outside = (y < -1.0) | (y > H) | (x < -1.0) | (x > W)
    y = y.clamp(min=0)
    x = x.clamp(min=0)
    y_low = y.floor().long().clamp(max=H - 1)
    x_low = x.floor().long().clamp(max=W - 1)
    y_high = (y_low + 1).clamp(max=H - 1)
    x_high = (x_low + 1).clamp(max=W - 1)
    y = torch.where(y_low >= H - 1, y_low.to(y.dtype), y)
    x = torch.where(x_low >= W - 1, x_low.to(x.dtype), x)
    ly, lx = y - y_low.to(y.dtype), x - x_low.to(x.dtype)
    hy, hx = 1.0 - ly, 1.0 - lx
    v = (feat[:, y_low, x_low] * (hy * hx) + feat[:, y_low, x_high] * (hy * lx)
         + feat[:, y_high, x_low] * (ly * hx) + feat[:, y_high, x_high] * (ly * lx))
    return torch.where(outside, torch.zeros_like(v), v)


def roi_align(feats: List[torch.Tensor], rois: torch.Tensor, P: int, scales: Sequence[float],
              sampling_ratio=2, aligned=True, k_min=2, canonical_scale=224.0, canonical_level=4):
    """feats: list of [B,C,H,W]; rois [R,5]; -> [R,C,P,P]. Differentiable w.r.t. feats."""
    lvl = roi_level_map(rois.detach(), k_min, len(feats), canonical_scale, canonical_level)
    out = []
    off = 0.5 if aligned else 0.0
    for r in range(rois.shape[0]):
        l = int(lvl[r])
        f = feats[l][int(rois[r, 0])]
        s = scales[l]
        sw_, sh_ = rois[r, 1] * s - off, rois[r, 2] * s - off
        ew, eh = rois[r, 3] * s - off, rois[r, 4] * s - off
        rw, rh = ew - sw_, eh - sh_
        if not aligned:
            rw, rh = rw.clamp(min=1.0), rh.clamp(min=1.0)
        bh, bw = rh / P, rw / P
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(rh) / P))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(rw) / P))
        cnt = max(gh * gw, 1)
        ph = torch.arange(P, dtype=torch.float32)
        iy = torch.arange(gh, dtype=torch.float32)
        ix = torch.arange(gw, dtype=torch.float32)
        ys = sh_ + ph[:, None] * bh + (iy[None, :] + 0.5) * bh / max(gh, 1)  # [P,gh]
        xs = sw_ + ph[:, None] * bw + (ix[None, :] + 0.5) * bw / max(gw, 1)  # [P,gw]
        Y = ys[:, None, :, None].expand(P, P, gh, gw)
        X = xs[None, :, None, :].expand(P, P, gh, gw)
        v = _bilinear(f, Y, X)  # [C,P,P,gh,gw]
        out.append(v.sum(dim=(3, 4)) / cnt)
    if not out:
        return torch.zeros((0, feats[0].shape[1], P, P))
    return torch.stack(out, 0)


def _round_half_away(v: float) -> int:
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


def roi_pool(feats: List[torch.Tensor], rois: torch.Tensor, P: int, scales: Sequence[float], k_min=2,
             canonical_scale=224.0, canonical_level=4):
    """Quantised RoI max-pooling (README.md:73 "-RoIPooling"): slices + amax, differentiable w.r.t. feats
    (autograd routes the gradient to the first maximum of a slice only when maxima are unique)."""
    lvl = roi_level_map(rois.detach(), k_min, len(feats), canonical_scale, canonical_level)
    out = []
    f32 = torch.float32
    for r in range(rois.shape[0]):
        l = int(lvl[r])
        f = feats[l][int(rois[r, 0])]
        Hh, Ww = f.shape[1], f.shape[2]
        s = torch.tensor(scales[l], dtype=f32)
        x1, y1, x2, y2 = [_round_half_away(float(rois[r, k].to(f32) * s)) for k in (1, 2, 3, 4)]
        rw, rh = max(x2 - x1 + 1, 1), max(y2 - y1 + 1, 1)
        bw = torch.tensor(rw, dtype=f32) / torch.tensor(P, dtype=f32)
        bh = torch.tensor(rh, dtype=f32) / torch.tensor(P, dtype=f32)
        rows = []
        for ph in range(P):
            hs = min(max(int(torch.floor(torch.tensor(ph, dtype=f32) * bh)) + y1, 0), Hh)
            he = min(max(int(torch.ceil(torch.tensor(ph + 1, dtype=f32) * bh)) + y1, 0), Hh)
            cols = []
            for pw in range(P):
                ws = min(max(int(torch.floor(torch.tensor(pw, dtype=f32) * bw)) + x1, 0), Ww)
                we = min(max(int(torch.ceil(torch.tensor(pw + 1, dtype=f32) * bw)) + x1, 0), Ww)
                if he <= hs or we <= ws:
                    cols.append(torch.zeros((f.shape[0],), dtype=f.dtype))
                else:
                    cols.append(f[:, hs:he, ws:we].amax(dim=(1, 2)))
            rows.append(torch.stack(cols, 1))
        out.append(torch.stack(rows, 1))
    if not out:
        return torch.zeros((0, feats[0].shape[1], P, P))
    return torch.stack(out, 0)


# ------------------------------------------------------------------------------- CIoU
def ciou_xyxy(p: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """CIoU of matched xyxy boxes (v6.1 bbox_iou form; alpha is detached)."""
    w1, h1 = p[..., 2] - p[..., 0], p[..., 3] - p[..., 1] + EPS
    w2, h2 = t[..., 2] - t[..., 0], t[..., 3] - t[..., 1] + EPS
    inter = (torch.minimum(p[..., 2], t[..., 2]) - torch.maximum(p[..., 0], t[..., 0])).clamp(min=0) * \
            (torch.minimum(p[..., 3], t[..., 3]) - torch.maximum(p[..., 1], t[..., 1])).clamp(min=0)
    union = w1 * h1 + w2 * h2 - inter + EPS
    iou = inter / union
    cw = torch.maximum(p[..., 2], t[..., 2]) - torch.minimum(p[..., 0], t[..., 0])
    ch = torch.maximum(p[..., 3], t[..., 3]) - torch.minimum(p[..., 1], t[..., 1])
    c2 = cw ** 2 + ch ** 2 + EPS
    rho2 = ((t[..., 0] + t[..., 2] - p[..., 0] - p[..., 2]) ** 2 + (t[..., 1] + t[..., 3] - p[..., 1] - p[..., 3]) ** 2) / 4
    v = (4 / math.pi ** 2) * (torch.atan(w2 / h2) - torch.atan(w1 / h1)) ** 2
    with torch.no_grad():
        alpha = v / (v - iou + (1 + EPS))
    return iou - (rho2 / c2 + v * alpha)


# ------------------------------------------------------------------------------- YOLOv5
def yolo_decode(preds: List[torch.Tensor], strides, anchors) -> torch.Tensor:
    """preds: list of [B,na,H,W,no]; anchors [L,na,2] px -> [B, n, no] (level, anchor, y, x)."""
    z = []
    for l, p in enumerate(preds):
        B, na, H, W, no = p.shape
        yv, xv = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        grid = torch.stack((xv, yv), 2).view(1, 1, H, W, 2)
        ag = torch.as_tensor(anchors[l], dtype=torch.float32).view(1, na, 1, 1, 2)
        y = p.sigmoid()
        xy = (y[..., 0:2] * 2 - 0.5 + grid) * strides[l]
        wh = (y[..., 2:4] * 2) ** 2 * ag
        z.append(torch.cat((xy, wh, y[..., 4:]), -1).view(B, -1, no))
    return torch.cat(z, 1)


def yolo_build_targets(shapes, strides, anchors, gt_boxes, gt_cls, m_valid, anchor_t=4.0):
    """Compact (not slot) form, per level: (b, a, gj, gi, tbox[4], tcls, slot) sorted by slot."""
    B, M = gt_cls.shape
    res = []
    for l, (na, H, W, no) in enumerate(shapes):
        st = strides[l]
        g = gt_boxes  # [B,M,4]
        gx, gy = ((g[..., 0] + g[..., 2]) * 0.5) / st, ((g[..., 1] + g[..., 3]) * 0.5) / st
        gw, gh = (g[..., 2] - g[..., 0]) / st, (g[..., 3] - g[..., 1]) / st
        valid = torch.arange(M)[None, :] < m_valid[:, None].long()  # [B,M]
        anc = torch.as_tensor(anchors[l], dtype=torch.float32) / st  # [na,2]
        rw, rh = gw[None] / anc[:, 0, None, None], gh[None] / anc[:, 1, None, None]  # [na,B,M]
        worst = torch.maximum(torch.maximum(rw, 1 / rw), torch.maximum(rh, 1 / rh))
        match = (worst < anchor_t) & valid[None]
        fx, fy = gx - gx.floor(), gy - gy.floor()
        ix, iy = W - gx, H - gy
        fix, fiy = ix - ix.floor(), iy - iy.floor()
        sels = [torch.ones_like(valid), (fx < 0.5) & (gx > 1), (fy < 0.5) & (gy > 1),
                (fix < 0.5) & (ix > 1), (fiy < 0.5) & (iy > 1)]
        offs = [(0.0, 0.0), (0.5, 0.0), (0.0, 0.5), (-0.5, 0.0), (0.0, -0.5)]
        rows = []
        for k in range(5):
            m = match & sels[k][None]  # [na,B,M]
            a_i, b_i, t_i = torch.nonzero(m, as_tuple=True)
            gi = (gx[b_i, t_i] - offs[k][0]).long().clamp(0, W - 1)
            gj = (gy[b_i, t_i] - offs[k][1]).long().clamp(0, H - 1)
            tb = torch.stack([gx[b_i, t_i] - gi, gy[b_i, t_i] - gj, gw[b_i, t_i], gh[b_i, t_i]], 1)
            slot = ((k * na + a_i) * B + b_i) * M + t_i
            rows.append((b_i, a_i, gj, gi, tb, gt_cls[b_i, t_i].long(), slot))
        cat = [torch.cat([r[i] for r in rows]) for i in range(7)]
        res.append(tuple(cat))
    return res


def focal_bce_mean(logits, target, gamma, alpha):
    """FocalLoss over BCEWithLogits (the public ultralytics form), mean reduction."""
    bce = torch.nn.functional.binary_cross_entropy_with_logits(logits, target, reduction="none")
    p = torch.sigmoid(logits)
    p_t = target * p + (1 - target) * (1 - p)
    mod = (1.0 - p_t) ** gamma
    af = target * alpha + (1 - target) * (1 - alpha) if alpha > 0 else 1.0
    return (bce * af * mod).mean()


def yolo_loss(preds, strides, anchors, gt_boxes, gt_cls, m_valid, anchor_t=4.0, box_gain=0.05, obj_gain=1.0,
              cls_gain=0.5, balance=(4.0, 1.0, 0.4), focal=None):
    """v6.1 ComputeLoss form (gr=1, no label smoothing, duplicates -> max). Returns (total, lbox, lobj, lcls).
    focal = (gamma, alpha, terms) wraps the BCE terms named in `terms` ("cls", "obj")."""
    def bce_cls(x, t):
        if focal is not None and "cls" in focal[2]:
            return focal_bce_mean(x, t, focal[0], focal[1])
        return torch.nn.functional.binary_cross_entropy_with_logits(x, t)

    def bce_obj(x, t):
        if focal is not None and "obj" in focal[2]:
            return focal_bce_mean(x, t, focal[0], focal[1])
        return torch.nn.functional.binary_cross_entropy_with_logits(x, t)

    shapes = [tuple(p.shape[1:]) for p in preds]
    B = preds[0].shape[0]
    tg = yolo_build_targets(shapes, strides, anchors, gt_boxes, gt_cls, m_valid, anchor_t)
    lbox = torch.zeros(()); lobj = torch.zeros(()); lcls = torch.zeros(())
    for l, p in enumerate(preds):
        b, a, gj, gi, tbox, tcls, _ = tg[l]
        na, H, W, no = shapes[l]
        nc = no - 5
        tobj = torch.zeros((B, na, H, W), dtype=p.dtype)
        n = b.shape[0]
        if n:
            ps = p[b, a, gj, gi]
            anc = torch.as_tensor(anchors[l], dtype=torch.float32)[a] / strides[l]
            pxy = ps[:, :2].sigmoid() * 2 - 0.5
            pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * anc
            pb = torch.cat((pxy - pwh / 2, pxy + pwh / 2), 1)
            tb = torch.cat((tbox[:, :2] - tbox[:, 2:] / 2, tbox[:, :2] + tbox[:, 2:] / 2), 1)
            ciou = ciou_xyxy(pb, tb)
            lbox = lbox + (1.0 - ciou).mean()
            flat = ((b * na + a) * H + gj) * W + gi
            tflat = tobj.view(-1)
            tflat.scatter_reduce_(0, flat, ciou.detach().clamp(min=0).to(p.dtype), reduce="amax", include_self=True)
            if nc > 1:
                t = torch.zeros_like(ps[:, 5:])
                t[torch.arange(n), tcls] = 1.0
                lcls = lcls + bce_cls(ps[:, 5:], t)
        lobj = lobj + bce_obj(p[..., 4], tobj) * balance[l]
    lbox, lobj, lcls = lbox * box_gain, lobj * obj_gain, lcls * cls_gain
    return (lbox + lobj + lcls) * B, lbox.detach(), lobj.detach(), lcls.detach()


def yolo_nms(dec, conf_thr=0.25, iou_thr=0.45, max_nms=30000, max_det=300):
    B, n, no = dec.shape
    det = torch.zeros((B, max_det, 6))
    cnt = torch.zeros((B,), dtype=torch.int32)
    for b in range(B):
        r = dec[b]
        best, bc = r[:, 5:].max(1)
        bc = (r[:, 5:] == best[:, None]).to(torch.int32).argmax(1)
        conf = r[:, 4] * best
        bx = torch.stack([r[:, 0] - r[:, 2] * 0.5, r[:, 1] - r[:, 3] * 0.5, r[:, 0] + r[:, 2] * 0.5, r[:, 1] + r[:, 3] * 0.5], 1)
        ok = (r[:, 4] > conf_thr) & (conf > conf_thr)
        sc = torch.where(ok, conf, torch.full_like(conf, -float("inf")))
        k = nms_single(bx, sc, bc, -float("inf"), iou_thr, max_nms, max_det)
        cnt[b] = len(k)
        det[b, : len(k), :4] = bx[k]
        det[b, : len(k), 4] = sc[k]
        det[b, : len(k), 5] = bc[k].float()
    return det, cnt


# ------------------------------------------------------------------------------- WBF
def wbf_single(boxes, scores, labels, model_weight, weight_sum, iou_thr=0.55, skip_thr=0.0):
    keep = torch.nonzero(scores >= skip_thr).flatten()
    conf = (scores * model_weight)[keep]
    order = keep[_order_desc(conf)]
    clusters = []  # dict(sum[4], sconf, n, label, box)
    for i in order.tolist():
        c = float(scores[i] * model_weight[i])
        c32 = (scores[i] * model_weight[i])
        best, best_iou = -1, iou_thr
        for ci, cl in enumerate(clusters):
            if cl["label"] != int(labels[i]):
                continue
            v = float(box_iou(cl["box"][None], boxes[i][None])[0, 0])
            if v > best_iou:
                best, best_iou = ci, v
        if best < 0:
            clusters.append({"sum": torch.zeros(4), "sconf": torch.zeros(()), "n": 0, "label": int(labels[i]), "box": None})
            best = len(clusters) - 1
        cl = clusters[best]
        cl["sum"] = cl["sum"] + c32 * boxes[i]
        cl["sconf"] = cl["sconf"] + c32
        cl["n"] += 1
        cl["box"] = cl["sum"] / cl["sconf"]
        del c
    rows = []
    for ci, cl in enumerate(clusters):
        n = torch.tensor(float(cl["n"]))
        ws = torch.tensor(float(weight_sum))
        sc = (cl["sconf"] / n) * torch.minimum(ws, n) / ws
        rows.append((float(sc), cl["label"], ci, cl["box"]))
    rows.sort(key=lambda r: (-r[0], r[1], r[2]))
    out = torch.zeros((len(rows), 6))
    for i, r in enumerate(rows):
        out[i, :4] = r[3]
        out[i, 4] = r[0]
        out[i, 5] = r[1]
    return out


# ------------------------------------------------------------------------------- Mosaic-4
def mosaic4(images, gt_boxes, gt_cls, m_valid, src, center, apply, fill=114.0 / 255.0, min_wh=2.0, min_area_ratio=0.1):
    """Independent restatement of the Mosaic-4 convention: builds the 2S x 2S canvas by
    pasting whole source images with slice arithmetic (later tiles never overlap earlier ones), then averages
    2x2 blocks in the order ((tl + tr) + (bl + br)) * 0.25."""
    B, _, S, _ = images.shape
    M = gt_boxes.shape[1]
    out = torch.empty_like(images)
    ob = torch.zeros((B, 4 * M, 4))
    oc = torch.zeros((B, 4 * M), dtype=torch.int32)
    cnt = torch.zeros((B,), dtype=torch.int32)
    for b in range(B):
        if not int(apply[b]):
            s0 = int(src[b, 0])
            out[b] = images[s0]
            n = min(int(m_valid[s0]), M)
            ob[b, :n], oc[b, :n], cnt[b] = gt_boxes[s0, :n], gt_cls[s0, :n], n
            continue
        xc, yc = int(center[b, 0]), int(center[b, 1])
        canvas = torch.full((3, 2 * S, 2 * S), float(fill), dtype=images.dtype)
        boxes, classes = [], []
        for k in range(4):
            ox = xc if (k & 1) else xc - S
            oy = yc if (k & 2) else yc - S
            # visible rectangle of this tile on the canvas
            lx, hx = (xc, min(ox + S, 2 * S)) if (k & 1) else (max(ox, 0), xc)
            ly, hy = (yc, min(oy + S, 2 * S)) if (k & 2) else (max(oy, 0), yc)
            s_ = int(src[b, k])
            if hx > lx and hy > ly:
                canvas[:, ly:hy, lx:hx] = images[s_][:, ly - oy:hy - oy, lx - ox:hx - ox]
            n = min(int(m_valid[s_]), M)
            if n:
                g = gt_boxes[s_, :n].clone()
                area0 = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
                g[:, [0, 2]] = (g[:, [0, 2]] + float(ox)).clamp(float(lx), float(hx))
                g[:, [1, 3]] = (g[:, [1, 3]] + float(oy)).clamp(float(ly), float(hy))
                g = g * 0.5
                w, h = g[:, 2] - g[:, 0], g[:, 3] - g[:, 1]
                keep = (w > min_wh) & (h > min_wh) & ((w * h) * 4.0 > min_area_ratio * area0)
                boxes.append(g[keep])
                classes.append(gt_cls[s_, :n][keep])
        out[b] = ((canvas[:, 0::2, 0::2] + canvas[:, 0::2, 1::2]) + (canvas[:, 1::2, 0::2] + canvas[:, 1::2, 1::2])) * 0.25
        if boxes:
            bb, cc = torch.cat(boxes), torch.cat(classes)
            ob[b, : bb.shape[0]], oc[b, : bb.shape[0]], cnt[b] = bb, cc.to(torch.int32), bb.shape[0]
    return out, ob, oc, cnt
