"""mAP evaluator.  The reference evaluates with pycocotools (README.md:61), which is
not installed here; this is a small stand-alone COCO-style evaluator: per class, detections sorted by
score are greedily matched to the unmatched ground truth of highest IoU >= threshold, AP is the
101-point interpolated area under the precision/recall curve, mAP50 averages classes with at least
one ground-truth box, mAP50-95 averages the IoU thresholds 0.50:0.05:0.95 as well.

Host-side NumPy on purpose: evaluation is outside the training hot path and runs once per epoch on
gathered detections; the IoU here is the library convention of DESIGN.md §2.2."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np


def _iou_matrix(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    if len(a) == 0 or len(b) == 0:
        return np.zeros((len(a), len(b)), np.float32)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    iw = np.clip(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
    inter = iw * ih
    union = area_a[:, None] + area_b[None, :] - inter
    return np.where(union > 0, inter / np.maximum(union, 1e-30), 0.0).astype(np.float32)


class MeanAP:
    """Accumulates (detections, ground truth) per image; `compute()` returns mAP50, mAP50-95, per-class AP50."""

    def __init__(self, num_classes: int, iou_thresholds: Sequence[float] = tuple(np.arange(0.5, 0.96, 0.05))):
        self.nc = int(num_classes)
        self.thr = np.asarray(iou_thresholds, np.float64)
        self._det: List[Tuple[int, np.ndarray]] = []   # (image id, [n,6] x1 y1 x2 y2 score cls)
        self._gt: Dict[int, Tuple[np.ndarray, np.ndarray]] = {}
        self._next = 0

    def add_image(self, det: np.ndarray, gt_boxes: np.ndarray, gt_cls: np.ndarray) -> None:
        i = self._next
        self._next += 1
        self._det.append((i, np.asarray(det, np.float32).reshape(-1, 6)))
        self._gt[i] = (np.asarray(gt_boxes, np.float32).reshape(-1, 4), np.asarray(gt_cls).astype(np.int64).reshape(-1))

    def add_batch(self, det, det_count, gt_boxes, gt_cls, m_valid) -> None:
        """Padded batch tensors as the detectors return them (torch tensors or arrays)."""
        to_np = lambda t: t.detach().float().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
        det, det_count, gt_boxes, gt_cls, m_valid = map(to_np, (det, det_count, gt_boxes, gt_cls, m_valid))
        for b in range(det.shape[0]):
            self.add_image(det[b, : int(det_count[b])], gt_boxes[b, : int(m_valid[b])], gt_cls[b, : int(m_valid[b])])

    def compute(self) -> Dict[str, object]:
        T = len(self.thr)
        ap = np.full((self.nc, T), np.nan)
        rec_pts = np.linspace(0.0, 1.0, 101)
        for c in range(self.nc):
            n_gt = sum(int((g[1] == c).sum()) for g in self._gt.values())
            if n_gt == 0:
                continue
            rows = []  # (score, image, det index within class)
            per_img = {}
            for img, d in self._det:
                dc = d[d[:, 5] == c]
                if len(dc):
                    order = np.argsort(-dc[:, 4], kind="stable")
                    dc = dc[order]
                    gb = self._gt[img][0][self._gt[img][1] == c]
                    per_img[img] = (dc, _iou_matrix(dc[:, :4], gb))
                    rows += [(float(s), img, k) for k, s in enumerate(dc[:, 4])]
            if not rows:
                ap[c] = 0.0
                continue
            rows.sort(key=lambda r: -r[0])
            tp = np.zeros((T, len(rows)), bool)
            used = {img: np.zeros((T, m[1].shape[1]), bool) for img, m in per_img.items()}
            for j, (_, img, k) in enumerate(rows):
                iou = per_img[img][1][k]
                if iou.size == 0:
                    continue
                for t in range(T):
                    cand = np.where(~used[img][t], iou, -1.0)
                    g = int(cand.argmax())
                    if cand[g] >= self.thr[t]:
                        used[img][t, g] = True
                        tp[t, j] = True
            for t in range(T):
                ctp = np.cumsum(tp[t])
                cfp = np.cumsum(~tp[t])
                recall = ctp / n_gt
                precision = ctp / np.maximum(ctp + cfp, 1)
                for i in range(len(precision) - 2, -1, -1):   # monotone envelope
                    precision[i] = max(precision[i], precision[i + 1])
                idx = np.searchsorted(recall, rec_pts, side="left")
                p = np.where(idx < len(precision), precision[np.minimum(idx, len(precision) - 1)], 0.0)
                ap[c, t] = p.mean()
        valid = ~np.isnan(ap[:, 0])
        t50 = int(np.argmin(np.abs(self.thr - 0.5)))
        return {"mAP50": float(np.nanmean(ap[valid, t50])) if valid.any() else 0.0,
                "mAP50_95": float(np.nanmean(ap[valid])) if valid.any() else 0.0,
                "AP50_per_class": ap[:, t50], "classes_with_gt": int(valid.sum())}
