"""Assigners — the plugin point where the HIP IoU / build-targets kernels enter the detectors
(reference README.md:8-9: heads and their components are swappable; SURVEY.md §8a rows
"Batched pairwise IoU" and "YOLOv5 target assignment")."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch

from . import ops
from .registry import ASSIGNERS


@ASSIGNERS.register()
class YOLOv5Assigner:
    """Anchor-shape matching + 2 neighbour cells (v6.1 build_targets form) -> fixed slot arrays."""

    def __init__(self, anchor_t: float = 4.0):
        self.anchor_t = float(anchor_t)

    def __call__(self, shapes, strides, anchors, gt_boxes, gt_cls, m_valid):
        return ops.yolo_assign(shapes, strides, anchors, gt_boxes, gt_cls, m_valid, self.anchor_t)


@ASSIGNERS.register()
class MaxIoUAssigner:
    """IoU-threshold assignment for RPN anchors / RoI proposals on the fused HIP max/argmax kernel.

    labels: 1 = positive, 0 = negative, -1 = ignored;  matched: index of the matched gt (>= 0
    wherever any valid gt exists).  With ``match_low_quality`` the single best box of each gt
    (ties -> lowest index, as the kernel defines) is made positive too.
    """

    def __init__(self, pos_iou_thr: float = 0.7, neg_iou_thr: float = 0.3, match_low_quality: bool = True):
        self.pos, self.neg, self.low = float(pos_iou_thr), float(neg_iou_thr), bool(match_low_quality)

    def __call__(self, boxes: torch.Tensor, gt_boxes: torch.Tensor, m_valid: Optional[torch.Tensor]
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        mx, am, gmx, gam = ops.iou_assign(boxes, gt_boxes, m_valid, with_gt=self.low)
        labels = torch.full_like(am, -1)
        labels = torch.where(mx < self.neg, torch.zeros_like(labels), labels)
        labels = torch.where(mx >= self.pos, torch.ones_like(labels), labels)
        if self.low:
            B, M = gam.shape
            valid = (gam >= 0) & (gmx > 0)
            idx = torch.where(valid, gam, torch.zeros_like(gam)).long()
            # scatter "1" at the best box of every valid gt, and re-point its match to that gt
            upd = torch.zeros_like(labels)
            upd.scatter_(1, idx, valid.to(labels.dtype))
            labels = torch.where(upd > 0, torch.ones_like(labels), labels)
            gt_index = torch.arange(M, device=am.device, dtype=am.dtype)[None, :].expand(B, M)
            forced = torch.full_like(am, -1)
            forced.scatter_(1, idx, torch.where(valid, gt_index, torch.full_like(gt_index, -1)))
            am = torch.where((upd > 0) & (forced >= 0), forced, am)
        return labels, am, mx
