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
            N = labels.shape[1]
            valid = (gam >= 0) & (gmx > 0)
            # Padded gts and gts without any overlap go to a dummy column N (sliced off below): scattering them
            # onto a real box would collide with a valid gt whose best box is that one, and scatter_ with duplicate
            # indices is not deterministic on the GPU.  Two valid gts sharing one best box: every write of `upd` is
            # the same 1, and `forced` takes the higher gt index ('amax' — the last writer of the sequential
            # per-gt loop in the public MaxIoUAssigner).
            idx = torch.where(valid, gam, torch.full_like(gam, N)).long()
            upd = torch.zeros((B, N + 1), dtype=labels.dtype, device=labels.device)
            upd.scatter_(1, idx, torch.ones_like(gam, dtype=labels.dtype))
            upd = upd[:, :N]
            labels = torch.where(upd > 0, torch.ones_like(labels), labels)
            gt_index = torch.arange(M, device=am.device, dtype=am.dtype)[None, :].expand(B, M).contiguous()
            forced = torch.full((B, N + 1), -1, dtype=am.dtype, device=am.device)
            forced.scatter_reduce_(1, idx, gt_index, "amax", include_self=True)
            forced = forced[:, :N]
            am = torch.where(forced >= 0, forced, am)
        return labels, am, mx
