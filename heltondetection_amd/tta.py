"""Test-time augmentation around the WBF kernel (reference README.md:19 "TTA + WBF").

Views: identity, horizontal flip, and optional rescales.  Each view runs the detector's own
inference path (HIP decode + per-class NMS); boxes are mapped back to the original frame on the GPU,
concatenated into the padded [B, K, ...] layout `ops.wbf` consumes, and fused by the HIP kernel."""
from __future__ import annotations

from typing import Callable, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import ops


@torch.no_grad()
def tta_detect(detect: Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]], images: torch.Tensor,
               scales: Sequence[float] = (1.0,), flips: Sequence[bool] = (False, True),
               view_weights: Sequence[float] | None = None, iou_thr: float = 0.55, skip_thr: float = 0.0,
               size_divisor: int = 32):
    """detect(images) -> (det [B, D, 6] = x1 y1 x2 y2 score cls, count [B]).  Returns fused (out [B,K,6], count [B])."""
    B, _, H, W = images.shape
    views = [(s, f) for s in scales for f in flips]
    weights = list(view_weights) if view_weights is not None else [1.0] * len(views)
    boxes, scores, labels, mws, valid = [], [], [], [], []
    for (s, flip), wv in zip(views, weights):
        x = images
        if s != 1.0:
            nh = max(size_divisor, int(round(H * s / size_divisor)) * size_divisor)
            nw = max(size_divisor, int(round(W * s / size_divisor)) * size_divisor)
            x = F.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False)
        else:
            nh, nw = H, W
        if flip:
            x = torch.flip(x, dims=(3,))
        det, cnt = detect(x.contiguous(memory_format=torch.channels_last))
        bx = det[..., :4].clone()
        if flip:
            x1 = nw - bx[..., 2]
            x2 = nw - bx[..., 0]
            bx[..., 0], bx[..., 2] = x1, x2
        bx[..., 0::2] *= W / nw
        bx[..., 1::2] *= H / nh
        D = det.shape[1]
        ok = torch.arange(D, device=det.device)[None, :] < cnt[:, None]
        boxes.append(bx)
        scores.append(torch.where(ok, det[..., 4], torch.full_like(det[..., 4], -1.0)))  # padding never passes skip_thr
        labels.append(det[..., 5].to(torch.int32))
        mws.append(torch.full((B, D), float(wv), device=det.device))
        valid.append(ok)
    boxes, scores = torch.cat(boxes, 1).contiguous(), torch.cat(scores, 1).contiguous()
    labels, mws = torch.cat(labels, 1).contiguous(), torch.cat(mws, 1).contiguous()
    return ops.wbf(boxes, scores, labels, mws, None, float(sum(weights)), iou_thr, max(skip_thr, 0.0))
