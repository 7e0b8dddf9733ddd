"""Batch-level augmentation on device, the step right before the hot path.

`Mosaic(p)` recombines the images of a batch already resident in HBM: with probability `p` an output
image is a 4-tile mosaic of itself and three other images of the batch around a random centre
(README.md:76-77,129-130: "Mosaic p=0.5 / 1.0").  The random draws are made on the device with a
torch.Generator (no host round trip); train.Trainer makes them eagerly right before each step (`draw`) and the step itself
— eager or replayed from a hipGraph — only runs `apply` on the drawn tensors (rng.py: nothing random inside a captured
region); the pixel / box work is csrc/mosaic.hip."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import ops


class Mosaic:
    def __init__(self, p: float = 0.5, fill: float = 114.0 / 255.0, min_wh: float = 2.0, min_area_ratio: float = 0.1,
                 max_boxes: Optional[int] = None, generator: Optional[torch.Generator] = None):
        if not 0.0 <= p <= 1.0:
            raise ValueError("Mosaic: p must be in [0, 1]")
        self.p, self.fill, self.min_wh, self.min_area_ratio = float(p), float(fill), float(min_wh), float(min_area_ratio)
        self.max_boxes = max_boxes
        self.generator = generator

    def draw(self, B: int, S: int, device) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(src [B,4], center [B,2], apply [B]) int32 on `device`: tile 0 is the image itself, tiles 1-3 are
        uniformly drawn other images (with replacement), the centre is uniform in [S/2, 3S/2]."""
        g = self.generator
        own = torch.arange(B, device=device, dtype=torch.int64)[:, None]
        others = torch.randint(0, max(B - 1, 1), (B, 3), device=device, generator=g)
        others = others + (others >= own).to(others.dtype) if B > 1 else torch.zeros_like(others)
        src = torch.cat([own, others], 1).to(torch.int32)
        center = torch.randint(S // 2, 3 * S // 2 + 1, (B, 2), device=device, generator=g).to(torch.int32)
        apply = (torch.rand((B,), device=device, generator=g) < self.p).to(torch.int32)
        return src, center, apply

    def __call__(self, images, gt_boxes, gt_cls, m_valid):
        B, _, S, _ = images.shape
        return self.apply(images, gt_boxes, gt_cls, m_valid, *self.draw(B, S, images.device))

    def apply(self, images, gt_boxes, gt_cls, m_valid, src, center, apply):
        """The deterministic half: recombine the batch according to drawn (src, center, apply).  train.Trainer calls
        `draw()` eagerly before every step and this inside the (possibly hipGraph-captured) step — see rng.py."""
        out, boxes, cls, cnt = ops.mosaic4(images, gt_boxes, gt_cls, m_valid, src, center, apply, self.fill,
                                           self.min_wh, self.min_area_ratio)
        if self.max_boxes is not None and boxes.shape[1] > self.max_boxes:  # fixed padded width for the assigners
            boxes, cls = boxes[:, : self.max_boxes].contiguous(), cls[:, : self.max_boxes].contiguous()
            cnt = cnt.clamp(max=self.max_boxes)
        return out, boxes, cls, cnt
