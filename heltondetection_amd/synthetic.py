"""Synthetic inputs for tests and benchmarks (no datasets, no network).

Images are uniform [0,1); ground-truth boxes have centres uniform over the image and
log-uniform sides: "coco" = 1..40 boxes per image with sides in [8, S/2] px, "visdrone" =
400..600 boxes with sides in [4, 64] px; boxes are clipped to the image; classes uniform.
Everything is drawn from a seeded ``torch.Generator`` on the CPU (seed = 1234 + rank in the
benchmark, 0 in fixtures) so that every rank / run sees the same data for the same seed.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch


def make_generator(seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return g


def random_boxes(n: int, size: float, gen: torch.Generator, min_side: float = 8.0,
                 max_side: float | None = None) -> torch.Tensor:
    """n xyxy boxes, centres uniform, sides log-uniform in [min_side, max_side], clipped."""
    max_side = size / 2 if max_side is None else max_side
    c = torch.rand((n, 2), generator=gen) * size
    lo, hi = math.log(min_side), math.log(max_side)
    wh = torch.exp(torch.rand((n, 2), generator=gen) * (hi - lo) + lo)
    b = torch.cat([c - wh / 2, c + wh / 2], 1).clamp_(0, size)
    return b


def make_targets(batch: int, size: int, gen: torch.Generator, style: str = "coco",
                 num_classes: int = 80, m_pad: int | None = None
                 ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Padded ground truth: boxes [B,M,4] f32, cls [B,M] i32, m_valid [B] i32."""
    if style == "coco":
        lo_n, hi_n, lo_s, hi_s = 1, 40, 8.0, size / 2
    elif style == "visdrone":
        lo_n, hi_n, lo_s, hi_s = 400, 600, 4.0, 64.0
    else:
        raise ValueError(f"unknown style {style!r}")
    counts = torch.randint(lo_n, hi_n + 1, (batch,), generator=gen)
    M = int(m_pad if m_pad is not None else hi_n)
    boxes = torch.zeros((batch, M, 4), dtype=torch.float32)
    cls = torch.zeros((batch, M), dtype=torch.int32)
    for b in range(batch):
        n = min(int(counts[b]), M)
        bx = random_boxes(n, float(size), gen, lo_s, hi_s)
        # degenerate boxes after clipping are nudged to at least 2 px
        bx[:, 2:] = torch.maximum(bx[:, 2:], bx[:, :2] + 2.0).clamp_(max=float(size))
        bx[:, :2] = torch.minimum(bx[:, :2], bx[:, 2:] - 2.0).clamp_(min=0.0)
        boxes[b, :n] = bx
        cls[b, :n] = torch.randint(0, num_classes, (n,), generator=gen, dtype=torch.int32)
        counts[b] = n
    return boxes, cls, counts.to(torch.int32)


def make_images(batch: int, size: int, gen: torch.Generator) -> torch.Tensor:
    return torch.rand((batch, 3, size, size), generator=gen, dtype=torch.float32)
