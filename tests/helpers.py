"""Shared seeded input builders for the parity tests (CPU tensors; tests move them to the GPU)."""
import math

import numpy as np
import torch

from heltondetection_amd import synthetic as syn

import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_v1.npz")
YOLO_ANCHORS = [[[10, 13], [16, 30], [33, 23]], [[30, 61], [62, 45], [59, 119]], [[116, 90], [156, 198], [373, 326]]]
YOLO_STRIDES = [8, 16, 32]


def boxes_batch(B, N, size, seed, min_side=8.0, max_side=None):
    g = syn.make_generator(seed)
    return torch.stack([syn.random_boxes(N, float(size), g, min_side, max_side) for _ in range(B)])


def clustered_boxes(B, N, size, seed, n_centers=40, jitter=12.0):
    """Boxes bunched around a few centres so NMS has real work (many overlaps)."""
    g = syn.make_generator(seed)
    out = []
    for _ in range(B):
        c = syn.random_boxes(n_centers, float(size), g, 16.0, size / 3)
        pick = torch.randint(0, n_centers, (N,), generator=g)
        b = c[pick] + torch.randn((N, 4), generator=g) * jitter
        b = torch.stack([torch.minimum(b[:, 0], b[:, 2] - 1), torch.minimum(b[:, 1], b[:, 3] - 1), b[:, 2], b[:, 3]], 1)
        out.append(b.clamp(0, size))
    return torch.stack(out)


def distinct_scores(B, N, seed):
    g = syn.make_generator(seed)
    s = torch.rand((B, N), generator=g)
    # make ties impossible: add a tiny index-dependent offset and re-round
    return (s * 0.999 + torch.arange(N)[None, :] * 1e-7).float()


def yolo_preds(B, size, no, seed, scale=1.0):
    g = syn.make_generator(seed)
    return [torch.randn((B, 3, size // s, size // s, no), generator=g) * scale for s in YOLO_STRIDES]


def yolo_shapes(size, no):
    return [(3, size // s, size // s, no) for s in YOLO_STRIDES]


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12)) if a.size else 0.0


def max_rel_elem(a, b, floor=1e-6):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float((np.abs(a - b) / (np.abs(b) + floor)).max()) if a.size else 0.0


def elem_excess(a, b, rtol):
    """Element-wise parity bound (north_star: "within 1e-4 rel for IoU/RoIAlign floats"): the largest value of
    |a - b| / (rtol * |b| + floor) over all elements, with floor = rtol * median(|b| over the non-zero elements of b) —
    so every element must match to `rtol` of its OWN magnitude, and an element much smaller than the typical one (a
    near-cancelling sum) to `rtol` of the typical magnitude.  <= 1 passes.  Unlike rel_err (max-norm), a small bin that is
    10 % wrong fails here."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    assert a.shape == b.shape, (a.shape, b.shape)
    nz = np.abs(b[b != 0])
    floor = rtol * (float(np.median(nz)) if nz.size else 1.0)
    return float((np.abs(a - b) / (rtol * np.abs(b) + floor)).max())


def assert_elem_close(a, b, rtol, what=""):
    ex = elem_excess(a, b, rtol)
    assert ex <= 1.0, f"{what}: element-wise bound |a-b| <= {rtol}*|b| + {rtol}*median|b| exceeded by a factor {ex:.3g}"
