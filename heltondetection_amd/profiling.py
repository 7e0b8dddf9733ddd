"""Model size metrics reported next to mAP at evaluation time (reference README.md:24: "FLOPs and Params
added to eval").  `thop` is not in the image, so FLOPs are counted with forward hooks on the modules that
carry them (convolutions and linear layers: 2 x multiply-accumulates); normalisation, activation and the
detection ops are not counted, as the usual counters do."""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch
import torch.nn as nn


def count_params(model: nn.Module, trainable_only: bool = False) -> int:
    return sum(p.numel() for p in model.parameters() if p.requires_grad or not trainable_only)


@torch.no_grad()
def count_flops(model: nn.Module, forward: Callable[[], object]) -> Tuple[int, Dict[str, int]]:
    """Runs `forward()` once (any callable that drives the model, e.g. ``lambda: model.extract(x)``) and
    returns (total FLOPs, FLOPs per top-level child)."""
    per_module: Dict[nn.Module, int] = {}

    def conv_hook(m: nn.Conv2d, inp, out):
        kh, kw = m.kernel_size
        macs = out.numel() * (m.in_channels // m.groups) * kh * kw
        per_module[m] = per_module.get(m, 0) + 2 * macs + (out.numel() if m.bias is not None else 0)

    def linear_hook(m: nn.Linear, inp, out):
        macs = out.numel() * m.in_features
        per_module[m] = per_module.get(m, 0) + 2 * macs + (out.numel() if m.bias is not None else 0)

    hooks = []
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            hooks.append(m.register_forward_hook(conv_hook))
        elif isinstance(m, nn.Linear):
            hooks.append(m.register_forward_hook(linear_hook))
    try:
        forward()
    finally:
        for h in hooks:
            h.remove()
    owner = {}
    for name, child in model.named_children():
        for m in child.modules():
            owner[m] = name
    by_child: Dict[str, int] = {}
    for m, f in per_module.items():
        by_child[owner.get(m, "")] = by_child.get(owner.get(m, ""), 0) + f
    return sum(per_module.values()), by_child


def model_summary(model: nn.Module, forward: Callable[[], object]) -> Dict[str, float]:
    """{'params_M', 'gflops'} as evaluation logs quote them."""
    flops, _ = count_flops(model, forward)
    return {"params_M": count_params(model) / 1e6, "gflops": flops / 1e9}
