"""One training step of the hot path: forward (MIOpen convs, bf16 autocast, channels-last) ->
HIP loss kernels -> backward with bucketed RCCL all-reduce overlapped -> fused AdamW.
LR schedule: linear warmup + cosine (reference README.md:13,61)."""
from __future__ import annotations

import math
from typing import Optional

import torch

from .dist import GradReducer


def warmup_cosine(step: int, total: int, warmup: int, min_ratio: float = 0.01) -> float:
    if step < warmup:
        return (step + 1) / max(warmup, 1)
    t = (step - warmup) / max(total - warmup, 1)
    return min_ratio + (1 - min_ratio) * 0.5 * (1 + math.cos(math.pi * min(t, 1.0)))


class Trainer:
    def __init__(self, model: torch.nn.Module, lr: float = 1e-3, weight_decay: float = 5e-4, amp_dtype=torch.bfloat16,
                 total_steps: int = 1000, warmup_steps: int = 10, bucket_bytes: int = 32 << 20, device=None):
        self.model = model
        self.device = device or next(model.parameters()).device
        self.amp_dtype = amp_dtype
        self.reducer = GradReducer(model, bucket_bytes=bucket_bytes)
        decay, no_decay = [], []
        for n, p in model.named_parameters():
            if not p.requires_grad:
                continue
            (no_decay if p.ndim <= 1 else decay).append(p)
        fused = self.device.type == "cuda"
        self.opt = torch.optim.AdamW([{"params": decay, "weight_decay": weight_decay},
                                      {"params": no_decay, "weight_decay": 0.0}], lr=lr, betas=(0.937, 0.999),
                                     fused=fused)
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda s: warmup_cosine(s, total_steps, warmup_steps))

    def step(self, images, gt_boxes, gt_cls, m_valid):
        self.reducer.prepare()
        use_amp = self.amp_dtype is not None and self.device.type == "cuda"
        with torch.autocast(device_type=self.device.type, dtype=self.amp_dtype, enabled=use_amp):
            loss, parts = self.model.forward_train(images, gt_boxes, gt_cls, m_valid)
        loss.backward()
        self.reducer.finish()
        self.opt.step()
        self.sched.step()
        return loss.detach(), parts
