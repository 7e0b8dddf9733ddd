"""Random numbers of a training step, drawn OUTSIDE the captured region.

A step replayed from a hipGraph must not draw random numbers inside the graph: torch refreshes a captured generator's
Philox seed / offset only in `CUDAGraph.replay()`, and the multi-rank mode of train.Trainer launches the captured step as
raw graph SEGMENTS (ops.GraphSegments), which never passes through it — draws made inside the region would repeat the
values of the capture step for the whole run, identically on every rank.  So everything random in a step reads STATIC
tensors that the Trainer refills eagerly (ordinary launches on the same stream) right before the step, in eager, one-graph
and split-graph mode alike — the three modes therefore also consume the generator identically:
  * augment.Mosaic is two-phase: `draw()` (eager) -> `apply()` (inside the region, reads the drawn tensors);
  * layers.DropBlock2d asks the active `StepRandom` pool for its uniform field.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch


class StepRandom:
    """Pool of persistent uniform [0, 1) tensors keyed by (owner, shape).  `refresh()` refills every tensor with one launch
    each; `uniform()` hands out the tensor (creating and filling it at the first request, which must happen in an eager
    warm-up step — a request for a NEW tensor during stream capture raises)."""

    _active: Optional["StepRandom"] = None

    def __init__(self, device, generator: Optional[torch.Generator] = None):
        self.device = torch.device(device)
        self.generator = generator
        self._pool: Dict[Tuple[int, Tuple[int, ...]], torch.Tensor] = {}

    @classmethod
    def active(cls) -> Optional["StepRandom"]:
        return cls._active

    def __enter__(self):
        self._prev, StepRandom._active = StepRandom._active, self
        return self

    def __exit__(self, *exc):
        StepRandom._active = self._prev
        return False

    def uniform(self, owner, shape) -> torch.Tensor:
        key = (id(owner), tuple(int(s) for s in shape))
        t = self._pool.get(key)
        if t is None:
            if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("StepRandom: a new random tensor was requested during hipGraph capture; every random "
                                   "tensor of the step must have been requested once in an eager warm-up step")
            t = torch.empty(key[1], dtype=torch.float32, device=self.device).uniform_(generator=self.generator)
            self._pool[key] = t
        return t

    @torch.no_grad()
    def refresh(self) -> None:
        for t in self._pool.values():
            t.uniform_(generator=self.generator)

    def __len__(self):
        return len(self._pool)
