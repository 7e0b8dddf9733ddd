"""Shared plumbing of the op modules: pointers, streams, workspaces and the kernel-only timing hooks bench.py uses."""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from .. import _lib
from .._lib import RoiLevels, Rows2, YoloLevels, YoloLossCfg, check

_F32, _I32 = torch.float32, torch.int32


def _stream(t: torch.Tensor) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _need(t: torch.Tensor, dtype, name: str, contiguous: bool = True) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.HdError(f"{name}: expected a tensor on a HIP device (there is no CPU path)")
    if t.dtype != dtype:
        raise _lib.HdError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        t = t.contiguous()
    return t


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


# ---- optional HIP-event timing of individual C-ABI launches (bench.py's roofline leg) ---------
# Events are recorded on the stream the kernel is launched on (torch's current stream, the one
# handed to the C ABI), immediately around the call; nothing is synchronised here.
_TIMED: dict = {}
_TIMED_BYTES: dict = {}
_TIMED_TAGS: dict = {}


def enable_timing(*names: str) -> None:
    _TIMED.clear()
    _TIMED_BYTES.clear()
    _TIMED_TAGS.clear()
    for n in names:
        _TIMED[n] = []
        _TIMED_BYTES[n] = []
        _TIMED_TAGS[n] = []


def disable_timing() -> None:
    _TIMED.clear()
    _TIMED_BYTES.clear()
    _TIMED_TAGS.clear()


def _note_bytes(name: str, nbytes: int, tag=None) -> None:
    """Algorithmic bytes (and a shape tag) of a timed launch, noted by ops that know their own operand sizes, in launch
    order — bench.py's step-weighted rooflines pair them with the launch's event times."""
    if name in _TIMED_BYTES:
        _TIMED_BYTES[name].append(int(nbytes))
        _TIMED_TAGS[name].append(tag)


def collect_bytes() -> dict:
    return {n: list(v) for n, v in _TIMED_BYTES.items()}


def collect_tags() -> dict:
    return {n: list(v) for n, v in _TIMED_TAGS.items()}


def collect_timing() -> dict:
    """After a device synchronise: {name: [ms, ...]} for every timed launch since enable_timing()."""
    return {n: [a.elapsed_time(b) for a, b in ev] for n, ev in _TIMED.items()}


class _timed:
    """Event pair around a whole C-ABI call (ops that launch several kernels)."""
    __slots__ = ("name", "start")

    def __init__(self, name: str):
        self.name = name if name in _TIMED else None

    def __enter__(self):
        if self.name is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if self.name is not None:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            _TIMED[self.name].append((self.start, end))
        return False


def _arm_kernel_timing():
    """Arms libhd_ops so that the NEXT launch of an hd_launch() kernel stamps a fresh event pair with the kernel's own
    begin / end (hd_profile_time_next_launch: the dispatch's timestamps, what a profiler reports).  Returns the pair."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()  # creates the hipEvent_t handles (torch creates them lazily at the first record)
    b.record()
    check(_lib.load().hd_profile_time_next_launch(C.c_void_p(a.cuda_event), C.c_void_p(b.cuda_event)), "hd_profile_time_next_launch")
    return a, b


class _ktimed:
    """Kernel-only timing of a single-kernel C-ABI call: the event pair is stamped by the launch itself."""
    __slots__ = ("name", "pair")

    def __init__(self, name: str):
        self.name = name if name in _TIMED else None

    def __enter__(self):
        if self.name is not None:
            self.pair = _arm_kernel_timing()
        return self

    def __exit__(self, *exc):
        if self.name is not None:
            _lib.load().hd_profile_time_next_launch(None, None)  # in case the call did not launch (empty input)
            _TIMED[self.name].append(self.pair)
        return False


def _ws(nbytes: int, device) -> Tuple[Optional[torch.Tensor], int]:
    if nbytes <= 0:
        return None, 0
    t = torch.empty((nbytes + 255) // 256 * 256, dtype=torch.uint8, device=device)
    return t, t.numel()


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]   # private helpers included: the sibling modules and the tests use them
