"""hipGraph hygiene: memset-node rewrite, capture markers, split instantiation (csrc/graph_fix.hip)."""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from .. import _lib
from .._lib import RoiLevels, Rows2, YoloLevels, YoloLossCfg, check
from ._core import *  # noqa: F401,F403
from ._core import _ptr, _stream  # noqa: F401

# ------------------------------------------------------------------------------- hipGraph hygiene
def graph_replace_memsets(raw_graph) -> Tuple[int, int]:
    """raw_graph: the hipGraph_t of a captured but not yet instantiated graph
    (``torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()``).  Returns (nodes, memsets replaced)."""
    lib = _lib.load()
    n, r = C.c_int(0), C.c_int(0)
    ptr = raw_graph if isinstance(raw_graph, int) else int(raw_graph)
    check(lib.hd_graph_replace_memsets(C.c_void_p(ptr), C.byref(n), C.byref(r)), "hd_graph_replace_memsets")
    return n.value, r.value


def graph_mark(slot: torch.Tensor, index: int) -> None:
    """Marks the current point of a stream capture (one-lane kernel storing `index` into the int32 tensor `slot`)."""
    check(_lib.load().hd_graph_mark(_ptr(slot), int(index), _stream(slot)), "hd_graph_mark")


class GraphSegments:
    """Executable segments of a captured graph cut at its markers (hd_graph_split_instantiate)."""

    def __init__(self, raw_graph, slot: torch.Tensor, max_segments: int):
        lib = _lib.load()
        execs = (C.c_void_p * max_segments)()
        marks = (C.c_int * max_segments)()
        n, nn = C.c_int(0), C.c_int(0)
        ptr = raw_graph if isinstance(raw_graph, int) else int(raw_graph)
        check(lib.hd_graph_split_instantiate(C.c_void_p(ptr), _ptr(slot), int(max_segments), execs, marks, C.byref(n),
                                             C.byref(nn)), "hd_graph_split_instantiate")
        self.execs = [execs[i] for i in range(n.value)]
        self.nodes = nn.value
        self._slot = slot  # keeps the marker slot alive for the replays

        self.marker_of = [marks[i] for i in range(n.value)]  # entry k valid for the marker segments only

    def launch(self, k: int, like: torch.Tensor) -> None:
        check(_lib.load().hd_graph_exec_launch(C.c_void_p(self.execs[k]), _stream(like)), "hd_graph_exec_launch")

    def __len__(self):
        return len(self.execs)

    def __del__(self):
        try:
            lib = _lib.load()
            for e in self.execs:
                lib.hd_graph_exec_destroy(C.c_void_p(e))
        except Exception:
            pass


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]   # private helpers included: the sibling modules and the tests use them
