"""Shipped MIOpen find results for the library convolutions of the benchmark configurations.

The strided / 32-channel / head convolutions run on MIOpen.  With `torch.backends.cudnn.benchmark`
on, MIOpen searches its solvers at the first call of every convolution shape and keeps what its (noisy, first-call) timings
favoured: a fresh machine spends 25 s (YOLOv5s) to 3.5 min (FasterRCNN-R50 832^2) in that search and lands on a slightly
different set of library kernels every time (config 3: 49.1 - 50.7 ms/step over five fresh boxes).  `miopen_db/` holds the
user find-db / perf-db of one such search on an MI355X (text files, keyed by problem description and MIOpen version: entries
that do not match are ignored, new shapes are searched and appended as usual); `use_shipped_find_db()` points MIOpen at it
unless the caller already chose a MIOPEN_USER_DB_PATH.  Regenerate with `tools/finddb_probe.sh`."""
from __future__ import annotations

import os

_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def use_shipped_find_db() -> bool:
    """Call before the first convolution of the process.  -> True when the shipped directory was selected."""
    if os.environ.get("MIOPEN_USER_DB_PATH") or os.environ.get("HD_NO_SHIPPED_FIND_DB") == "1":
        return False
    if not os.path.isdir(_DB) or not os.access(_DB, os.W_OK):
        return False
    os.environ["MIOPEN_USER_DB_PATH"] = _DB
    return True
