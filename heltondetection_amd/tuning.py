"""Shipped MIOpen find results for the library convolutions of the benchmark configurations.

The strided / 32-channel / head convolutions run on MIOpen.  With `torch.backends.cudnn.benchmark`
on, MIOpen searches its solvers at the first call of every convolution shape and keeps what its (noisy, first-call) timings
favoured: a fresh machine spends 25 s (YOLOv5s) to 3.5 min (FasterRCNN-R50 832^2) in that search and lands on a slightly
different set of library kernels every time (config 3: 49.1 - 50.7 ms/step over five fresh boxes).  `miopen_db/` holds the
user find-db / perf-db of one such search on an MI355X (text files, keyed by problem description and MIOpen version: entries
that do not match are ignored, new shapes are searched and appended as usual); `use_shipped_find_db()` points MIOpen at it
unless the caller already chose a MIOPEN_USER_DB_PATH.  Regenerate with `tools/finddb_probe.sh`."""
from __future__ import annotations

import atexit
import os
import shutil
import tempfile

_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")
_private = None


def use_shipped_find_db() -> bool:
    """Call before the first convolution of the process.  -> True when the shipped database was selected.

    MIOpen treats MIOPEN_USER_DB_PATH as WRITABLE: every shape it has not seen is searched and appended, by every process
    that points there.  The shipped directory is therefore never handed to it (a tracked tree would go dirty, later runs
    would depend on earlier ones, and the N ranks of one node would append to the same two text files concurrently): each
    process gets a PRIVATE COPY in a fresh temporary directory, removed at exit.  `HD_FIND_DB_DIR=<dir>` names a persistent
    private directory instead (seeded from the shipped files once; `<dir>/rank<k>` per rank under torch.distributed.run) —
    `tools/finddb_probe.sh` uses it to regenerate the shipped files."""
    global _private
    if _private is not None and os.environ.get("MIOPEN_USER_DB_PATH") == _private:
        return True     # a second call in the same process
    if os.environ.get("MIOPEN_USER_DB_PATH") or os.environ.get("HD_NO_SHIPPED_FIND_DB") == "1":
        return False
    if not os.path.isdir(_DB):
        return False
    keep = os.environ.get("HD_FIND_DB_DIR")
    try:
        if keep:
            rank = os.environ.get("LOCAL_RANK") or os.environ.get("RANK")
            dst = os.path.join(keep, f"rank{rank}") if rank is not None and int(os.environ.get("WORLD_SIZE", "1")) > 1 else keep
            os.makedirs(dst, exist_ok=True)
        else:
            dst = tempfile.mkdtemp(prefix="hd_miopen_db_")
            atexit.register(shutil.rmtree, dst, True)
        for f in os.listdir(_DB):
            src = os.path.join(_DB, f)
            if os.path.isfile(src) and not f.endswith(".time") and not os.path.exists(os.path.join(dst, f)):
                shutil.copyfile(src, os.path.join(dst, f))
    except OSError:
        return False
    _private = dst
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return True


def find_db_dir():
    """The private directory this process handed to MIOpen (None when the shipped database is not in use)."""
    return _private
