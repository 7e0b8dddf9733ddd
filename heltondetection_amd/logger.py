"""Run logging (reference README.md:13-14 "logger + tensorboard record the training process").

One `RunLogger` per process: rank 0 prints human-readable lines to stdout and appends machine-readable events to
`<work_dir>/log.jsonl` (one JSON object per line: {"t": wall seconds, "event": ..., **fields}); scalars additionally
go to TensorBoard when `torch.utils.tensorboard` is importable (it is not in this image — the JSONL file carries the
same series, `scalars()` reads them back).  Other ranks stay silent unless `all_ranks=True`.
"""
from __future__ import annotations

import json
import os
import sys
import time
from typing import Any, Dict, Iterator, List, Optional


class RunLogger:
    def __init__(self, work_dir: Optional[str], rank: int = 0, stdout: bool = True, jsonl: bool = True,
                 tensorboard: Any = "auto", all_ranks: bool = False):
        self.rank = int(rank)
        self.active = self.rank == 0 or all_ranks
        self.work_dir = work_dir
        self._t0 = time.time()
        self._stdout = bool(stdout) and self.active
        self._fh = None
        self._tb = None
        if work_dir and self.active:
            os.makedirs(work_dir, exist_ok=True)
            if jsonl:
                name = "log.jsonl" if self.rank == 0 else f"log.rank{self.rank}.jsonl"
                self._fh = open(os.path.join(work_dir, name), "a", buffering=1)
            if tensorboard and self.rank == 0:
                try:
                    from torch.utils.tensorboard import SummaryWriter  # absent here: tensorboard is not installed
                    self._tb = SummaryWriter(os.path.join(work_dir, "tb"))
                except Exception:
                    if tensorboard is True:
                        raise
                    self._tb = None

    # ---- events ---------------------------------------------------------------------------------
    def event(self, event: str, **fields) -> None:
        if not self.active:
            return
        rec = {"t": round(time.time() - self._t0, 3), "event": event, **{k: _plain(v) for k, v in fields.items()}}
        if self._fh:
            self._fh.write(json.dumps(rec) + "\n")
        if self._stdout:
            body = " ".join(f"{k}={_fmt(v)}" for k, v in rec.items() if k not in ("t", "event"))
            print(f"[{time.strftime('%H:%M:%S')}] {event}: {body}", file=sys.stdout, flush=True)

    def scalars(self, step: int, **values) -> None:
        """Training-curve points: one event + one TensorBoard scalar per value."""
        self.event("scalars", step=int(step), **values)
        if self._tb is not None:
            for k, v in values.items():
                self._tb.add_scalar(k, float(v), int(step))

    def info(self, msg: str) -> None:
        self.event("info", msg=msg)

    def close(self) -> None:
        if self._fh:
            self._fh.close()
            self._fh = None
        if self._tb is not None:
            self._tb.close()
            self._tb = None


def _plain(v):
    if isinstance(v, (str, int, float, bool)) or v is None:
        return v
    if isinstance(v, dict):
        return {str(k): _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    if hasattr(v, "tolist"):  # tensors / arrays (0-dim -> scalar)
        return v.tolist()
    return repr(v)


def _fmt(v) -> str:
    if isinstance(v, float):
        return f"{v:.5g}"
    if isinstance(v, dict):
        return "{" + ", ".join(f"{k}: {_fmt(x)}" for k, x in v.items()) + "}"
    return str(v)


def read_events(path: str, event: Optional[str] = None) -> List[Dict[str, Any]]:
    out = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            rec = json.loads(line)
            if event is None or rec.get("event") == event:
                out.append(rec)
    return out


def read_scalars(path: str, name: str) -> Iterator[tuple]:
    for rec in read_events(path, "scalars"):
        if name in rec:
            yield rec["step"], rec[name]
