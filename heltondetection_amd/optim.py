"""AdamW on the one-launch HIP kernel of csrc/optim.hip (`hd_adamw_multi`).

`HdAdamW` is a torch.optim.AdamW — same constructor arguments, param groups, state layout (`step`, `exp_avg`,
`exp_avg_sq` per parameter), `state_dict()` / `load_state_dict()` and LR-scheduler behaviour — whose `step()` runs one
kernel per parameter group over a device-resident descriptor table instead of PyTorch's multi-tensor-apply launches, and
that also writes the bf16 shadows of the convolution weights (train.LowpShadows) in the same pass, so the per-step
re-cast of the shadows disappears.  Reference: README.md:13 ("training ... optimizer"); the optimizer of the reference is
not in the snapshot, AdamW with YOLOv5's betas is this repo's choice (train.Trainer)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

from . import _lib

# elements per workgroup of hd_adamw_multi.  4096: 1760 workgroups for YOLOv5s' 7.2 M parameters — at 16384 the 440 workgroups kept
# ~7 waves per CU in flight and the pass ran at 4 TB/s (50 us); 4096 / 2048: 5.293 -> 5.276 ms per step, 8192: 5.288
_CHUNK = 4096   # elements per workgroup of the AdamW pass (round 4 sweep: 5.293 -> 5.276 ms per step against 1024)


class _Desc(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("shadow", C.c_void_p),
                ("n", C.c_int64), ("weight_decay", C.c_float), ("g_dtype", C.c_int32)]


class HdAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 shadows: Optional[Dict[torch.Tensor, torch.Tensor]] = None):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, fused=False)
        self.shadows = dict(shadows or {})   # float32 master parameter -> its bf16 shadow (same memory order)
        self._tables = None                  # per group: (descriptor tensor, chunk_tensor, chunk_start, n_chunks, key)
        # True: a parameter whose own .grad is None takes the gradient of its shadow (bfloat16) where autograd left it — the
        # one-rank form of train.Trainer, where no exchange packs the gradients into float32 buckets first
        self.grad_from_shadows = False

    def _grad(self, p):
        if p.grad is not None or not self.grad_from_shadows:
            return p.grad
        sh = self.shadows.get(p)
        return sh.grad if sh is not None else None

    @property
    def writes_shadows(self) -> bool:
        return bool(self.shadows)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = None

    def _key(self, group):
        return tuple((p.data_ptr(), g.data_ptr()) for p, g in ((p, self._grad(p)) for p in group["params"]) if g is not None)

    def _step_of(self, p) -> float:
        st = self.state.get(p)
        return float(st["step"]) if st else 0.0

    def _build(self, group):
        """-> list of launch tables, one per distinct step count among the group's parameters (torch.optim.AdamW keeps a
        step per parameter; they differ when a parameter started receiving gradients later or after a loaded state_dict with
        mixed counts, and the bias corrections are per step count).  Parameters advance together afterwards, so the
        partition stays valid until the gradient set changes (the key)."""
        ps_all = [p for p in group["params"] if self._grad(p) is not None]
        if not ps_all:
            return None
        by_step = {}
        for p in ps_all:
            by_step.setdefault(self._step_of(p), []).append(p)
        return [self._build_one(group, ps) for _, ps in sorted(by_step.items())]

    def _build_one(self, group, ps):
        dev = ps[0].device
        descs, ct, cs = (_Desc * len(ps))(), [], []
        for i, p in enumerate(ps):
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            g, m, v, sh = self._grad(p), st["exp_avg"], st["exp_avg_sq"], self.shadows.get(p)
            for t, what in ((p, "parameter"), (g, "gradient"), (m, "exp_avg"), (v, "exp_avg_sq")):
                ok_dtype = t.dtype == torch.float32 or (what == "gradient" and t.dtype == torch.bfloat16)
                if not ok_dtype or t.device != dev or not _same_order(t, p) or not _dense(t):
                    raise _lib.HdError(f"HdAdamW: {what} must be a dense float32 tensor (gradients: or bfloat16) in the "
                                       f"parameter's memory order")
            if sh is not None and (sh.dtype != torch.bfloat16 or not _same_order(sh, p)):
                raise _lib.HdError("HdAdamW: a shadow must be a bfloat16 tensor of the parameter's shape and memory order")
            descs[i] = _Desc(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), sh.data_ptr() if sh is not None else None,
                             p.numel(), float(group["weight_decay"]), 1 if g.dtype == torch.bfloat16 else 0)
            for start in range(0, p.numel(), _CHUNK):
                ct.append(i)
                cs.append(start)
        # the descriptors live in a PINNED host buffer owned by this table and travel with a non-blocking copy: when gradient
        # addresses change (one-rank direct mode: autograd allocates the gradients anew, and inside a hipGraph capture they
        # come from the graph's pool) only the pointers are patched and re-sent (`_repoint`) — cheap in eager mode, and legal
        # inside a capture, where the copy becomes a graph node whose source stays alive as long as the optimizer
        # TWO host buffers, used alternately, each with the event of its last copy: a non-blocking copy reads its pinned source
        # when the stream gets there, so a buffer may be patched again only after that event (the host can run
        # more than a step ahead of the device in eager steps, and step N's kernel would read step N + 1's gradient addresses)
        hosts = [torch.empty(C.sizeof(descs), dtype=torch.uint8).pin_memory() if dev.type == "cuda" else
                 torch.empty(C.sizeof(descs), dtype=torch.uint8) for _ in range(2)]
        for h in hosts:
            C.memmove(h.data_ptr(), C.addressof(descs), C.sizeof(descs))
        raw = torch.empty(C.sizeof(descs), dtype=torch.uint8, device=dev)
        host = _HostPair(hosts)
        host.send(raw)
        return [raw, torch.tensor(ct, dtype=torch.int32, device=dev), torch.tensor(cs, dtype=torch.int64, device=dev), len(ct),
                self._key(group), ps, host]

    def _repoint(self, group, tabs) -> bool:
        """Same parameters, new gradient tensors: patch the descriptors' gradient pointers in place.  -> False when the set of
        parameters with a gradient (or a gradient's dtype / layout) changed and the tables have to be rebuilt."""
        have = [p for p in group["params"] if self._grad(p) is not None]
        if sum(len(t[5]) for t in tabs) != len(have) or {id(p) for t in tabs for p in t[5]} != {id(p) for p in have}:
            return False
        key = self._key(group)
        for t in tabs:
            raw, ps, host = t[0], t[5], t[6]
            descs = (_Desc * len(ps)).from_address(host.next_buffer().data_ptr())
            for i, p in enumerate(ps):
                g = self._grad(p)
                # dtype and element count are re-checked; the memory order of a gradient is a property of the op that
                # produced it and was validated when the table was built
                if (1 if g.dtype == torch.bfloat16 else 0) != descs[i].g_dtype or g.numel() != descs[i].n:
                    return False
                descs[i].g = g.data_ptr()
            host.send(raw)
            t[4] = key
        return True

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        lib = _lib.load()
        if self._tables is None:
            self._tables = [None] * len(self.param_groups)
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):
                raise _lib.HdError("HdAdamW: amsgrad / maximize are not implemented")
            tabs = self._tables[gi]
            if tabs is None:                                     # first step, loaded state
                tabs = self._tables[gi] = self._build(group)
            elif tabs[0][4] != self._key(group) and not self._repoint(group, tabs):   # new gradient tensors / a new set
                tabs = self._tables[gi] = self._build(group)
            if tabs is None:
                continue
            b1, b2 = group["betas"]
            for raw, ct, cs, n_chunks, _, ps, _host in tabs:   # one launch per distinct step count (normally one)
                t = float(self.state[ps[0]]["step"]) + 1.0
                for p in ps:
                    self.state[p]["step"] += 1
                stream = torch.cuda.current_stream(raw.device).cuda_stream
                rc = lib.hd_adamw_multi(C.c_void_p(raw.data_ptr()), C.c_void_p(ct.data_ptr()), C.c_void_p(cs.data_ptr()), n_chunks,
                                        _CHUNK, float(group["lr"]), float(b1), float(b2), float(group["eps"]), 1.0 - b1 ** t,
                                        1.0 - b2 ** t, C.c_void_p(stream))
                if rc != 0:
                    raise _lib.HdError(f"hd_adamw_multi failed ({rc})")
        return loss


class _HostPair:
    """The descriptor table's two pinned host copies.  `next_buffer()` returns the one that is NOT the source of the latest
    copy, after waiting for its own last copy to have executed; `send()` copies it to the device table and records the event.
    Inside a hipGraph capture nothing is waited for or recorded (an event recorded there belongs to the graph; the copy node
    reads its source at replay time, and that source is left alone from then on because the capture's gradient addresses
    never change again)."""

    def __init__(self, hosts):
        self.hosts, self.events, self.cur = hosts, [None, None], 0

    def next_buffer(self) -> torch.Tensor:
        nxt = 1 - self.cur
        ev = self.events[nxt]
        if ev is not None:
            ev.synchronize()
            self.events[nxt] = None
        src, dst = self.hosts[self.cur], self.hosts[nxt]
        if src.data_ptr() != dst.data_ptr():
            dst.copy_(src)      # start from the latest table: only the gradient pointers are patched
        self.cur = nxt
        return dst

    def send(self, raw: torch.Tensor) -> None:
        raw.copy_(self.hosts[self.cur], non_blocking=True)
        if raw.is_cuda and not torch.cuda.is_current_stream_capturing():
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(raw.device))
            self.events[self.cur] = ev


def _same_order(t: torch.Tensor, p: torch.Tensor) -> bool:
    """Same shape and the same strides on every dimension that has more than one element (the stride of a size-1 dimension is
    arbitrary: a [N, K, 1, 1] gradient viewed from a [N, K] matrix and its channels-last parameter share one memory order)."""
    return t.shape == p.shape and all(a == b for a, b, n in zip(t.stride(), p.stride(), t.shape) if n > 1)


def _dense(t: torch.Tensor) -> bool:
    """Non-overlapping and dense: the numel() elements occupy numel() consecutive storage slots (any dimension order)."""
    if t.numel() <= 1:
        return True
    dims = sorted(((st, sz) for st, sz in zip(t.stride(), t.shape) if sz > 1))
    expect = 1
    for st, sz in dims:
        if st != expect:
            return False
        expect *= sz
    return True
