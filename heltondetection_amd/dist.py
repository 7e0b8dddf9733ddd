"""Data-parallel plumbing: one process per GPU, RCCL over xGMI (reference README.md:29 "DDP").

The hot path shards by image: every detection op is per-image, so the only
exchange is the gradient all-reduce.  GradReducer does that exchange MI355X-first rather than
by wrapping torch's DDP:
  * gradients are packed into flat fp32 buckets sized from the model (plan_buckets: ~8-20 per step — xGMI is 7
    point-to-point links per GPU and a ring all-reduce is bound by one link, so a handful of large collectives beat
    many small ones; the first bucket is smaller so communication starts early in backward, and the LAST one, whose
    all-reduce nothing can hide, is at most 2 MB);
  * a bucket's all-reduce is launched (async, RCCL's own stream) the moment its last gradient
    has been produced by autograd, so it overlaps the rest of backward;
  * one multi-tensor copy per bucket packs the gradients, and parameters' .grad are re-pointed
    at views of the reduced flat buffer, so the optimizer reads the averaged gradients in place.
Works unchanged on the gloo backend (CPU tests, world_size 2).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple[int, int, int]:
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # HD_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal on a 1-GPU box)
            backend = os.environ.get("HD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local, world


class _Bucket:
    __slots__ = ("params", "flat", "views", "pending", "work", "packed")

    def __init__(self, params: List[torch.nn.Parameter]):
        self.params = params
        n = sum(p.numel() for p in params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=params[0].device)
        self.views, off = [], 0
        for p in params:
            # same strides as the parameter (channels-last conv weights stay channels-last), so
            # multi-tensor optimizer kernels take their fast path on (param, grad) pairs
            self.views.append(self.flat.as_strided(p.shape, p.stride(), off))
            off += p.numel()
        self.pending = len(params)
        self.work = None
        self.packed = False


def plan_buckets(sizes: List[int], bucket_bytes: Optional[int] = None, first_bucket_bytes: Optional[int] = None,
                 tail_bytes: Optional[int] = None) -> List[List[int]]:
    """Partition of the gradients, given as byte sizes IN ARRIVAL ORDER (reverse registration order: the head's gradients
    first, the stem's last), into consecutive buckets.  -> list of index lists.

    Sized by time to completion, not by a byte constant (32 MB buckets put 84 % of YOLOv5s's 29 MB into one
    bucket that is complete only when backward ends — nothing left to overlap it with):
      * `bucket_bytes` (default: total / 16, at least 4 MB): YOLOv5s 29 MB -> 8 buckets, YOLOv5l 186 MB -> 12 MB buckets,
        ResNet50-PAFPN 164 MB -> 10 MB.  A bucket's all-reduce overlaps the backward segments behind it; 8-20 collectives
        per step keep the per-collective latency (tens of microseconds on 8 ranks) small against the step.
      * `first_bucket_bytes` (default: a quarter of that): communication starts early.
      * `tail_bytes` (default 2 MB, at most half a bucket): the LAST bucket — the gradients that arrive last (stem, first
        stage) and whose all-reduce nothing can hide — is kept this small by closing the previous bucket early."""
    total = sum(sizes)
    if bucket_bytes is None:
        bucket_bytes = max(4 << 20, total // 16)
    bucket_bytes = max(int(bucket_bytes), 1)
    first = int(first_bucket_bytes) if first_bucket_bytes is not None else max(bucket_bytes // 4, 1)
    first = min(first, bucket_bytes)
    tail = int(tail_bytes) if tail_bytes is not None else min(2 << 20, max(bucket_bytes // 2, 1))
    # the tail: the longest suffix of at most `tail` bytes (at least the last gradient, at most all but one)
    n = len(sizes)
    cut, acc = n, 0
    while cut > 1 and acc + sizes[cut - 1] <= tail:
        cut -= 1
        acc += sizes[cut]
    if cut == n and n > 1:
        cut = n - 1          # the last gradient alone exceeds the tail budget: it is the tail
    if n <= 1 or total <= tail:
        cut = 0 if total <= tail else cut
    out, cur, cur_bytes, cap = [], [], 0, first
    for i in range(cut):
        cur.append(i)
        cur_bytes += sizes[i]
        if cur_bytes >= cap:
            out.append(cur)
            cur, cur_bytes, cap = [], 0, bucket_bytes
    if cur:
        out.append(cur)
    if cut < n:
        out.append(list(range(cut, n)))
    return out


class GradReducer:
    def __init__(self, module: torch.nn.Module, bucket_bytes: Optional[int] = None, first_bucket_bytes: Optional[int] = None,
                 process_group=None, broadcast_params: bool = True, defer_collectives: bool = False,
                 grad_sources: Optional[Dict[torch.nn.Parameter, torch.Tensor]] = None, direct: bool = False,
                 tail_bytes: Optional[int] = None):
        self.module = module
        # grad_sources: parameter -> the leaf tensor that receives its gradient instead of the parameter
        # (a bf16 weight shadow, train.LowpShadows); the bucket copy converts to fp32 while packing
        self._src = dict(grad_sources or {})
        # defer_collectives: hooks only pack gradients; finish() launches the all-reduces.  Used when
        # forward+backward are replayed from a hipGraph (collectives stay outside the captured region).
        self.defer = defer_collectives
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # direct (one rank only): nothing is exchanged, so nothing is packed either — gradients stay where autograd left them
        # (float32 on the parameters, bfloat16 on the weight shadows) and the optimizer reads them there (optim.HdAdamW with
        # grad_from_shadows): two multi-tensor copies of all gradients per step less.  Parameters without a gradient are then
        # SKIPPED by the optimizer (torch.optim semantics) instead of receiving zeros.
        self.direct = bool(direct) and self.world == 1
        params = [p for p in module.parameters() if p.requires_grad]
        if self.world > 1 and broadcast_params:
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t.data, src=0, group=process_group)
        # gradients become ready roughly in reverse registration order
        order = list(reversed(params))
        plan = plan_buckets([p.numel() * 4 for p in order], bucket_bytes, first_bucket_bytes, tail_bytes)
        self.buckets: List[_Bucket] = [_Bucket([order[i] for i in idx]) for idx in plan]
        self._bucket_of = {}
        for b in self.buckets:
            for p in b.params:
                self._bucket_of[p] = b
        self._avg_native = self.world > 1 and dist.get_backend(process_group) == "nccl"
        self._next = 0  # index of the next bucket whose collective may be launched (strict bucket order)
        # hipGraph capture with split replay (train.Trainer): `_hold` keeps every collective back while the step is
        # being captured, `_mark` is the int32 slot the bucket markers are written to (ops.graph_mark)
        self._hold = False
        self._mark = None
        self._index_of = {id(b): i for i, b in enumerate(self.buckets)}
        self._hooks = []
        for p in params:
            src = self._src.get(p)
            if src is None:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            else:
                self._hooks.append(src.register_post_accumulate_grad_hook(lambda _t, p=p: self._on_grad(p)))

    # autograd calls this once per parameter per backward, right after .grad is final
    def _on_grad(self, p: torch.nn.Parameter):
        b = self._bucket_of[p]
        b.pending -= 1
        if b.pending == 0:
            self._pack(b)
            self._flush()

    def _grad_of(self, p):
        src = self._src.get(p)
        return p.grad if src is None else src.grad

    def _pack(self, b: _Bucket):
        """Copies the bucket's gradients into its flat buffer (one multi-tensor copy per gradient dtype: fp32 from
        parameters, bf16 from weight shadows — _foreach_copy_ keeps its single-launch path only for
        dtype-homogeneous lists), zero-fills the slots of parameters that received no gradient this step, and
        re-points .grad at the views.  Never launches a collective: that is _flush()'s job, in bucket order."""
        if self.direct:
            b.pending = 0
            b.packed = True
            return
        if b.flat.is_cuda:
            from . import ops
            ops.flush_wgrad()   # a chained weight gradient is complete only once its slice sum has been launched
        groups: Dict[torch.dtype, Tuple[list, list]] = {}
        for p, v in zip(b.params, b.views):
            g = self._grad_of(p)
            if g is None:
                v.zero_()  # unused parameter: contributes zeros to the average
            elif g.data_ptr() != v.data_ptr():
                vs, gs = groups.setdefault(g.dtype, ([], []))
                vs.append(v)
                gs.append(g)
        for vs, gs in groups.values():
            torch._foreach_copy_(vs, gs)
        for p, v in zip(b.params, b.views):
            p.grad = v
            src = self._src.get(p)
            if src is not None:
                src.grad = None
        b.pending = 0
        b.packed = True
        if self._mark is not None:
            # bucket boundary of the captured step: the split replay launches this bucket's all-reduce right after
            # the segment that ends here (one-lane marker kernel; found again by hd_graph_split_instantiate)
            from . import ops
            ops.graph_mark(self._mark, self._index_of[id(b)])

    def _flush(self, force: bool = False):
        """Launches the all-reduce of every packed bucket whose predecessors have been launched, strictly in
        bucket-index order: gradient-readiness order can differ between ranks (a bucket left incomplete on one
        rank is packed last there), and collectives issued in different orders on different ranks hang or mix
        up buffers on RCCL.  Deferred mode (hipGraph replay) launches nothing until finish() forces it."""
        if self.world <= 1 or self._hold or (self.defer and not force):
            return
        op = dist.ReduceOp.AVG if self._avg_native else dist.ReduceOp.SUM
        while self._next < len(self.buckets) and self.buckets[self._next].packed:
            b = self.buckets[self._next]
            b.work = dist.all_reduce(b.flat, op=op, group=self.group, async_op=True)
            self._next += 1

    def prepare(self):
        """Call before backward: drop last step's gradient views so autograd hands over fresh grads."""
        self._next = 0
        for b in self.buckets:
            b.pending = len(b.params)
            b.work = None
            b.packed = False
            for p in b.params:
                p.grad = None
                src = self._src.get(p)
                if src is not None:
                    src.grad = None

    def begin_split_replay(self):
        """Split hipGraph replay: the segments repack the buckets on the device; nothing is packed or launched yet."""
        self._next = 0
        for b in self.buckets:
            b.packed = False
            b.work = None

    def bucket_replayed(self, index: int):
        """The replayed segment that packs bucket `index` has been enqueued: its all-reduce may go (in bucket order)."""
        self.buckets[index].packed = True
        self._flush(force=True)

    def pack_incomplete(self):
        """Call right after backward (inside the captured region in hipGraph mode, so the packing copies of
        buckets with unused parameters are graph nodes too and replay every step): packs the buckets autograd
        did not complete.  Idempotent."""
        for b in self.buckets:
            if not b.packed:
                self._pack(b)
        self._flush()

    def finish(self):
        """Call after backward: every bucket is reduced exactly once, in bucket order; then wait for all of
        them (the current stream then depends on RCCL's)."""
        self.pack_incomplete()
        if self.defer:
            # replay mode: the hooks did not run this step (the graph replayed their packing copies), so the
            # launch cursor is wherever the capture step left it
            self._next = 0
        self._flush(force=True)
        for b in self.buckets:
            if b.work is not None:
                b.work.wait()
                if not self._avg_native:
                    b.flat.div_(self.world)
                b.work = None

    def grad_bytes(self) -> int:
        return sum(b.flat.numel() * 4 for b in self.buckets)


def shard_batch(n_items: int, rank: int, world: int) -> slice:
    """Contiguous shard of a global batch for this rank (weak scaling keeps n_items/world fixed)."""
    per = n_items // world
    return slice(rank * per, (rank + 1) * per)
