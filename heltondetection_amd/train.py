"""One training step of the hot path: forward (MIOpen convs, bf16 autocast, channels-last) ->
HIP loss kernels -> backward with bucketed RCCL all-reduce overlapped -> fused AdamW.
LR schedule: linear warmup + cosine (reference README.md:13,61)."""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from .dist import GradReducer
from .layers import LowpWeights, ShadowConv2d


class LowpShadows:
    """bf16 shadows of the convolution weights (layers.ShadowConv2d): `refresh()` re-casts every master
    weight with one multi-tensor launch; the shadows are the autograd leaves of the step, so weight
    gradients arrive in bf16 and are converted once, while dist.GradReducer packs its buckets."""

    def __init__(self, model: torch.nn.Module, dtype: torch.dtype):
        self.masters, self.shadows = [], []
        for m in model.modules():
            if isinstance(m, ShadowConv2d) and m.weight.requires_grad and m.weight.is_cuda:
                lp = torch.empty_like(m.weight, dtype=dtype).requires_grad_(True)  # keeps channels-last strides
                m.weight_lp = lp
                self.masters.append(m.weight)
                self.shadows.append(lp)

    def grad_sources(self):
        return dict(zip(self.masters, self.shadows))

    @torch.no_grad()
    def refresh(self):
        if self.shadows:
            torch._foreach_copy_(self.shadows, self.masters)

    def release(self, model: torch.nn.Module):
        for m in model.modules():
            if isinstance(m, ShadowConv2d):
                m.weight_lp = None


def warmup_cosine(step: int, total: int, warmup: int, min_ratio: float = 0.01) -> float:
    if step < warmup:
        return (step + 1) / max(warmup, 1)
    t = (step - warmup) / max(total - warmup, 1)
    return min_ratio + (1 - min_ratio) * 0.5 * (1 + math.cos(math.pi * min(t, 1.0)))


class Trainer:
    """``use_graph=True`` captures forward + loss + backward + gradient packing into one hipGraph on
    static input buffers (the step issues ~1500 kernels; replay removes the host launch cost, measured
    13.3 -> 11.4 ms on YOLOv5s bs16).  The gradient all-reduce and the optimizer stay outside the
    graph, launched right after the replay, so 1-GPU and N-GPU runs execute the same structure."""

    def __init__(self, model: torch.nn.Module, lr: float = 1e-3, weight_decay: float = 5e-4, amp_dtype=torch.bfloat16,
                 total_steps: int = 1000, warmup_steps: int = 10, bucket_bytes: Optional[int] = None, device=None,
                 use_graph: bool = False, graph_warmup: int = 3, graph_sync: Optional[bool] = None,
                 lowp_weights: bool = True, augment=None, graph_split: bool = True, hd_optimizer: bool = True,
                 seed: Optional[int] = None):
        """bucket_bytes: None = dist.GradReducer's policy (buckets sized from the model's gradient bytes, small final bucket).
        seed: the step's random draws (Mosaic, DropBlock: rng.py) come from a per-RANK device generator seeded with
        `seed * 1000003 + rank`, so the ranks of a data-parallel run augment differently while each stays reproducible
        (with the default generator seeded identically on every rank, all ranks drew the same mosaics).  None:
        the default device generator, as before (one rank / tests that seed it themselves)."""
        self.model = model
        # augment: callable (images, gt_boxes, gt_cls, m_valid) -> same 4-tuple, run on the device at the head of every step
        # (e.g. augment.Mosaic).  A TWO-PHASE augment (has `draw(B, S, device) -> tensors` and `apply(images, gt_boxes,
        # gt_cls, m_valid, *tensors)`) has its random draws made eagerly before each step into static tensors and only
        # `apply` inside the step, which is what a hipGraph-replayed step needs (rng.py: a captured region must not draw
        # random numbers — split replay never refreshes the captured generator state).  A plain callable runs whole inside
        # the region and must be deterministic there.
        self.augment = augment
        self._aug_two_phase = augment is not None and hasattr(augment, "draw") and hasattr(augment, "apply")
        self._aug_params = None
        from .rng import StepRandom
        self.device = torch.device(device) if device is not None else next(model.parameters()).device
        self.generator = None
        if seed is not None and self.device.type == "cuda":
            import torch.distributed as _td
            rank = _td.get_rank() if _td.is_initialized() else 0
            self.generator = torch.Generator(device=self.device).manual_seed(int(seed) * 1000003 + rank)
            if self._aug_two_phase and getattr(augment, "generator", None) is None:
                augment.generator = self.generator
        self.rand = StepRandom(self.device, self.generator)
        self.amp_dtype = amp_dtype
        self.use_graph = bool(use_graph) and self.device.type == "cuda"
        from .layers import FusedBNAct
        if FusedBNAct.sync_group is not None:  # SyncBN all-reduces inside the layer: not capturable
            self.use_graph = False
        # bf16 weight shadows only make sense where autocast would otherwise cast per call
        self.shadows = None
        if lowp_weights and amp_dtype is not None and self.device.type == "cuda":
            self.shadows = LowpShadows(model, amp_dtype)
        import torch.distributed as tdist
        world = tdist.get_world_size() if tdist.is_initialized() else 1
        # Multi-rank hipGraph mode = SPLIT replay: the captured step is cut at the gradient-bucket boundaries
        # (ops.GraphSegments) and bucket k's all-reduce is launched right after segment k has been enqueued, so it runs
        # on RCCL's stream while segment k + 1 (the rest of backward) replays: the overlap north_star asks for, kept
        # under hipGraph replay.  graph_split=False (or one rank) replays one graph and launches the collectives
        # behind it (`defer`).
        # graph_split="force": split replay even on one rank (no collectives then; tests compare it with the one-graph replay)
        if os.environ.get("HD_GRAPH_SPLIT") == "0" and graph_split != "force":
            graph_split = False
        self.graph_split = bool(graph_split) and self.use_graph and (world > 1 or graph_split == "force")
        # one rank + the HIP optimizer: no packing at all, HdAdamW reads the gradients where autograd left them
        # (hipGraph mode only: an eager step is launch-bound on the host, where re-pointing the optimizer's descriptor
        # table at each step's fresh gradient tensors costs more than the two packing launches it saves)
        direct = (world == 1 and self.device.type == "cuda" and hd_optimizer and self.use_graph and not self.graph_split
                  and os.environ.get("HD_DIRECT_GRADS", "1") != "0")
        self.reducer = GradReducer(model, bucket_bytes=bucket_bytes,
                                   defer_collectives=self.use_graph and not self.graph_split,
                                   grad_sources=self.shadows.grad_sources() if self.shadows else None, direct=direct)
        # graph_sync: drain the stream after every replayed graph / segment before the collectives are launched.
        # Needed only for the gloo REHEARSAL backend (several ranks time-slicing one GPU): gloo's worker thread
        # blocks on a host-side stream synchronise for its D2H staging copy, and with two processes sharing the
        # device that wait was observed to return seconds late on alternate replays (DESIGN.md section 9,
        # tools/graph_event_probe.py).  RCCL has no host-side wait on this path (its stream waits on an event
        # device-side), so the default is off for it.
        # HD_GRAPH_SYNC=1 forces the drained form on any backend, HD_GRAPH_SPLIT=0 (read above) the one-graph replay with
        # the collectives behind it: the two conservative forms, one environment variable away if a node misbehaves.
        if graph_sync is None and os.environ.get("HD_GRAPH_SYNC") in ("0", "1"):
            graph_sync = os.environ["HD_GRAPH_SYNC"] == "1"
        if graph_sync is None:
            graph_sync = self.reducer.world > 1 and tdist.get_backend(self.reducer.group) != "nccl"
        self.graph_sync = bool(graph_sync)
        self._segments = None
        decay, no_decay = [], []
        for n, p in model.named_parameters():
            if not p.requires_grad:
                continue
            (no_decay if p.ndim <= 1 else decay).append(p)
        groups = [{"params": decay, "weight_decay": weight_decay}, {"params": no_decay, "weight_decay": 0.0}]
        if self.device.type == "cuda" and hd_optimizer:
            # one launch per parameter group (csrc/optim.hip) that also writes the bf16 weight shadows, so the per-step
            # re-cast of the shadows (LowpShadows.refresh) is needed only when the masters were changed from outside
            from .optim import HdAdamW
            self.opt = HdAdamW(groups, lr=lr, betas=(0.937, 0.999),
                               shadows=self.shadows.grad_sources() if self.shadows else None)
            self.opt.grad_from_shadows = self.reducer.direct
        else:
            self.opt = torch.optim.AdamW(groups, lr=lr, betas=(0.937, 0.999), fused=self.device.type == "cuda")
        self._shadows_dirty = True  # the shadows start uninitialised; weights_changed() marks them stale again
        # the 3x3 / stride 1 layers' data gradients run hd_conv3x3_fwd on flipped, transposed weights: one launch per step
        # rebuilds them all from the bf16 shadows (ops.FlippedWeights)
        self._flipped = None
        if self.shadows is not None and self.device.type == "cuda":
            from . import ops
            ws = [m.weight_lp for m in model.modules() if isinstance(m, ShadowConv2d) and m.weight_lp is not None
                  and m.kernel_size == (3, 3) and m.stride == (1, 1) and m.padding == (1, 1) and m.groups == 1
                  and ((m.in_channels % 64 == 0 and m.out_channels % 64 == 0) or (m.in_channels == m.out_channels == 32))]
            # ... and the 1x1 layers whose data gradient is a long reduction (Cout = 512): their plain transposes, same launch
            ws += [m.weight_lp for m in model.modules() if isinstance(m, ShadowConv2d) and m.weight_lp is not None
                   and m.kernel_size == (1, 1) and m.stride == (1, 1) and m.groups == 1 and m.bias is None
                   and m.out_channels == 512 and m.in_channels % 128 == 0 and m.in_channels <= 4096]
            fl = ops.FlippedWeights(ws)
            self._flipped = fl if len(fl) else None
        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lambda s: warmup_cosine(s, total_steps, warmup_steps))
        self._graph = None
        self._static = None
        self._static_out = None
        self._eager_steps = 0
        self._graph_warmup = graph_warmup
        self.graph_error = None
        self.graph_nodes = self.graph_memsets_replaced = 0
        # overlapped split replay is self-checking (see _verify_overlap); HD_GRAPH_VERIFY_OVERLAP=0 skips the two extra steps
        self.verify_overlap = os.environ.get("HD_GRAPH_VERIFY_OVERLAP", "1") != "0"
        self.graph_overlap_check = None
        self.graph_overlap_error = None

    def grad_of(self, p: torch.nn.Parameter):
        """The gradient the last step left for parameter `p` (float32 copy / tensor), or None.  With several ranks every
        parameter's .grad is a float32 view of its (averaged) bucket; with ONE rank and the HIP optimizer nothing is packed:
        a convolution weight's gradient then lives on its bfloat16 shadow and `p.grad` is None (GradReducer `direct`)."""
        if p.grad is not None:
            return p.grad
        src = self.shadows.grad_sources().get(p) if self.shadows is not None else None
        return src.grad.float() if src is not None and src.grad is not None else None

    def _fwd_bwd(self, images, gt_boxes, gt_cls, m_valid):
        self.reducer.prepare()
        if self.device.type == "cuda":
            from . import ops
            ops.bn_arena_reset(self.device)  # one kernel clears every fused-BN statistics slot
        use_amp = self.amp_dtype is not None and self.device.type == "cuda"
        if self.augment is not None:
            with torch.no_grad():
                if self._aug_two_phase:
                    images, gt_boxes, gt_cls, m_valid = self.augment.apply(images, gt_boxes, gt_cls, m_valid, *self._aug_params)
                else:
                    images, gt_boxes, gt_cls, m_valid = self.augment(images, gt_boxes, gt_cls, m_valid)
        if self.shadows is not None:
            if not getattr(self.opt, "writes_shadows", False):
                self.shadows.refresh()
            elif self._shadows_dirty and not torch.cuda.is_current_stream_capturing():
                # HdAdamW keeps the shadows current; they are stale only before the first step / after weights_changed()
                self.shadows.refresh()
                self._shadows_dirty = False
        LowpWeights.on = self.shadows is not None
        chain = None
        if self.device.type == "cuda":
            from . import ops
            chain = ops._WgradChain   # weight-gradient slice sums ride in the next weight-gradient launch (ops._WgradChain)
            chain.enabled = True
            if self._flipped is not None:
                self._flipped.refresh()
                self._flipped.__enter__()
        try:
            with self.rand, torch.autocast(device_type=self.device.type, dtype=self.amp_dtype, enabled=use_amp):
                loss, parts = self.model.forward_train(images, gt_boxes, gt_cls, m_valid)
            loss.backward()
            if chain is not None:
                ops.flush_wgrad()
            # buckets autograd left incomplete (parameters without a gradient this step) are packed here, inside
            # the captured region in hipGraph mode, so their packing copies replay with the graph
            self.reducer.pack_incomplete()
        except BaseException:
            if chain is not None:
                chain.reset()   # nothing of an aborted step may ride in a later launch (see ops._WgradChain.reset)
            raise
        finally:
            LowpWeights.on = False
            if chain is not None:
                chain.enabled = False
                if self._flipped is not None:
                    self._flipped.__exit__()
        return loss.detach(), parts

    def _capture(self, batch):
        self._static = [t.clone() for t in batch]
        side = torch.cuda.Stream(self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        hold = self.reducer._hold
        self.reducer._hold = True  # the warm-up pass computes gradients nobody uses: no collectives for it
        try:
            with torch.cuda.stream(side):  # private-pool warm-up as torch.cuda.graphs prescribes
                self._fwd_bwd(*self._static)
        finally:
            self.reducer._hold = hold
        torch.cuda.current_stream(self.device).wait_stream(side)
        g = torch.cuda.CUDAGraph(keep_graph=True)
        if self.graph_split:
            # no collective may be launched while capturing; every packed bucket leaves a marker in the graph
            self.reducer._hold = True
            self._mark_slot = torch.zeros(1, dtype=torch.int32, device=self.device)
        # thread_local: only this thread's calls are policed during capture (RCCL's watchdog thread
        # queries events concurrently in multi-GPU runs)
        try:
            if self.graph_split:
                self.reducer._mark = self._mark_slot
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self._fwd_bwd(*self._static)
        finally:
            self.reducer._mark = None
            self.reducer._hold = False
        # MIOpen's conv-backward solvers zero their accumulators with hipMemsetAsync, which the capture
        # turned into memset nodes; rewrite those into fill-kernel nodes before instantiating
        from . import ops
        self.graph_nodes, self.graph_memsets_replaced = ops.graph_replace_memsets(g.raw_cuda_graph())
        if self.graph_split:
            # `g` is never instantiated as a whole: it stays alive as the owner of the capture's memory pool
            self._segments = ops.GraphSegments(g.raw_cuda_graph(), self._mark_slot, len(self.reducer.buckets) + 1)
            marked = sorted(self._segments.marker_of[: len(self.reducer.buckets)])
            if marked != list(range(len(self.reducer.buckets))):
                raise RuntimeError(f"split capture: bucket markers {marked} do not cover the {len(self.reducer.buckets)} buckets")
        else:
            g.instantiate()
        self._graph, self._static_out = g, out

    def _replay_split(self, tap=None):
        """Segment k, then bucket k's all-reduce on RCCL's stream while segment k + 1 is already being enqueued.
        tap(bucket_index): called between the two (tests / _verify_overlap copy the packed local gradients there)."""
        seg, red = self._segments, self.reducer
        red.begin_split_replay()
        stream = torch.cuda.current_stream(self.device)
        n_marked = len(red.buckets)
        for k in range(len(seg)):
            seg.launch(k, self._mark_slot)
            if k < n_marked:
                if self.graph_sync:  # gloo rehearsal only, see __init__
                    stream.synchronize()
                if tap is not None:
                    tap(seg.marker_of[k])
                red.bucket_replayed(seg.marker_of[k])

    @torch.no_grad()
    def _draw_randoms(self, images) -> None:
        """Everything random in the coming step, drawn eagerly into static tensors (rng.py): the two-phase augment's
        parameters and every StepRandom field a module asked for.  Same calls in eager, one-graph and split-graph mode."""
        if self._aug_two_phase:
            drawn = self.augment.draw(images.shape[0], images.shape[2], images.device)
            if self._aug_params is None:
                self._aug_params = [t.clone() for t in drawn]
            else:
                for dst, src in zip(self._aug_params, drawn):
                    dst.copy_(src)
        self.rand.refresh()

    def _verify_overlap(self) -> None:
        """Multi-rank split replay launches bucket k's all-reduce without draining the stream (graph_sync off).  Before the
        first real replayed step this replays the captured step once, overlapped, with a stream-ordered copy of every
        bucket's LOCAL packed gradients taken between segment k and its collective (`tap`); after a full synchronise the
        copies are all-reduced again, drained.  Same inputs, same collective algorithm: the two results are equal (to
        rounding) iff every overlapped collective consumed the finished bucket of this very replay.  Comparing two whole
        replays instead would not work: at fixed weights the bf16 step's gradients differ by 0.2-0.5 (relative L2) from one
        run to the next (tools/determinism_probe.py), drained or not.  On a mismatch (any rank) the
        trainer falls back to the drained form for the rest of the run and says so in `graph_overlap_error`."""
        import torch.distributed as tdist
        red = self.reducer
        local = {}

        def tap(index: int):
            local[index] = red.buckets[index].flat.clone()   # enqueued behind segment k, ahead of the all-reduce's event

        self._replay_split(tap)
        red.finish()
        torch.cuda.synchronize(self.device)
        op = tdist.ReduceOp.AVG if red._avg_native else tdist.ReduceOp.SUM
        worst = 0.0
        for index, b in enumerate(red.buckets):
            ref = local[index]
            tdist.all_reduce(ref, op=op, group=red.group)
            if not red._avg_native:
                ref.div_(red.world)
            den = float(ref.double().norm())
            err = float((b.flat.double() - ref.double()).norm()) / den if den > 0 else float(b.flat.abs().max())
            worst = max(worst, err if err == err else float("inf"))
        bad = torch.tensor([worst], dtype=torch.float32, device=self.device)
        tdist.all_reduce(bad, op=tdist.ReduceOp.MAX, group=red.group)
        self.graph_overlap_check = float(bad.item())
        if not (self.graph_overlap_check < 1e-5):
            self.graph_overlap_error = f"an overlapped all-reduce did not see its finished bucket (rel. L2 {self.graph_overlap_check:.3g} " \
                                       "against the drained all-reduce of the same packed gradients); falling back to graph_sync"
            self.graph_sync = True

    def weights_changed(self) -> None:
        """Tell the trainer that the float32 master weights were modified outside `step()` (a checkpoint was loaded):
        the bf16 shadows are re-cast before the next forward."""
        self._shadows_dirty = True

    def step(self, images, gt_boxes, gt_cls, m_valid):
        batch = (images, gt_boxes, gt_cls, m_valid)
        self._draw_randoms(images)
        if self._shadows_dirty:
            if self.shadows is not None:
                self.shadows.refresh()  # an ordinary launch ahead of the captured region / the replay
            self._shadows_dirty = False
        if self.use_graph and self._graph is None and self._eager_steps >= self._graph_warmup:
            err = None
            try:
                self._capture(batch)
            except Exception as e:  # stay correct: fall back to eager and remember why
                err = repr(e)
            if self.reducer.world > 1:
                # every rank must take the same branch: a rank that fell back to eager would launch its collectives
                # from the autograd hooks while the others launch theirs after the replay
                import torch.distributed as dist
                ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=self.device)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.reducer.group)
                if int(ok.item()) == 0 and err is None:
                    err = "hipGraph capture failed on another rank"
            if err is not None:
                self.graph_error = err
                self.use_graph = False
                self.reducer.defer = False
                self._graph = self._static = self._static_out = self._segments = None
                from . import ops
                ops._WgradChain.reset()   # a capture that failed on ANOTHER rank leaves this rank's chain state too
                # eager steps allocate fresh gradient tensors every step: back to the packed buckets, whose addresses never
                # change (the optimizer's descriptor table is then built once; see optim.HdAdamW._repoint)
                if self.reducer.direct:
                    self.reducer.direct = False
                    if hasattr(self.opt, "grad_from_shadows"):
                        self.opt.grad_from_shadows = False
                torch.cuda.synchronize(self.device)
            elif self._segments is not None and self.reducer.world > 1 and not self.graph_sync and self.verify_overlap:
                self._verify_overlap()
        if self._graph is not None:
            for dst, src in zip(self._static, batch):
                dst.copy_(src, non_blocking=True)
            if self._segments is not None:
                self._replay_split()
            else:
                self._graph.replay()
                if self.graph_sync:  # gloo rehearsal only, see __init__
                    torch.cuda.current_stream(self.device).synchronize()
            loss, parts = self._static_out
        else:
            loss, parts = self._fwd_bwd(*batch)
            self._eager_steps += 1
        self.reducer.finish()
        self.opt.step()
        self.sched.step()
        return loss, parts
