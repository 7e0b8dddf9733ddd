"""Backbone / neck ops on channels-last rows: in-place concatenation, max-pools, fused BatchNorm + activation, the 1x1 / 3x3 /
stem / Detect convolutions with their gradients and the weight-gradient slice-sum chain (csrc/bn_act.hip, pool.hip, upcat.hip,
pointwise_conv.hip, pointwise_bwd.hip, conv3x3.hip, conv3x3_wgrad.hip, stem_conv.hip)."""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from .. import _lib
from .._lib import RoiLevels, Rows2, YoloLevels, YoloLossCfg, check
from ._core import *  # noqa: F401,F403
from ._core import _F32, _I32, _TIMED, _ktimed, _need, _note_bytes, _ptr, _stream, _timed, _ws  # noqa: F401

# ------------------------------------------------------------------------------- SPPF max-pool
class _OutSlot:
    """Non-tensor wrapper (so autograd ignores it) around the tensor a custom Function writes its result into."""
    __slots__ = ("t",)

    def __init__(self, t: torch.Tensor):
        self.t = t


class CatBuffer:
    """Destination of a channel concatenation whose inputs are written IN PLACE by the kernels that produce them
    (fused BN+act, SPPF max-pool) instead of being copied there by torch.cat.

        buf = ops.CatBuffer(like, [c0, c1, ...])      # one channels-last allocation [B, sum(c), H, W]
        y0 = bn_act(..., out=buf.slot(0)); y1 = ...   # producers take the slot as `out`
        z = buf.join([y0, y1, ...])                   # parts already in their slot cost nothing; others are copied in

    slot(i) and join() hand out FRESH tensor objects over the buffer's storage (Tensor.set_ on the storage — not
    views), so autograd sees ordinary independent tensors and no view / version-counter bookkeeping is involved; the
    gradient of join() is the channel slices of the incoming gradient (views; the BN backward kernels read a strided
    dy in place)."""

    def __init__(self, like: torch.Tensor, channels: Sequence[int]):
        B, _, H, W = like.shape
        self.channels = [int(c) for c in channels]
        self.offsets = [0]
        for c in self.channels:
            self.offsets.append(self.offsets[-1] + c)
        self.buf = torch.empty((B, self.offsets[-1], H, W), dtype=like.dtype, device=like.device,
                               memory_format=torch.channels_last)

    def _alias(self, off: int, c: int) -> torch.Tensor:
        b = self.buf
        B, _, H, W = b.shape
        t = torch.empty(0, dtype=b.dtype, device=b.device)
        t.set_(b.untyped_storage(), b.storage_offset() + off, (B, c, H, W), b.stride())
        return t

    def slot(self, i: int) -> torch.Tensor:
        return self._alias(self.offsets[i], self.channels[i])

    def in_place(self, i: int, t: torch.Tensor) -> bool:
        return (t.data_ptr() == self.buf.data_ptr() + self.offsets[i] * self.buf.element_size()
                and t.shape[1] == self.channels[i] and t.stride() == self.buf.stride() and t.dtype == self.buf.dtype)

    def join(self, parts: Sequence[torch.Tensor]) -> torch.Tensor:
        if len(parts) != len(self.channels):
            raise _lib.HdError("CatBuffer.join: one tensor per slot")
        return _CatJoinFn.apply(self, *parts)


class _CatJoinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cb: "CatBuffer", *parts):
        for i, p in enumerate(parts):
            if not cb.in_place(i, p):
                cb.slot(i).copy_(p)  # a producer that could not write in place (or a foreign tensor): torch.cat's copy
        ctx.offsets = cb.offsets
        return cb._alias(0, cb.offsets[-1])

    @staticmethod
    def backward(ctx, g):
        o = ctx.offsets
        return (None,) + tuple(g[:, o[i]:o[i + 1]] for i in range(len(o) - 1))


class _MaxPool5Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, out_slot):
        lib = _lib.load()
        B, Cc, H, W = x.shape
        dt = {torch.float32: 0, torch.bfloat16: 1}[x.dtype]
        vec = 4 if dt == 0 else 8
        ld_in = _cl_row_stride(x, vec)
        out = out_slot.t if out_slot is not None else torch.empty((B, Cc, H, W), dtype=x.dtype, device=x.device,
                                                                  memory_format=torch.channels_last)
        ld_out = _cl_row_stride(out, vec)
        if ld_in is None or ld_out is None:
            raise _lib.HdError("maxpool5: input / output must be (channel slices of) channels-last tensors")
        idx = torch.empty((B, H, W, Cc), dtype=torch.uint8, device=x.device)
        check(lib.hd_maxpool5_fwd(_ptr(x), int(ld_in), dt, B, H, W, Cc, _ptr(out), int(ld_out), _ptr(idx), _stream(x)),
              "hd_maxpool5_fwd")
        ctx.save_for_backward(idx)
        ctx.meta = (dt, B, H, W, Cc)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        dt, B, H, W, Cc = ctx.meta
        ld = _cl_row_stride(g, 4 if dt == 0 else 8)
        if ld is None or g.dtype != (torch.float32 if dt == 0 else torch.bfloat16):
            g = g.to(torch.float32 if dt == 0 else torch.bfloat16).contiguous(memory_format=torch.channels_last)
            ld = Cc
        gin = torch.empty((B, Cc, H, W), dtype=g.dtype, device=g.device, memory_format=torch.channels_last)
        check(lib.hd_maxpool5_bwd(_ptr(g), int(ld), _ptr(idx), dt, B, H, W, Cc, _ptr(gin), _stream(g)), "hd_maxpool5_bwd")
        return gin, None


def maxpool5_supported(x: torch.Tensor) -> bool:
    """HIP path: HIP tensor, channels-last (dense, or a channel slice of a wider channels-last tensor), float32
    (C % 4 == 0) or bfloat16 (C % 8 == 0)."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    vec = 4 if x.dtype == torch.float32 else 8
    return x.shape[1] % vec == 0 and _cl_row_stride(x, vec) is not None


def maxpool5(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """5x5 / stride 1 / pad 2 max-pool on a channels-last HIP tensor (csrc/pool.hip): forward records the winning
    tap, backward is a gather (no atomics).  out: a CatBuffer slot to write the result into."""
    if not maxpool5_supported(x):
        raise _lib.HdError("maxpool5: needs a channels-last float32/bfloat16 HIP tensor")
    return _MaxPool5Fn.apply(x, _OutSlot(out) if out is not None else None)


class _MaxPool3s2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        B, Cc, H, W = x.shape
        dt = {torch.float32: 0, torch.bfloat16: 1}[x.dtype]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        out = torch.empty((B, Cc, Ho, Wo), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        idx = torch.empty((B, Ho, Wo, Cc), dtype=torch.uint8, device=x.device)
        check(lib.hd_maxpool3s2_fwd(_ptr(x), dt, B, H, W, Cc, _ptr(out), _ptr(idx), _stream(x)), "hd_maxpool3s2_fwd")
        ctx.save_for_backward(idx)
        ctx.meta = (dt, B, H, W, Cc)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        dt, B, H, W, Cc = ctx.meta
        want = torch.float32 if dt == 0 else torch.bfloat16
        if g.dtype != want or not g.is_contiguous(memory_format=torch.channels_last):
            g = g.to(want).contiguous(memory_format=torch.channels_last)
        gin = torch.empty((B, Cc, H, W), dtype=g.dtype, device=g.device, memory_format=torch.channels_last)
        check(lib.hd_maxpool3s2_bwd(_ptr(g), _ptr(idx), dt, B, H, W, Cc, _ptr(gin), _stream(g)), "hd_maxpool3s2_bwd")
        return gin


def maxpool3s2_supported(x: torch.Tensor) -> bool:
    """HIP path of the 3x3 / stride 2 / pad 1 max-pool: a dense channels-last float32 (C % 4 == 0) / bfloat16 (C % 8 == 0) tensor."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    vec = 4 if x.dtype == torch.float32 else 8
    return x.shape[1] % vec == 0 and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0


def maxpool3s2(x: torch.Tensor) -> torch.Tensor:
    """3x3 / stride 2 / pad 1 max-pool (the ResNet stem) on a dense channels-last HIP tensor (csrc/pool.hip): one byte of tap
    index per output element, gather backward without atomics; values and gradients equal F.max_pool2d(x, 3, 2, 1)."""
    if not maxpool3s2_supported(x):
        raise _lib.HdError("maxpool3s2: needs a dense channels-last float32/bfloat16 HIP tensor")
    return _MaxPool3s2Fn.apply(x)


# ------------------------------------------------------------------------------- fused BN + act
_ACT_CODE = {None: 0, "none": 0, "silu": 1, "relu": 2}


class _StatArena:
    """One zero-initialised fp32 arena per device for the per-channel sums the BN kernels accumulate
    with atomics.  ``reset()`` clears every slot with ONE kernel (the Trainer calls it once per
    step); a layer that runs twice without a reset in between clears its own slot."""

    def __init__(self, device):
        self.device = device
        self.chunks: List[torch.Tensor] = []
        self.used = 0
        self.gen = 0
        self.chunk_elems = 1 << 20

    def alloc(self, n: int) -> "_StatSlot":
        n = (n + 63) // 64 * 64
        if not self.chunks or self.used + n > self.chunk_elems:
            self.chunks.append(torch.zeros(max(self.chunk_elems, n), dtype=_F32, device=self.device))
            self.used = 0
        buf = self.chunks[-1][self.used:self.used + n]
        self.used += n
        return _StatSlot(self, buf)

    def reset(self):
        self.gen += 1
        for c in self.chunks:
            c.zero_()


class _StatSlot:
    __slots__ = ("arena", "buf", "fwd_gen", "bwd_gen")

    def __init__(self, arena, buf):
        self.arena, self.buf, self.fwd_gen, self.bwd_gen = arena, buf, -1, -1


_ARENAS: dict = {}


def _arena(device) -> _StatArena:
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _ARENAS:
        _ARENAS[key] = _StatArena(device)
    return _ARENAS[key]


def bn_arena_reset(device) -> None:
    """Zero every BN statistics slot on `device` (one kernel).  Call once per training step."""
    _arena(torch.device(device)).reset()


def bn_act_supported(x: torch.Tensor) -> bool:
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    v = 4 if x.dtype == torch.float32 else 8
    C = x.shape[1]
    return C % v == 0 and C // v <= 256 and x.is_contiguous(memory_format=torch.channels_last)


def _cl_row_stride(t: torch.Tensor, vec: int):
    """Row stride (elements) of a [B,C,H,W] tensor whose memory is channels-last rows of >= C channels —
    dense channels-last, or a channel slice of a wider channels-last tensor (a torch.cat input's gradient).
    None when the layout is anything else or the 16-byte vector accesses would be misaligned."""
    B, Cc, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if Cc > 1 and sc != 1:
        return None
    ld = sw if W > 1 else (sh if H > 1 else (sb if B > 1 else Cc))
    if ld < Cc or ld % vec != 0:
        return None
    if (W > 1 and sw != ld) or (H > 1 and sh != W * ld) or (B > 1 and sb != H * W * ld):
        return None
    if (t.data_ptr() % 16) != 0:
        return None
    return ld


# ------------------------------------------------------------------------------- upsample 2x + concat
class _UpCatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, small, big):
        lib = _lib.load()
        B, Cs, Hs, Ws = small.shape
        Cb, H, W = big.shape[1], big.shape[2], big.shape[3]
        dt = 0 if small.dtype == torch.float32 else 1
        vec = 4 if dt == 0 else 8
        out = torch.empty((B, Cs + Cb, H, W), dtype=small.dtype, device=small.device, memory_format=torch.channels_last)
        check(lib.hd_upsample2x_cat_fwd(_ptr(small), int(_cl_row_stride(small, vec)), _ptr(big), int(_cl_row_stride(big, vec)), dt,
                                        B, H, W, Cs, Cb, _ptr(out), Cs + Cb, _stream(small)), "hd_upsample2x_cat_fwd")
        ctx.meta = (dt, B, H, W, Cs)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        dt, B, H, W, Cs = ctx.meta
        vec = 4 if dt == 0 else 8
        want = torch.float32 if dt == 0 else torch.bfloat16
        if g.dtype != want or _cl_row_stride(g, vec) is None:
            g = g.to(want).contiguous(memory_format=torch.channels_last)
        d_small = None
        if ctx.needs_input_grad[0]:
            d_small = torch.empty((B, Cs, H // 2, W // 2), dtype=want, device=g.device, memory_format=torch.channels_last)
            check(lib.hd_upsample2x_cat_bwd(_ptr(g), int(_cl_row_stride(g, vec)), dt, B, H, W, Cs, _ptr(d_small), Cs, _stream(g)),
                  "hd_upsample2x_cat_bwd")
        return d_small, (g[:, Cs:] if ctx.needs_input_grad[1] else None)


def upsample2x_cat_supported(small: torch.Tensor, big: torch.Tensor) -> bool:
    if not (small.is_cuda and big.is_cuda and small.dim() == 4 and big.dim() == 4 and small.dtype == big.dtype
            and small.dtype in (torch.float32, torch.bfloat16)):
        return False
    vec = 4 if small.dtype == torch.float32 else 8
    return (big.shape[0] == small.shape[0] and big.shape[2] == 2 * small.shape[2] and big.shape[3] == 2 * small.shape[3]
            and small.shape[1] % vec == 0 and big.shape[1] % vec == 0
            and _cl_row_stride(small, vec) is not None and _cl_row_stride(big, vec) is not None)


def upsample2x_cat(small: torch.Tensor, big: torch.Tensor) -> torch.Tensor:
    """cat(nearest-upsample-2x(small), big) along channels, one kernel, channels-last (csrc/upcat.hip)."""
    if not upsample2x_cat_supported(small, big):
        raise _lib.HdError("upsample2x_cat: needs same-dtype channels-last HIP tensors with big = 2x small spatially")
    return _UpCatFn.apply(small, big)


# ------------------------------------------------------------------------------- 1x1 convolution (MFMA GEMM)
def pointwise_conv_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """x: [B, Cin, H, W] bf16 channels-last rows (dense or a channel slice); weight: [Cout, Cin, 1, 1] bf16 dense."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
            and weight.dim() == 4 and weight.shape[2:] == (1, 1) and weight.shape[1] == x.shape[1]):
        return False
    B, Cin, H, W = x.shape
    if not _lib.load().hd_pointwise_conv_supported(B * H * W, Cin, weight.shape[0]):
        return False
    return _cl_row_stride(x, 8) is not None and weight.is_contiguous() and weight.data_ptr() % 16 == 0


def pointwise_conv_rows(x: torch.Tensor, w2: torch.Tensor, stats: Optional[torch.Tensor] = None,
                        out: Optional[torch.Tensor] = None, name: str = "pointwise_conv") -> torch.Tensor:
    """y[B, N, H, W] = 1x1 convolution of x[B, K, H, W] (channels-last rows) with w2[N, K] (bf16, dense), no autograd.
    stats: float32 [replicas * 2 * N], zeroed by the caller — the fused BatchNorm's (sum, sum of squares) are added."""
    lib = _lib.load()
    B, K, H, W = x.shape
    N = w2.shape[0]
    ld_x = _cl_row_stride(x, 8)
    if ld_x is None or x.dtype != torch.bfloat16 or w2.dtype != torch.bfloat16 or w2.shape[1] != K or not w2.is_contiguous():
        raise _lib.HdError("pointwise_conv: needs bf16 channels-last rows and a dense bf16 [Cout, Cin] weight")
    if out is None:
        out = torch.empty((B, N, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    ld_y = _cl_row_stride(out, 8)
    if ld_y is None or out.shape != (B, N, H, W) or out.dtype != torch.bfloat16:
        raise _lib.HdError("pointwise_conv: `out` must be a bf16 channels-last tensor (or channel slice) [B, Cout, H, W]")
    _note_bytes(name, 2 * (B * H * W * (K + N) + N * K), {"flops": 2 * B * H * W * K * N, "shape": (B * H * W, K, N)})
    with _ktimed(name):
        check(lib.hd_pointwise_conv(_ptr(x), int(ld_x), _ptr(w2), B * H * W, K, N, _ptr(out), int(ld_y), _ptr(stats), _stream(x)),
              "hd_pointwise_conv")
    return out


def _fwd_sums(slot: "_StatSlot", channels: int) -> torch.Tensor:
    """The slot's forward accumulators [replicas, 2, C], zero: the arena reset zeroed them once per step; a second
    forward of the same layer since then clears its own."""
    sums = slot.buf[: _lib.load().hd_bn_act_replicas() * 2 * channels]
    if slot.fwd_gen == slot.arena.gen:
        sums.zero_()
    slot.fwd_gen = slot.arena.gen
    return sums


class _WgradChain:
    """Weight gradients are computed as row-sliced float32 partial products (a slab) plus their sum over the slices.  Inside
    a training step (`enabled`, set by train.Trainer around backward) the sum of layer L does not get a launch of its own:
    it rides in extra workgroups of the NEXT weight-gradient launch (hd_wgrad_job `piggyback`), and only the last one of a
    backward pass is launched by itself (`flush`, called by dist.GradReducer before it reads any gradient and by the Trainer
    after backward).  Outside a step every weight gradient is summed at once, as before.

    Slabs: two per device, used alternately — the pending job's slab must survive until the next launch has consumed it, and
    that launch writes the other one.  A slab that has to grow is replaced, never released (a captured hipGraph may still
    point at the old one: ADVICE r02), so the superseded buffers stay referenced for the life of the process."""

    enabled = False
    _slabs: dict = {}      # device key -> [buf0, buf1]
    _retired: list = []    # superseded slabs, kept alive
    _pending: dict = {}    # device key -> (WgradJob, keep-alive tensors)
    _handed: list = []     # (weakref to the weight leaf, dw.data_ptr()) of the chained gradients since the last flush

    @staticmethod
    def _key(device):
        return (device.type, device.index if device.index is not None else torch.cuda.current_device())

    @classmethod
    def slab(cls, device, nbytes: int) -> torch.Tensor:
        """A slab of >= nbytes that is NOT the one the pending job's partial products live in."""
        key = cls._key(device)
        pair = cls._slabs.setdefault(key, [None, None])
        pend = cls._pending.get(key)
        busy = pend[1][0] if pend is not None else None
        t = 1 if (busy is not None and pair[0] is busy) else 0
        buf = pair[t]
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                cls._retired.append(buf)
            buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
            pair[t] = buf
        return buf

    @staticmethod
    def _reduce_now(job, device) -> None:
        """The slice sum as a launch of its own (noted for bench.py's per-launch rooflines: slab in, bf16 gradient out)."""
        _note_bytes("wgrad_reduce", int(job.slices) * int(job.count) * 4 + int(job.count) * 2, {"flops": int(job.slices) * int(job.count)})
        with _ktimed("wgrad_reduce"):
            check(_lib.load().hd_wgrad_reduce(C.byref(job), C.c_void_p(torch.cuda.current_stream(device).cuda_stream)),
                  "hd_wgrad_reduce")

    @classmethod
    def piggyback(cls, device):
        """The pending job of `device` (to ride in the launch being prepared), or None."""
        pend = cls._pending.pop(cls._key(device), None)
        return pend

    @classmethod
    def submit(cls, device, job, keep, leaf=None) -> None:
        """`job` (a filled WgradJob) has been launched: chain it, or sum it right away.  Chained only inside a training step
        AND when the gradient goes to an autograd LEAF (the bf16 weight shadow): AccumulateGrad takes a leaf's first gradient
        as it is and nothing reads it before the bucket is packed; a non-leaf weight (autocast's cast of a float32 master)
        has its gradient cast back by the very next autograd node, which must see the finished sum."""
        if cls.enabled and leaf is not None and leaf.is_leaf:
            cls._pending[cls._key(device)] = (job, keep)
            if leaf is not None:
                import weakref
                cls._handed.append((weakref.ref(leaf), int(job.dw)))
        else:
            cls._reduce_now(job, device)

    @classmethod
    def flush_device(cls, device) -> None:
        """Sum the pending job of one device now (a launch that cannot carry it is about to use the stream)."""
        pend = cls._pending.pop(cls._key(device), None)
        if pend is not None:
            cls._reduce_now(pend[0], device)

    @classmethod
    def reset(cls) -> None:
        """Forget the pending slice sum and the hand-over records WITHOUT launching anything: for a step that ended in an
        exception (a failed hipGraph capture above all — the pending job's gradient pointer then lies in the destroyed capture's
        private pool, and a later launch that carried it would write through freed memory).  The gradients of the
        aborted step are lost, which is what an aborted step means."""
        cls._pending.clear()
        cls._handed.clear()

    @classmethod
    def flush(cls) -> None:
        for key, (job, keep) in list(cls._pending.items()):
            cls._reduce_now(job, torch.device(key[0], key[1]))
        cls._pending.clear()
        # a chained gradient is complete only after the launch that carried its sum: autograd must have taken the returned
        # tensor as it is (AccumulateGrad steals a fresh dense gradient); a copy taken earlier would hold unsummed bytes
        for ref, ptr in cls._handed:
            leaf = ref()
            if leaf is not None and leaf.grad is not None and leaf.grad.data_ptr() != ptr:
                cls._handed.clear()
                raise _lib.HdError("a chained weight gradient was copied before its slices were summed (autograd did not take "
                                   "the tensor as returned); run with ops._WgradChain.enabled = False")
        cls._handed.clear()


def flush_wgrad() -> None:
    """Sum the partial products of the last weight-gradient launch (no-op when nothing is pending): after this call every
    weight gradient handed out so far is complete for the current stream."""
    _WgradChain.flush()


def _wgrad_workspace(device, nbytes: int) -> torch.Tensor:
    """One of the device's two float32 slabs (see _WgradChain), at least `nbytes` long."""
    return _WgradChain.slab(device, nbytes)


def pointwise_dgrad_rows(dy: torch.Tensor, w2: torch.Tensor, out: Optional[torch.Tensor] = None,
                         add: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx[B, Cin, H, W] = dy[B, Cout, H, W] . w2[Cout, Cin] on channels-last rows, no autograd; Cout in {32, 64, 128, 256}.
    add: bf16 channels-last rows [B, Cin, H, W] summed into the result in the kernel's epilogue (the gradient that reaches
    the same input through a shortcut) — float32 add of the two bf16 values, one rounding, like a separate bf16 add."""
    lib = _lib.load()
    B, _, H, W = dy.shape
    N = w2.shape[0]
    K = w2.shape[1]
    ld_dy = _cl_row_stride(dy, 8)
    if ld_dy is None or dy.dtype != torch.bfloat16 or w2.dtype != torch.bfloat16 or not w2.is_contiguous() \
            or dy.shape[1] != N:
        raise _lib.HdError("pointwise_dgrad: needs bf16 channels-last rows and the dense bf16 [Cout, Cin] forward weight")
    if out is None:
        out = torch.empty((B, K, H, W), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    ld_dx = _cl_row_stride(out, 8)
    if ld_dx is None or out.shape != (B, K, H, W) or out.dtype != torch.bfloat16:
        raise _lib.HdError("pointwise_dgrad: `out` must be a bf16 channels-last tensor (or channel slice) [B, Cin, H, W]")
    _note_bytes("pointwise_dgrad", 2 * (B * H * W * (K + N) + N * K),
                {"flops": 2 * B * H * W * K * N, "shape": (B * H * W, N, K)})
    if add is not None:
        ld_add = _cl_row_stride(add, 8)
        if ld_add is None or add.dtype != torch.bfloat16 or tuple(add.shape) != (B, K, H, W):
            raise _lib.HdError("pointwise_dgrad: `add` must be bf16 channels-last rows of the result's shape")
        with _ktimed("pointwise_dgrad"):
            check(lib.hd_pointwise_conv_dgrad_add(_ptr(dy), int(ld_dy), _ptr(w2), B * H * W, K, N, _ptr(add), int(ld_add), _ptr(out),
                                                  int(ld_dx), _stream(dy)), "hd_pointwise_conv_dgrad_add")
        return out
    with _ktimed("pointwise_dgrad"):
        check(lib.hd_pointwise_conv_dgrad(_ptr(dy), int(ld_dy), _ptr(w2), B * H * W, K, N, _ptr(out), int(ld_dx), _stream(dy)),
              "hd_pointwise_conv_dgrad")
    return out


def pointwise_wgrad_supported(R: int, cin: int, cout: int) -> bool:
    return bool(_lib.load().hd_pointwise_wgrad_supported(int(R), int(cin), int(cout)))


def pointwise_wgrad_rows(dy: torch.Tensor, x: torch.Tensor, leaf: Optional[torch.Tensor] = None,
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dw[Cout, Cin] (bf16) = sum over rows of dy[B, Cout, H, W]^T x[B, Cin, H, W], channels-last rows, no autograd.
    leaf: the tensor autograd will hand the result to (checked by the chain, see _WgradChain).  Inside a training step the
    returned tensor is complete only after the next weight-gradient launch or flush_wgrad()."""
    lib = _lib.load()
    B, K, H, W = x.shape
    N = int(dy.shape[1])
    ld_dy, ld_x = _cl_row_stride(dy, 8), _cl_row_stride(x, 8)
    if (ld_dy is None or ld_x is None or dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16
            or x.shape[0] != dy.shape[0] or x.shape[2:] != dy.shape[2:]):
        raise _lib.HdError("pointwise_wgrad: needs bf16 channels-last rows of equal batch and spatial size")
    R = B * H * W
    nbytes = lib.hd_pointwise_wgrad_workspace_bytes(R, K, N)
    if nbytes == 0:
        raise _lib.HdError("pointwise_wgrad: Cin and Cout must be multiples of 64, or multiples of 32 up to 256")
    ws = _wgrad_workspace(dy.device, nbytes)
    dw = out if out is not None else torch.empty((N, K), dtype=torch.bfloat16, device=dy.device)
    if dw.dtype != torch.bfloat16 or dw.numel() != N * K or not dw.is_contiguous():
        raise _lib.HdError("pointwise_wgrad: `out` must be a dense bf16 tensor of Cout x Cin elements")
    pend = _WgradChain.piggyback(dy.device)
    job = _lib.WgradJob()
    _note_bytes("pointwise_wgrad", 2 * (R * (K + N) + N * K), {"flops": 2 * R * K * N, "shape": (R, K, N)})
    with _ktimed("pointwise_wgrad"):
        check(lib.hd_pointwise_wgrad_partial(_ptr(dy), int(ld_dy), _ptr(x), int(ld_x), R, K, N, _ptr(dw),
                                             _ptr(ws), ws.numel(), C.byref(pend[0]) if pend is not None else None, C.byref(job),
                                             _stream(dy)), "hd_pointwise_wgrad_partial")
    _WgradChain.submit(dy.device, job, (ws,), leaf)  # not dw: a second reference would make autograd copy it
    return dw


class PwBwdSwitch:
    """The one-launch backward of the 1x1 layers (csrc/pointwise_bwd.hip: data gradient + weight-gradient partials from one
    staging of dy).  HD_FUSED_PW_BWD=0 goes back to the pair hd_pointwise_conv_dgrad + hd_pointwise_wgrad_partial."""
    enabled = os.environ.get("HD_FUSED_PW_BWD", "1") != "0"


def pointwise_bwd_supported(R: int, cin: int, cout: int) -> bool:
    return PwBwdSwitch.enabled and bool(_lib.load().hd_pointwise_bwd_supported(int(R), int(cin), int(cout)))


def pointwise_bwd_rows(dy: torch.Tensor, w2: torch.Tensor, x: torch.Tensor, add: Optional[torch.Tensor] = None,
                       leaf: Optional[torch.Tensor] = None, out_dx: Optional[torch.Tensor] = None,
                       out_dw: Optional[torch.Tensor] = None):
    """(dx, dw) of a 1x1 convolution y = x . w2^T on channels-last rows, ONE launch, no autograd: dx[B, Cin, H, W] = dy . w2
    (+ add), dw[Cout, Cin] = dy^T . x (bf16; complete after the next weight-gradient launch or flush_wgrad(), see
    _WgradChain).  dy [B, Cout, H, W], x [B, Cin, H, W] bf16 channels-last rows; w2 the dense bf16 [Cout, Cin] forward weight."""
    lib = _lib.load()
    B, N, H, W = dy.shape
    K = w2.shape[1]
    R = B * H * W
    ld_dy, ld_x = _cl_row_stride(dy, 8), _cl_row_stride(x, 8)
    if (ld_dy is None or ld_x is None or dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16 or w2.dtype != torch.bfloat16
            or not w2.is_contiguous() or tuple(w2.shape) != (N, K) or tuple(x.shape) != (B, K, H, W)):
        raise _lib.HdError("pointwise_bwd: needs bf16 channels-last rows dy [B, Cout, H, W], x [B, Cin, H, W] and the dense bf16 "
                           "[Cout, Cin] forward weight")
    nbytes = lib.hd_pointwise_bwd_workspace_bytes(R, K, N)
    if nbytes == 0:
        raise _lib.HdError("pointwise_bwd: Cout must be 32 / 64 / 128 / 256 and Cin a multiple of 32")
    dx = out_dx if out_dx is not None else torch.empty((B, K, H, W), dtype=torch.bfloat16, device=dy.device,
                                                       memory_format=torch.channels_last)
    ld_dx = _cl_row_stride(dx, 8)
    if ld_dx is None or tuple(dx.shape) != (B, K, H, W) or dx.dtype != torch.bfloat16:
        raise _lib.HdError("pointwise_bwd: `out_dx` must be a bf16 channels-last tensor (or channel slice) [B, Cin, H, W]")
    ld_add = 0
    if add is not None:
        ld_add = _cl_row_stride(add, 8)
        if ld_add is None or add.dtype != torch.bfloat16 or tuple(add.shape) != (B, K, H, W):
            raise _lib.HdError("pointwise_bwd: `add` must be bf16 channels-last rows of dx's shape")
    dw = out_dw if out_dw is not None else torch.empty((N, K), dtype=torch.bfloat16, device=dy.device)
    if dw.dtype != torch.bfloat16 or dw.numel() != N * K or not dw.is_contiguous():
        raise _lib.HdError("pointwise_bwd: `out_dw` must be a dense bf16 tensor of Cout x Cin elements")
    ws = _wgrad_workspace(dy.device, nbytes)
    pend = _WgradChain.piggyback(dy.device)
    job = _lib.WgradJob()
    # each operand once: dy, x in; dx out (+ the addend in); w and the bf16 dw
    _note_bytes("pointwise_bwd", 2 * (R * (N + 2 * K) + 2 * N * K) + (2 * R * K if add is not None else 0),
                {"flops": 4 * R * K * N, "shape": (R, K, N)})
    with _ktimed("pointwise_bwd"):
        check(lib.hd_pointwise_bwd(_ptr(dy), int(ld_dy), _ptr(w2), _ptr(x), int(ld_x), R, K, N, _ptr(add), int(ld_add), _ptr(dx),
                                   int(ld_dx), _ptr(dw), _ptr(ws), ws.numel(), C.byref(pend[0]) if pend is not None else None,
                                   C.byref(job), _stream(dy)), "hd_pointwise_bwd")
    _WgradChain.submit(dy.device, job, (ws,), leaf)
    return dx, dw


# ------------------------------------------------------------------------------- Detect convolution (YOLOv5 head)
class HeadConvSwitch:
    """The Detect convolutions on the HIP kernels (hd_head_conv_fwd / hd_head_conv_bwd, padded float32 rows).  HD_HEAD_CONV=0:
    the library's biased convolution + cast, as in rounds 1-3."""
    enabled = os.environ.get("HD_HEAD_CONV", "1") != "0"
    # the loss writes its logit gradient as bf16 padded rows and runs the Detect convolutions' backward itself (yolo_loss(heads=))
    bf16_grad = os.environ.get("HD_HEAD_GRAD_BF16", "1") != "0"


def head_conv_pitch(cout: int) -> int:
    """Columns per row of the padded head output (the next of 32 / 64 / 128 / 256 >= cout; 0 = unsupported)."""
    return int(_lib.load().hd_head_conv_pitch(int(cout)))


def head_conv_supported(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> bool:
    """bf16 channels-last rows, a dense bf16 [Cout, Cin, 1, 1] weight, Cout <= 256 and a backward the fused kernel takes;
    `bias` (when given): float32, dense, on x's device, one entry per output channel — what hd_head_conv_fwd reads.
    (A forward the streaming kernel does not take — Cin not 32/64/128/256k — runs on the library with padded weights.)"""
    if bias is not None and not (bias.dtype == _F32 and bias.device == x.device and bias.is_contiguous()
                                 and bias.numel() == weight.shape[0]):
        return False
    if not (HeadConvSwitch.enabled and x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16
            and weight.dtype == torch.bfloat16 and weight.dim() == 4 and tuple(weight.shape[2:]) == (1, 1)
            and weight.shape[1] == x.shape[1] and weight.is_contiguous() and _cl_row_stride(x, 8) is not None):
        return False
    B, K, H, W = x.shape
    return bool(_lib.load().hd_head_conv_bwd_supported(B * H * W, K, int(weight.shape[0])))


def head_conv_rows(x: torch.Tensor, w2: torch.Tensor, bias: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y [B, pitch, H, W] float32 channels-last = x . w2^T + bias in the first Cout channels, zeros in the rest; no autograd."""
    lib = _lib.load()
    B, K, H, W = x.shape
    N = w2.shape[0]
    P = head_conv_pitch(N)
    ld_x = _cl_row_stride(x, 8)
    if (ld_x is None or x.dtype != torch.bfloat16 or w2.dtype != torch.bfloat16 or tuple(w2.shape) != (N, K) or not w2.is_contiguous()
            or bias.dtype != _F32 or bias.numel() != N or not bias.is_contiguous() or P == 0):
        raise _lib.HdError("head_conv: needs bf16 channels-last rows, a dense bf16 [Cout <= 256, Cin] weight and a float32 bias")
    if out is None:
        out = torch.empty((B, P, H, W), dtype=_F32, device=x.device, memory_format=torch.channels_last)
    ld_y = _cl_row_stride(out, 4)
    if ld_y is None or tuple(out.shape) != (B, P, H, W) or out.dtype != _F32:
        raise _lib.HdError("head_conv: `out` must be a float32 channels-last tensor [B, pitch, H, W]")
    R = B * H * W
    if K <= 256:   # one launch: timed and priced per launch (bench.py); longer reductions run as K / 256 launches, left unpriced
        _note_bytes("head_conv", 2 * R * K + 4 * R * P + 2 * N * K, {"flops": 2 * R * K * P, "shape": (R, K, N)})
    with (_ktimed("head_conv") if K <= 256 else contextlib.nullcontext()):
        check(lib.hd_head_conv_fwd(_ptr(x), int(ld_x), _ptr(w2), _ptr(bias), R, K, N, _ptr(out), int(ld_y), _stream(x)),
              "hd_head_conv_fwd")
    return out


class _HeadConvFn(torch.autograd.Function):
    """Detect convolution: x [B, Cin, H, W] bf16 rows -> y [B, pitch, H, W] FLOAT32 rows (accumulator + bias, unrounded; padding
    channels zero).  backward takes the float32 gradient of y whose padding channels are zero (the loss kernels write them),
    rounds it to bf16 once — what `.float()`'s backward did — and runs hd_head_conv_bwd: dx and the weight-gradient partials
    from one staging.  The bias takes no gradient here: ops.yolo_loss(biases=...) produces it."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        N, K = weight.shape[0], weight.shape[1]
        B, _, H, W = x.shape
        if bool(_lib.load().hd_head_conv_fwd_supported(B * H * W, K, N)):
            y = head_conv_rows(x, weight.view(N, K), bias.detach().contiguous())
        else:  # the library's convolution on the padded weights: the same rows, rounded to bf16 before the bias
            P = head_conv_pitch(N)
            wp = torch.cat([weight, weight.new_zeros((P - N, K, 1, 1))], 0) if P > N else weight
            bp = torch.cat([bias.detach(), bias.new_zeros(P - N)], 0) if P > N else bias.detach()
            xd = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            y = torch.nn.functional.conv2d(xd, wp).float().add_(bp.view(1, P, 1, 1))
            if not y.is_contiguous(memory_format=torch.channels_last):
                y = y.contiguous(memory_format=torch.channels_last)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        dy = gy.to(torch.bfloat16)
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        dx, dw = head_conv_bwd_rows(dy, x, weight)
        return dx, dw, None


def head_conv_bwd_rows(dy: torch.Tensor, x: torch.Tensor, weight: torch.Tensor):
    """(dx [B, Cin, H, W] bf16, dw like `weight` bf16) of the Detect convolution from dy [B, pitch, H, W] bf16 channels-last
    rows whose padding channels are zero; one launch, the weight gradient's slice sum chained as for every other layer
    (_WgradChain; complete after the next weight-gradient launch or flush_wgrad())."""
    lib = _lib.load()
    N, K = weight.shape[0], weight.shape[1]
    B, _, H, W = x.shape
    R = B * H * W
    ld_dy, ld_x = _cl_row_stride(dy, 8), _cl_row_stride(x, 8)
    if (ld_dy is None or ld_x is None or dy.dtype != torch.bfloat16 or dy.shape[1] != head_conv_pitch(N)
            or tuple(dy.shape[2:]) != (H, W) or dy.shape[0] != B):
        raise _lib.HdError("head_conv backward: the gradient must have the padded bf16 channels-last rows of the output")
    dx = torch.empty((B, K, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    dw = torch.empty((N, K), dtype=torch.bfloat16, device=x.device)
    nbytes = lib.hd_head_conv_bwd_workspace_bytes(R, K, N)
    ws = _wgrad_workspace(x.device, nbytes)
    pend = _WgradChain.piggyback(x.device)
    job = _lib.WgradJob()
    P = dy.shape[1]
    _note_bytes("head_conv_bwd", 2 * (R * (P + 2 * K) + 2 * N * K), {"flops": 4 * R * K * P, "shape": (R, K, N)})
    with _ktimed("head_conv_bwd"):
        check(lib.hd_head_conv_bwd(_ptr(dy), int(ld_dy), _ptr(weight), _ptr(x), int(ld_x), R, K, N, _ptr(dx), K, _ptr(dw),
                                   _ptr(ws), ws.numel(), C.byref(pend[0]) if pend is not None else None, C.byref(job),
                                   _stream(x)), "hd_head_conv_bwd")
    _WgradChain.submit(x.device, job, (ws,), weight)
    return dx, dw.view(weight.shape)


def head_conv(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """Differentiable (x, weight) Detect convolution -> padded float32 rows [B, head_conv_pitch(Cout), H, W]; see _HeadConvFn."""
    return _HeadConvFn.apply(x, weight, bias)


def conv3x3_wgrad_supported(x: torch.Tensor, cout: int, stride: int = 1) -> bool:
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and _cl_row_stride(x, 8) is not None):
        return False
    B, K, H, W = x.shape
    if stride not in (1, 2) or H % stride or W % stride:
        return False
    return bool(_lib.load().hd_conv3x3_wgrad_supported(B, H // stride, W // stride, K, int(cout), int(stride)))


def conv3x3_wgrad_rows(dy: torch.Tensor, x: torch.Tensor, stride: int = 1, out: Optional[torch.Tensor] = None,
                       leaf: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Weight gradient of a 3x3 / pad 1 convolution of stride 1 or 2: dy [B, Cout, H, W], x [B, Cin, stride H, stride W]
    bf16 channels-last rows -> dw [Cout, Cin, 3, 3] bf16 in channels-last memory ([Cout][ky][kx][Cin]); no autograd."""
    lib = _lib.load()
    B, N, H, W = dy.shape
    K = x.shape[1]
    ld_dy, ld_x = _cl_row_stride(dy, 8), _cl_row_stride(x, 8)
    if (ld_dy is None or ld_x is None or dy.dtype != torch.bfloat16 or x.dtype != torch.bfloat16
            or x.shape[0] != B or x.shape[2] != stride * H or x.shape[3] != stride * W):
        raise _lib.HdError("conv3x3_wgrad: needs bf16 channels-last rows, x spatially `stride` times dy")
    nbytes = lib.hd_conv3x3_wgrad_workspace_bytes(B, H, W, K, N, int(stride))
    if nbytes == 0:
        raise _lib.HdError("conv3x3_wgrad: channel counts must be multiples of 32 (<= 1024), stride 1 or 2, and one row block must fit LDS")
    ws = _wgrad_workspace(dy.device, nbytes)   # the other slab of the pair: a pending chained job stays intact
    dw = out if out is not None else torch.empty((N, K, 3, 3), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    if dw.dtype != torch.bfloat16 or tuple(dw.shape) != (N, K, 3, 3) or not dw.is_contiguous(memory_format=torch.channels_last):
        raise _lib.HdError("conv3x3_wgrad: `out` must be a channels-last bf16 [Cout, Cin, 3, 3] tensor")
    pend = _WgradChain.piggyback(dy.device)
    job = _lib.WgradJob()
    _note_bytes("conv3x3_wgrad", 2 * (B * H * W * (stride * stride * K + N) + 9 * N * K),
                {"flops": 18 * B * H * W * K * N, "shape": (B, H, W, K, N, stride)})
    with _ktimed("conv3x3_wgrad"):
        check(lib.hd_conv3x3_wgrad_partial(_ptr(dy), int(ld_dy), _ptr(x), int(ld_x), B, H, W, K, N, int(stride), _ptr(dw), _ptr(ws),
                                           ws.numel(), C.byref(pend[0]) if pend is not None else None, C.byref(job), _stream(dy)),
              "hd_conv3x3_wgrad_partial")
    _WgradChain.submit(dy.device, job, (ws,), leaf)
    return dw


def conv3x3_fwd_supported(x: torch.Tensor, cout: int) -> bool:
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and _cl_row_stride(x, 8) is not None):
        return False
    B, K, H, W = x.shape
    return bool(_lib.load().hd_conv3x3_fwd_supported(B, H, W, K, int(cout)))


def conv3x3_fwd_rows(x: torch.Tensor, w: torch.Tensor, stats: Optional[torch.Tensor] = None, name: str = "conv3x3_fwd") -> torch.Tensor:
    """y = 3x3 / stride 1 / pad 1 convolution of x [B, Cin, H, W] (bf16 channels-last rows) with w [Cout, Cin, 3, 3] (bf16,
    channels-last memory [Cout][3][3][Cin]); stats as for pointwise_conv_rows; no autograd."""
    lib = _lib.load()
    B, K, H, W = x.shape
    N = w.shape[0]
    ld_x = _cl_row_stride(x, 8)
    if (ld_x is None or x.dtype != torch.bfloat16 or w.dtype != torch.bfloat16 or tuple(w.shape) != (N, K, 3, 3)
            or not w.is_contiguous(memory_format=torch.channels_last)):
        raise _lib.HdError("conv3x3_fwd: needs bf16 channels-last rows and a channels-last bf16 [Cout, Cin, 3, 3] weight")
    y = torch.empty((B, N, H, W), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    _note_bytes(name, 2 * (B * H * W * (K + N) + 9 * N * K), {"flops": 18 * B * H * W * K * N, "shape": (B, H, W, K, N, 1)})
    with _ktimed(name):
        check(lib.hd_conv3x3_fwd(_ptr(x), int(ld_x), _ptr(w), B, H, W, K, N, _ptr(y), N, _ptr(stats), _stream(x)), "hd_conv3x3_fwd")
    return y


def conv3x3s2_fwd_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """hd_conv3x3s2_fwd (csrc/conv3x3s2_fwd.hip): the first stride-2 layer of the YOLOv5 backbone (32 -> 64 channels, 320 pixels wide
    in, even height)."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and _cl_row_stride(x, 8) is not None
            and weight.dtype == torch.bfloat16 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)
            and weight.shape[1] == x.shape[1] and weight.is_contiguous(memory_format=torch.channels_last)):
        return False
    B, K, H, W = x.shape
    if H % 2 or W % 2:
        return False
    return bool(_lib.load().hd_conv3x3s2_fwd_supported(B, H // 2, W // 2, K, int(weight.shape[0])))


def conv3x3s2_fwd_rows(x: torch.Tensor, weight: torch.Tensor, stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y [B, Cout, H / 2, W / 2] (bf16 channels-last) = 3x3 / stride 2 / pad 1 convolution of x with the channels-last weight
    [Cout, Cin, 3, 3]; stats as for pointwise_conv_rows; no autograd."""
    lib = _lib.load()
    if not conv3x3s2_fwd_supported(x, weight):
        raise _lib.HdError("conv3x3s2_fwd: needs bf16 channels-last rows [B, 32, 2 H, 320] and a channels-last bf16 [64, 32, 3, 3] weight")
    B, K, H, W = x.shape
    N, Ho, Wo = weight.shape[0], H // 2, W // 2
    y = torch.empty((B, N, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    _note_bytes("conv3x3s2_fwd", 2 * (B * Ho * Wo * (4 * K + N) + 9 * N * K), {"flops": 18 * B * Ho * Wo * K * N, "shape": (B, Ho, Wo, K, N, 2)})
    with _ktimed("conv3x3s2_fwd"):
        check(lib.hd_conv3x3s2_fwd(_ptr(x), int(_cl_row_stride(x, 8)), _ptr(weight), B, Ho, Wo, K, N, _ptr(y), N, _ptr(stats), _stream(x)),
              "hd_conv3x3s2_fwd")
    return y


def conv3x3s2_wgrad_supported(dy: torch.Tensor, x: torch.Tensor) -> bool:
    """hd_conv3x3s2_wgrad_c32 (csrc/conv3x3s2_wgrad.hip): the first stride-2 layer of the YOLOv5 backbone."""
    if not (dy.is_cuda and dy.dim() == 4 and x.dim() == 4 and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16
            and _cl_row_stride(dy, 8) is not None and _cl_row_stride(x, 8) is not None):
        return False
    B, N, Ho, Wo = dy.shape
    if tuple(x.shape[::3]) != (B, 2 * Wo) or x.shape[2] != 2 * Ho:
        return False
    return bool(_lib.load().hd_conv3x3s2_wgrad_c32_supported(B, Ho, Wo, int(x.shape[1]), N))


def conv3x3s2_wgrad_rows(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """dw [Cout, Cin, 3, 3] (bf16, channels-last memory [Cout][ky][kx][Cin]) of a 3x3 / stride 2 / pad 1 convolution from dy
    [B, Cout, Ho, Wo] and x [B, Cin, 2 Ho, 2 Wo] (bf16 channels-last rows); no autograd."""
    lib = _lib.load()
    if not conv3x3s2_wgrad_supported(dy, x):
        raise _lib.HdError("conv3x3s2_wgrad: needs bf16 channels-last rows dy [B, 64, H, 160] and x [B, 32, 2 H, 320]")
    B, N, Ho, Wo = dy.shape
    K = x.shape[1]
    nbytes = lib.hd_conv3x3s2_wgrad_c32_workspace_bytes(B, Ho, Wo, K, N)
    ws = torch.empty((nbytes // 4,), dtype=_F32, device=dy.device)
    dw = torch.empty((N, K, 3, 3), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    _note_bytes("conv3x3s2_wgrad", 2 * (B * Ho * Wo * (4 * K + N) + 9 * N * K), {"flops": 18 * B * Ho * Wo * K * N, "shape": (B, Ho, Wo, K, N, 2)})
    with _ktimed("conv3x3s2_wgrad"):
        check(lib.hd_conv3x3s2_wgrad_c32(_ptr(dy), int(_cl_row_stride(dy, 8)), _ptr(x), int(_cl_row_stride(x, 8)), B, Ho, Wo, K, N,
                                         _ptr(dw), _ptr(ws), nbytes, _stream(dy)), "hd_conv3x3s2_wgrad_c32")
    return dw


def conv3x3s2_dgrad_supported(dy: torch.Tensor, x_shape, weight: torch.Tensor) -> bool:
    """hd_conv3x3s2_dgrad (csrc/conv3x3s2_dgrad.hip): the first stride-2 layer of the YOLOv5 backbone (32 <- 64 channels, dy 160
    pixels wide, the input exactly twice the output)."""
    if not (dy.is_cuda and dy.dim() == 4 and dy.dtype == torch.bfloat16 and _cl_row_stride(dy, 8) is not None
            and weight.dtype == torch.bfloat16 and weight.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)
            and weight.is_contiguous(memory_format=torch.channels_last)):
        return False
    B, N, Ho, Wo = dy.shape
    if tuple(x_shape) != (B, weight.shape[1], 2 * Ho, 2 * Wo) or weight.shape[0] != N:
        return False
    return bool(_lib.load().hd_conv3x3s2_dgrad_supported(B, Ho, Wo, int(weight.shape[1]), N))


def conv3x3s2_dgrad_rows(dy: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """dx [B, Cin, 2 Ho, 2 Wo] (bf16 channels-last) of a 3x3 / stride 2 / pad 1 convolution from dy [B, Cout, Ho, Wo] and the layer's
    own weight [Cout, Cin, 3, 3] (channels-last memory); every element written once, no autograd."""
    lib = _lib.load()
    B, N, Ho, Wo = dy.shape
    K = weight.shape[1]
    if not conv3x3s2_dgrad_supported(dy, (B, K, 2 * Ho, 2 * Wo), weight):
        raise _lib.HdError("conv3x3s2_dgrad: needs bf16 channels-last rows [B, 64, H, 160] and a channels-last bf16 [64, 32, 3, 3] weight")
    dx = torch.empty((B, K, 2 * Ho, 2 * Wo), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    _note_bytes("conv3x3s2_dgrad", 2 * (B * Ho * Wo * (N + 4 * K) + 9 * N * K), {"flops": 18 * B * Ho * Wo * K * N, "shape": (B, Ho, Wo, K, N, 2)})
    with _ktimed("conv3x3s2_dgrad"):
        check(lib.hd_conv3x3s2_dgrad(_ptr(dy), int(_cl_row_stride(dy, 8)), _ptr(weight), B, Ho, Wo, K, N, _ptr(dx), K, _stream(dy)),
              "hd_conv3x3s2_dgrad")
    return dx


class FlippedWeights:
    """The flipped, transposed copies hd_conv3x3_fwd needs as a data gradient (wt[c][ky][kx][n] = w[n][2-ky][2-kx][c]), rebuilt
    for every registered weight by ONE launch at the head of a training step (hd_conv3x3_flip_weights) instead of a flip + a
    permuting copy per layer in backward.  One instance per train.Trainer (it owns the buffers for as long as the trainer —
    and any hipGraph captured from it — lives): built from the bf16 weight shadows of the model's 3x3 / stride 1 layers,
    `refresh()` inside the step before forward (so a captured graph replays it), and `with flipped:` around forward +
    backward makes it the one _Conv3x3Fn consults; outside that window the copy is built per call, so a stale one can never
    be used."""

    _current = None

    def __init__(self, weights):
        self.items = []
        self.by_ptr = {}
        for w in weights:
            if not (w.is_cuda and w.dim() == 4 and w.dtype == torch.bfloat16):
                continue
            k = tuple(w.shape[2:])
            if k == (3, 3) and w.is_contiguous(memory_format=torch.channels_last):
                taps = 9
            elif k == (1, 1) and w.is_contiguous():   # a 1x1 layer: the plain transpose [Cin][Cout] (see pointwise_dgrad_long)
                taps = 1
            else:
                continue
            if len(self.items) >= 32:   # HD_MAX_FLIP
                break
            N, K = w.shape[0], w.shape[1]
            wt = torch.empty((K, N) + k, dtype=torch.bfloat16, device=w.device,
                             memory_format=torch.channels_last if taps == 9 else torch.contiguous_format)
            self.items.append((w, wt, taps))
            self.by_ptr[w.data_ptr()] = wt

    def __len__(self):
        return len(self.items)

    def refresh(self) -> None:
        if not self.items:
            return
        n = len(self.items)
        src = (C.c_void_p * n)(*[w.data_ptr() for w, _, _ in self.items])
        dst = (C.c_void_p * n)(*[t.data_ptr() for _, t, _ in self.items])
        cout = (C.c_int * n)(*[int(w.shape[0]) for w, _, _ in self.items])
        cin = (C.c_int * n)(*[int(w.shape[1]) for w, _, _ in self.items])
        taps = (C.c_int * n)(*[int(t) for _, _, t in self.items])
        dev = self.items[0][0].device
        check(_lib.load().hd_conv_flip_weights(n, src, dst, cout, cin, taps, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
              "hd_conv_flip_weights")

    def __enter__(self):
        self._prev, FlippedWeights._current = FlippedWeights._current, self
        return self

    def __exit__(self, *exc):
        FlippedWeights._current = self._prev
        return False

    @classmethod
    def get(cls, weight: torch.Tensor):
        cur = cls._current
        return cur.by_ptr.get(weight.data_ptr()) if cur is not None else None


class _Conv3x3Fn(torch.autograd.Function):
    """3x3 / pad 1 convolution of stride 1 or 2.  Stride 1 with a supported shape: forward on hd_conv3x3_fwd (BatchNorm
    statistics into `slot` from its epilogue) and the data gradient on the same kernel with the flipped, transposed weights;
    stride 2 on the first backbone layer's shape: the data gradient on hd_conv3x3s2_dgrad; otherwise those stay with MIOpen.
    The weight gradient runs on hd_conv3x3_wgrad wherever that takes the shape."""

    @staticmethod
    def forward(ctx, x, weight, stride, slot, ours_fwd):
        ctx.save_for_backward(x, weight)
        ctx.stride = int(stride)
        N = weight.shape[0]
        ctx.ours = (bool(ours_fwd) or slot is not None) and ctx.stride == 1 and conv3x3_fwd_supported(x, N) \
            and weight.is_contiguous(memory_format=torch.channels_last)
        if ctx.ours:
            return conv3x3_fwd_rows(x, weight, _fwd_sums(slot, N) if slot is not None else None)
        if ctx.stride == 2 and (bool(ours_fwd) or slot is not None) and conv3x3s2_fwd_supported(x, weight):
            return conv3x3s2_fwd_rows(x, weight, _fwd_sums(slot, N) if slot is not None else None)   # the first stride-2 layer
        if slot is not None:
            raise _lib.HdError("conv3x3: statistics in the epilogue need a hand-written forward (supported shape)")
        return torch.nn.functional.conv2d(x, weight, None, ctx.stride, 1)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        st = ctx.stride
        dx = dw = None
        g = dy if (dy.dtype == torch.bfloat16 and _cl_row_stride(dy, 8) is not None) else \
            dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        if ctx.needs_input_grad[0]:
            if ctx.ours and conv3x3_fwd_supported(g, x.shape[1]):
                # dx = conv3x3(dy, wt) with wt[c][ky][kx][n] = w[n][2 - ky][2 - kx][c]
                wt = FlippedWeights.get(weight)   # rebuilt for all layers by one launch at the head of the step
                if wt is None:
                    wt = weight.flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)
                dx = conv3x3_fwd_rows(g, wt, None, name="conv3x3_dgrad")
            elif st == 2 and conv3x3s2_dgrad_supported(g, x.shape, weight):
                dx = conv3x3s2_dgrad_rows(g, weight)   # the first stride-2 layer: every dx element written once, no zero-fill
            else:
                dx = torch.ops.aten.convolution_backward(dy, x, weight, None, (st, st), (1, 1), (1, 1), False, (0, 0), 1,
                                                         (True, False, False))[0]
        if ctx.needs_input_grad[1]:
            if conv3x3_wgrad_supported(x, weight.shape[0], st):
                dw = conv3x3_wgrad_rows(g, x, st, leaf=weight)
            elif st == 2 and conv3x3s2_wgrad_supported(g, x):
                dw = conv3x3s2_wgrad_rows(g, x)   # the first stride-2 layer (32 -> 64 channels at 320^2)
            else:
                dw = torch.ops.aten.convolution_backward(dy, x, weight, None, (st, st), (1, 1), (1, 1), False, (0, 0), 1,
                                                         (False, True, False))[1]
        return dx, dw, None, None, None


def conv3x3(x: torch.Tensor, weight: torch.Tensor, stride: int = 1, slot: Optional["_StatSlot"] = None,
            ours_forward: bool = False) -> torch.Tensor:
    """3x3 / pad 1 convolution whose weight gradient runs on hd_conv3x3_wgrad.  ours_forward (or a BatchNorm `slot` to fill):
    forward and data gradient on hd_conv3x3_fwd too (stride 1, supported shapes), else MIOpen's."""
    return _Conv3x3Fn.apply(x, weight, stride, slot, ours_forward)


def conv3x3_takes_stats(x: torch.Tensor, weight: torch.Tensor, stride: int) -> bool:
    """True when conv3x3(x, weight, stride, slot) runs the hand-written forward (and so can fill a BatchNorm slot)."""
    if stride == 2:
        return conv3x3s2_fwd_supported(x, weight)
    return (stride == 1 and weight.dtype == torch.bfloat16 and weight.is_contiguous(memory_format=torch.channels_last)
            and conv3x3_fwd_supported(x, weight.shape[0]))


# ------------------------------------------------------------------------------- the 6x6 / stride 2 stem (3 input channels)
def stem_conv_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """x: [B, 3, H, W] float32 or bf16, dense channels-last; weight: [Cout, 3, 6, 6] bf16 in channels-last memory."""
    if not (x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and x.dtype in (torch.float32, torch.bfloat16)
            and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0):
        return False
    if not (weight.dim() == 4 and tuple(weight.shape[1:]) == (3, 6, 6) and weight.dtype == torch.bfloat16
            and weight.is_contiguous(memory_format=torch.channels_last)):
        return False
    B, _, H, W = x.shape
    return bool(_lib.load().hd_stem_conv_supported(B, H, W, int(weight.shape[0]), 0 if x.dtype == torch.float32 else 1))


def stem_conv_rows(x: torch.Tensor, weight: torch.Tensor, stats: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y [B, Cout, H/2, W/2] bf16 channels-last = Conv2d(3, Cout, 6, 2, 2)(x) on hd_stem_conv_fwd, no autograd; stats as for
    pointwise_conv_rows."""
    lib = _lib.load()
    if not stem_conv_supported(x, weight):
        raise _lib.HdError("stem_conv: needs dense channels-last [B, 3, H, W] images (H even, W % 4 == 0) and a channels-last bf16 "
                           "[Cout in {32, 64}, 3, 6, 6] weight")
    B, _, H, W = x.shape
    N = weight.shape[0]
    y = torch.empty((B, N, H // 2, W // 2), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    _note_bytes("stem_conv_fwd", x.numel() * x.element_size() + y.numel() * 2 + weight.numel() * 2,
                {"flops": 2 * B * (H // 2) * (W // 2) * N * 108})
    with _ktimed("stem_conv_fwd"):
        check(lib.hd_stem_conv_fwd(_ptr(x), 0 if x.dtype == torch.float32 else 1, _ptr(weight), B, H, W, N, _ptr(y), N, _ptr(stats),
                                   _stream(x)), "hd_stem_conv_fwd")
    return y


def stem_conv_wgrad_rows(dy: torch.Tensor, x: torch.Tensor, leaf: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dw [Cout, 3, 6, 6] bf16 (channels-last memory) of the stem from dy [B, Cout, H/2, W/2] bf16 channels-last rows and the
    images x; completes like every chained weight gradient (_WgradChain)."""
    lib = _lib.load()
    B, _, H, W = x.shape
    N = dy.shape[1]
    ld_dy = _cl_row_stride(dy, 8)
    if ld_dy is None or dy.dtype != torch.bfloat16 or tuple(dy.shape) != (B, N, H // 2, W // 2):
        raise _lib.HdError("stem_conv_wgrad: dy must be bf16 channels-last rows [B, Cout, H/2, W/2]")
    nbytes = lib.hd_stem_conv_wgrad_workspace_bytes(B, H, W, N)
    if nbytes == 0:
        raise _lib.HdError("stem_conv_wgrad: unsupported shape")
    _WgradChain.flush_device(dy.device)   # this launch carries no piggyback: sum what is pending first
    ws = _wgrad_workspace(dy.device, nbytes)
    dw = torch.empty((N, 3, 6, 6), dtype=torch.bfloat16, device=dy.device, memory_format=torch.channels_last)
    job = _lib.WgradJob()
    _note_bytes("stem_conv_wgrad", x.numel() * x.element_size() + dy.numel() * 2, {"flops": 2 * B * (H // 2) * (W // 2) * N * 108})
    with _ktimed("stem_conv_wgrad"):
        check(lib.hd_stem_conv_wgrad_partial(_ptr(dy), int(ld_dy), _ptr(x), 0 if x.dtype == torch.float32 else 1, B, H, W, N,
                                             _ptr(dw), _ptr(ws), ws.numel(), C.byref(job), _stream(dy)),
              "hd_stem_conv_wgrad_partial")
    _WgradChain.submit(dy.device, job, (ws,), leaf)  # not dw: a second reference would make autograd copy it
    return dw


class _StemConvFn(torch.autograd.Function):
    """The stem convolution on hd_stem_conv_fwd (BatchNorm statistics into `slot` from its epilogue) / hd_stem_conv_wgrad_partial;
    the images get no gradient."""

    @staticmethod
    def forward(ctx, x, weight, slot):
        ctx.save_for_backward(x, weight)
        return stem_conv_rows(x, weight, _fwd_sums(slot, weight.shape[0]) if slot is not None else None)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        if ctx.needs_input_grad[0]:
            raise _lib.HdError("stem_conv: the input images must not require a gradient")
        dw = None
        if ctx.needs_input_grad[1]:
            g = dy if (dy.dtype == torch.bfloat16 and _cl_row_stride(dy, 8) is not None) else \
                dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
            dw = stem_conv_wgrad_rows(g, x, weight)
        return None, dw, None


def stem_conv(x: torch.Tensor, weight: torch.Tensor, slot: Optional["_StatSlot"] = None) -> torch.Tensor:
    return _StemConvFn.apply(x, weight, slot)


POINTWISE_MAX_REDUCTION = 256  # hd_pointwise_conv's streaming form; longer reductions -> pointwise_longk.hip (Cin = 512, few rows) or MIOpen


class _PointwiseConvFn(torch.autograd.Function):
    """1x1 convolution (stride 1, no bias) of channels-last bf16 rows on the MFMA kernel of csrc/pointwise_conv.hip.
    forward: y = x . w^T, and the following fused BatchNorm's batch statistics go into `slot` from the epilogue.
    backward: the data gradient is the same kernel with the transposed weights (when Cout is short enough to be the
    reduction); the weight gradient is hd_pointwise_wgrad (channel counts that are multiples of 64), else MIOpen's."""

    @staticmethod
    def forward(ctx, x, weight, slot):
        N, K = weight.shape[0], weight.shape[1]
        if slot is not None or pointwise_conv_usable(x, weight):
            sums = _fwd_sums(slot, N) if slot is not None else None
            y = pointwise_conv_rows(x, weight.view(N, K), sums)
        else:  # a reduction too long for the resident-weight kernel: MIOpen's forward, our backward
            xd = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            y = torch.nn.functional.conv2d(xd, weight)
        ctx.save_for_backward(x, weight)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        N, K = weight.shape[0], weight.shape[1]
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if dy.dtype != torch.bfloat16 or _cl_row_stride(dy, 8) is None:
            dy = dy.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        dx = dw = None
        R = x.shape[0] * x.shape[2] * x.shape[3]
        ours = need_dx and bool(_lib.load().hd_pointwise_conv_dgrad_supported(R, K, N))
        ours_w = need_dw and pointwise_wgrad_supported(R, K, N) and _cl_row_stride(x, 8) is not None
        if ours and ours_w and pointwise_bwd_supported(R, K, N):   # both gradients from one staging of dy
            dx, dw = pointwise_bwd_rows(dy, weight.view(N, K), x, leaf=weight)
            return dx, dw.view(N, K, 1, 1), None
        if ours:
            dx = pointwise_dgrad_rows(dy, weight.view(N, K))
        if ours_w:
            dw = pointwise_wgrad_rows(dy, x, leaf=weight).view(N, K, 1, 1)
        if need_dx and not ours:
            dx = pointwise_dgrad_long(dy, weight)
        lib_w, lib_x = need_dw and not ours_w, need_dx and dx is None
        if lib_w or lib_x:
            xd = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            dyd = dy if dy.is_contiguous(memory_format=torch.channels_last) else dy.contiguous(memory_format=torch.channels_last)
            gi, gw, _ = torch.ops.aten.convolution_backward(dyd, xd, weight, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                            (lib_x, lib_w, False))
            if lib_w:
                dw = gw
            if lib_x:
                dx = gi
        return dx, dw, None


def pointwise_dgrad_long(dy: torch.Tensor, weight: torch.Tensor) -> Optional[torch.Tensor]:
    """Data gradient of a 1x1 layer whose Cout is too long for hd_pointwise_conv_dgrad's transpose fragments (the five Cout = 512
    layers of YOLOv5s): dx = hd_pointwise_conv(dy, w^T) on the long-reduction kernel (csrc/pointwise_longk.hip), with w^T the copy
    the step's one flip launch rebuilt (FlippedWeights, taps = 1).  None when no such copy is registered or the shape is not
    taken — the caller then asks the library (which also needs a zero-fill of dx)."""
    wt = FlippedWeights.get(weight)
    if wt is None or dy.dtype != torch.bfloat16 or _cl_row_stride(dy, 8) is None:
        return None
    N, K = weight.shape[0], weight.shape[1]
    R = dy.shape[0] * dy.shape[2] * dy.shape[3]
    if not _lib.load().hd_pointwise_conv_resident(R, N, K):
        return None
    return pointwise_conv_rows(dy, wt.view(K, N), None, name="pointwise_dgrad")


def pointwise_conv(x: torch.Tensor, weight: torch.Tensor, slot: Optional["_StatSlot"] = None) -> torch.Tensor:
    """Differentiable 1x1 convolution; with `slot` (a bn_act_slot of Cout channels) the BatchNorm statistics of the
    result are accumulated there, and the bn_act that follows is called with stats_done=True."""
    return _PointwiseConvFn.apply(x, weight, slot)


def pointwise_conv_usable(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """The forward kernel is preferred to the library where it runs in its streaming form (Cin <= 256, <= 8 column tiles)."""
    if not pointwise_conv_supported(x, weight):
        return False
    return bool(_lib.load().hd_pointwise_conv_resident(x.shape[0] * x.shape[2] * x.shape[3], weight.shape[1], weight.shape[0]))


def pointwise_backward_usable(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """The backward kernels alone are worth taking (weight gradient without zero-fill / cast launches) even where the
    forward stays with the library: channel counts that are multiples of 64, bf16 channels-last rows."""
    if not (x.is_cuda and x.dim() == 4 and x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and weight.dim() == 4
            and weight.shape[2:] == (1, 1) and weight.shape[1] == x.shape[1] and weight.is_contiguous()):
        return False
    B, K, H, W = x.shape
    return pointwise_wgrad_supported(B * H * W, K, weight.shape[0]) and _cl_row_stride(x, 8) is not None


class _BNState:
    """What the fused BatchNorm + activation forward leaves for its backward (shared by _BNActFn and _ConvBNActFn)."""
    __slots__ = ("dt", "R", "C", "act", "slot", "sync_group", "stat_rows", "res_mode", "split", "tag", "w", "b", "save", "res")


def _bn_forward(x, weight, bias, running_mean, running_var, eps, momentum, act, slot, sync_group, res, res_mode, out_slot,
                stats_done):
    """Statistics (unless the producer of x already left them in the slot) + apply.  -> (y, y2 | None, _BNState)."""
    lib = _lib.load()
    if res is not None:
        if res.shape != x.shape or res.dtype != x.dtype or res.stride() != x.stride():
            res = res.to(x.dtype).contiguous(memory_format=torch.channels_last)
            if res.shape != x.shape:
                raise _lib.HdError("bn_act: the residual must have the shape of the normalised tensor")
    B, Cc, H, W = x.shape
    R = B * H * W
    dt = 0 if x.dtype == torch.float32 else 1
    sums = slot.buf[: lib.hd_bn_act_replicas() * 2 * Cc] if stats_done else _fwd_sums(slot, Cc)
    vec = 4 if dt == 0 else 8
    split = 0
    y2, tail = None, None
    if out_slot is not None and isinstance(out_slot.t, tuple):
        # two destinations (merged C3 branch convolutions): channels [0, split) -> y, [split, C) -> y2
        y, y2 = out_slot.t
        split = y.shape[1]
        ld_y, ld_y2 = _cl_row_stride(y, vec), _cl_row_stride(y2, vec)
        if (ld_y is None or ld_y2 is None or y.shape[1] + y2.shape[1] != Cc or y.dtype != x.dtype or y2.dtype != x.dtype
                or split % vec or y.shape[0] != B or y.shape[2:] != x.shape[2:] or y2.shape[2:] != x.shape[2:]):
            raise _lib.HdError("bn_act: split outputs must be channels-last tensors / channel slices covering x's channels")
        tail = Rows2(y2.data_ptr(), int(ld_y2), int(split))
    else:
        y = out_slot.t if out_slot is not None else torch.empty_like(x)
        ld_y = _cl_row_stride(y, vec)
        if ld_y is None or y.shape != x.shape or y.dtype != x.dtype:
            raise _lib.HdError("bn_act: `out` must be a channels-last tensor (or channel slice) of x's shape and dtype")
    save = torch.empty((2, Cc), dtype=_F32, device=x.device)
    st = _stream(x)
    w, b = weight.detach().float(), bias.detach().float()
    tb = x.numel() * x.element_size()
    tag = (tuple(x.shape), int(act), int(res_mode) if res is not None else 0, bool(stats_done))
    if not stats_done:
        _note_bytes("bn_stats_kernel", tb, tag)
        with _ktimed("bn_stats_kernel"):
            check(lib.hd_bn_act_stats(_ptr(x), dt, R, Cc, _ptr(sums), st), "hd_bn_act_stats")
    stat_rows = 0
    if sync_group is not None:
        # synchronised BatchNorm: the per-channel (sum, sum of squares) of every rank are added between the
        # statistics kernel and the apply kernel; ranks hold equal shards, so the sums cover R * world rows
        import torch.distributed as dist
        dist.all_reduce(sums, group=sync_group)
        stat_rows = R * dist.get_world_size(sync_group)
    _note_bytes("bn_apply_kernel", (3 if res is not None else 2) * tb, tag)
    with _ktimed("bn_apply_kernel"):
        check(lib.hd_bn_act_apply(_ptr(x), dt, R, stat_rows, Cc, _ptr(sums), _ptr(w), _ptr(b), float(eps), float(momentum),
                                  _ptr(running_mean), _ptr(running_var), _ptr(save[0]), _ptr(save[1]), int(act),
                                  _ptr(res), int(res_mode) if res is not None else 0, _ptr(y), int(ld_y),
                                  C.byref(tail) if tail is not None else None, st), "hd_bn_act_apply")
    s_ = _BNState()
    s_.dt, s_.R, s_.C, s_.act, s_.slot, s_.sync_group, s_.stat_rows = dt, R, Cc, int(act), slot, sync_group, stat_rows
    s_.res_mode, s_.split, s_.tag = (0 if res is None else int(res_mode)), split, tag
    # the backward kernels need the residual only when it was added BEFORE the activation (mode 1)
    s_.w, s_.b, s_.save, s_.res = w, b, save, (res if (res is not None and int(res_mode) == 1) else None)
    return y, y2, s_


class _BNGrad:
    """The incoming gradient of a fused BatchNorm + activation as the kernels take it: (a channel slice of) a channels-last
    tensor, possibly in two places (the merged C3 branch convolutions)."""
    __slots__ = ("dy", "ld", "dy2", "tail", "dsums")


def _bn_backward_reduce(st: "_BNState", x, dy, dy2) -> "_BNGrad":
    """Per-channel sums of the backward (bn_bwd_reduce_kernel) into the slot's accumulators."""
    lib = _lib.load()
    vec = 4 if st.dt == 0 else 8
    Cc = st.C

    def prep(g, channels):
        """-> (tensor kept alive, row stride): a (channel slice of a) channels-last tensor, copied once otherwise."""
        if g is None:  # an unused half: zero gradient
            g = torch.zeros((x.shape[0], channels, x.shape[2], x.shape[3]), dtype=x.dtype, device=x.device,
                            memory_format=torch.channels_last)
        if g.dtype != x.dtype:
            g = g.to(x.dtype)
        ld_ = _cl_row_stride(g, vec)
        if ld_ is None:
            g = g.contiguous(memory_format=torch.channels_last)
            ld_ = channels
        return g, ld_

    g_ = _BNGrad()
    g_.tail, g_.dy2 = None, None
    if st.split:
        g_.dy, g_.ld = prep(dy, st.split)
        g_.dy2, ld2 = prep(dy2, Cc - st.split)
        g_.tail = Rows2(g_.dy2.data_ptr(), int(ld2), int(st.split))
    else:
        g_.dy, g_.ld = prep(dy, Cc)
    tail_ref = C.byref(g_.tail) if g_.tail is not None else None
    rep = lib.hd_bn_act_replicas()
    slot = st.slot
    dsums = slot.buf[rep * 2 * Cc: rep * 4 * Cc]
    if slot.bwd_gen == slot.arena.gen:
        dsums.zero_()
    slot.bwd_gen = slot.arena.gen
    g_.dsums = dsums
    tb = x.numel() * x.element_size()
    _note_bytes("bn_bwd_reduce_kernel", (3 if st.res is not None else 2) * tb, st.tag)  # x, dy (+ the pre-activation residual)
    with _ktimed("bn_bwd_reduce_kernel"):
        check(lib.hd_bn_act_bwd_reduce(_ptr(x), _ptr(g_.dy), int(g_.ld), st.dt, st.R, Cc, _ptr(st.save[0]), _ptr(st.save[1]),
                                       _ptr(st.w), _ptr(st.b), st.act, _ptr(st.res), _ptr(dsums), tail_ref, _stream(x)),
              "hd_bn_act_bwd_reduce")
    return g_


def _bn_backward_dx(st: "_BNState", x, g_: "_BNGrad"):
    """The elementwise half (bn_bwd_dx_kernel): -> (dx, dgamma, dbeta, d(residual) | None)."""
    lib = _lib.load()
    Cc = st.C
    rep = lib.hd_bn_act_replicas()
    local = None
    if st.sync_group is not None:
        # dgamma / dbeta stay this rank's own sums (the gradient all-reduce averages them like every other
        # parameter gradient); dx needs the sums over the whole synchronised batch
        import torch.distributed as dist
        local = g_.dsums.view(rep, 2, Cc).sum(0)
        dist.all_reduce(g_.dsums, group=st.sync_group)
    dx = torch.empty_like(x)
    dg = torch.empty((2, Cc), dtype=_F32, device=x.device)
    dres = torch.empty_like(x) if st.res is not None else None
    tb = x.numel() * x.element_size()
    _note_bytes("bn_bwd_dx_kernel", (5 if st.res is not None else 3) * tb, st.tag)  # x, dy -> dx (+ residual in, d(res) out)
    with _ktimed("bn_bwd_dx_kernel"):
        check(lib.hd_bn_act_bwd_dx(_ptr(x), _ptr(g_.dy), int(g_.ld), st.dt, st.R, st.stat_rows, Cc, _ptr(st.save[0]),
                                   _ptr(st.save[1]), _ptr(st.w), _ptr(st.b), st.act, _ptr(st.res), _ptr(g_.dsums), _ptr(dx),
                                   _ptr(dres), _ptr(dg[0]), _ptr(dg[1]), C.byref(g_.tail) if g_.tail is not None else None,
                                   _stream(x)), "hd_bn_act_bwd_dx")
    # d(residual): dz (written by the dx kernel) in mode 1, the incoming gradient itself in mode 2
    gres = dres if st.res_mode == 1 else (g_.dy if st.res_mode == 2 else None)
    if local is not None:
        return dx, local[1], local[0], gres
    return dx, dg[0], dg[1], gres


class _BNActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, act, slot, sync_group=None, res=None,
                res_mode=0, out_slot=None, stats_done=False):
        """stats_done: the producer of x (pointwise_conv) already added x's per-channel sums into the slot."""
        y, y2, st = _bn_forward(x, weight, bias, running_mean, running_var, eps, momentum, act, slot, sync_group, res, res_mode,
                                out_slot, stats_done)
        ctx.save_for_backward(x)
        ctx.st = st
        return y if y2 is None else (y, y2)

    @staticmethod
    def backward(ctx, dy, dy2=None):
        (x,) = ctx.saved_tensors
        st = ctx.st
        g_ = _bn_backward_reduce(st, x, dy, dy2)
        dx, dgamma, dbeta, gres = _bn_backward_dx(st, x, g_)
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, gres, None, None, None


class _PwConvBNActFn(torch.autograd.Function):
    """1x1 convolution -> BatchNorm (batch statistics) -> activation [-> + residual] as ONE autograd node: its backward runs
    bn_bwd_reduce, bn_bwd_dx and then the convolution's data + weight gradient in one launch (hd_pointwise_bwd), with a
    shortcut's gradient added in that kernel's epilogue (`fork`).  (Folding bn_bwd_dx into the gradient kernels' operand load
    was built in round 3 — bit-identical, slower — and removed in round 5: tools/experiments/r05_removed_paths.patch.)"""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, running_mean, running_var, eps, momentum, act, slot, res, res_mode, out_slot, ours_fwd,
                fork):
        """fork: also return an alias of x (last output).  A shortcut that takes THAT tensor instead of x sends its gradient
        through this node, whose data-gradient kernel adds it in its epilogue (hd_pointwise_conv_dgrad_add) — otherwise the
        autograd engine sums the two gradients of x with a launch of its own."""
        N, K = weight.shape[0], weight.shape[1]
        if ours_fwd:   # the epilogue leaves the batch statistics in the slot
            y_raw = pointwise_conv_rows(x, weight.view(N, K), _fwd_sums(slot, N))
        else:          # a reduction too long for the resident-weight kernel: the library's forward
            xd = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            y_raw = torch.nn.functional.conv2d(xd, weight)
        y, y2, st = _bn_forward(y_raw, gamma, beta, running_mean, running_var, eps, momentum, act, slot, None, res, res_mode,
                                out_slot, bool(ours_fwd))
        ctx.save_for_backward(x, weight, y_raw)
        ctx.st = st
        ctx.fork = bool(fork)
        outs = (y,) if y2 is None else (y, y2)
        if fork:
            outs = outs + (x.view_as(x),)
        return outs[0] if len(outs) == 1 else outs

    @staticmethod
    def backward(ctx, *grads):
        x, weight, y_raw = ctx.saved_tensors
        st = ctx.st
        d_skip = None
        if ctx.fork:
            grads, d_skip = grads[:-1], grads[-1]
        dout, dout2 = grads[0], (grads[1] if len(grads) > 1 else None)
        if d_skip is not None and (d_skip.dtype != torch.bfloat16 or _cl_row_stride(d_skip, 8) is None or d_skip.shape != x.shape):
            d_skip = d_skip.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        N, K = weight.shape[0], weight.shape[1]
        R = x.shape[0] * x.shape[2] * x.shape[3]
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        lib = _lib.load()
        g_ = _bn_backward_reduce(st, y_raw, dout, dout2)
        ours_dx = bool(lib.hd_pointwise_conv_dgrad_supported(R, K, N))
        ours_dw = pointwise_wgrad_supported(R, K, N) and _cl_row_stride(x, 8) is not None
        dy, dgamma, dbeta, gres = _bn_backward_dx(st, y_raw, g_)
        dx = dw = None
        if need_dw and ours_dw and need_dx and ours_dx and pointwise_bwd_supported(R, K, N):
            # one launch: data gradient (+ the shortcut's gradient in its epilogue) and the weight-gradient partials
            dx, dw = pointwise_bwd_rows(dy, weight.view(N, K), x, add=d_skip, leaf=weight)
            dw = dw.view(N, K, 1, 1)
            d_skip = None
        if need_dw and ours_dw and dw is None:
            dw = pointwise_wgrad_rows(dy, x, leaf=weight).view(N, K, 1, 1)
        if need_dx and ours_dx and dx is None:
            dx = pointwise_dgrad_rows(dy, weight.view(N, K), add=d_skip)   # the shortcut's gradient rides in the epilogue
            d_skip = None
        if need_dx and dx is None:
            dx = pointwise_dgrad_long(dy, weight)
        lib_w, lib_x = need_dw and dw is None, need_dx and dx is None
        if lib_w or lib_x:
            xd = x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)
            gi, gw, _ = torch.ops.aten.convolution_backward(dy, xd, weight, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1,
                                                            (lib_x, lib_w, False))
            dw = gw if lib_w else dw
            dx = gi if lib_x else dx
        if dx is not None and d_skip is not None:
            dx = dx + d_skip
        return dx, dw, dgamma, dbeta, None, None, None, None, None, None, gres, None, None, None, None


def pointwise_conv_bn_act(x, weight, gamma, beta, running_mean, running_var, eps: float, momentum: float, act: str,
                          slot: "_StatSlot", residual=None, residual_after_act: bool = False, out=None, ours_forward: bool = True,
                          fork: bool = False):
    """conv1x1(x, weight) -> BatchNorm2d (training) -> activation [+ residual] on the HIP kernels, one autograd node (see
    _PwConvBNActFn).  x: bf16 channels-last rows, weight: bf16 [Cout, Cin, 1, 1]; out as for bn_act.
    fork: returns (result, x_alias) — hand x_alias to the shortcut that bypasses this layer."""
    return _PwConvBNActFn.apply(x, weight, gamma, beta, running_mean, running_var, eps, momentum, _ACT_CODE[act], slot, residual,
                                (2 if residual_after_act else 1) if residual is not None else 0,
                                _OutSlot(out) if out is not None else None, bool(ours_forward), bool(fork))


class _BiasActFn(torch.autograd.Function):
    """y = act(x + bias[c]) for a biased convolution that was run without its bias (csrc/bn_act.hip: hd_bias_act_fwd / _bwd).
    Backward works from the output: dx = dy * (y > 0) for ReLU (dy itself for no activation) and the bias gradient is the
    per-channel sum of dx taken in the same pass."""

    @staticmethod
    def forward(ctx, x, bias, code: int, inplace: bool):
        lib = _lib.load()
        B, Cc, H, W = x.shape
        y = x if inplace else torch.empty_like(x, memory_format=torch.channels_last)
        b32 = bias if bias.dtype == torch.float32 else bias.float()
        check(lib.hd_bias_act_fwd(_ptr(x), 0 if x.dtype == torch.float32 else 1, B * H * W, Cc, _ptr(b32), code, _ptr(y),
                                  _stream(x)), "hd_bias_act_fwd")
        if inplace:
            ctx.mark_dirty(x)
        ctx.code, ctx.bias_dtype = code, bias.dtype
        ctx.save_for_backward(y if code != 0 else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        if not dy.is_contiguous(memory_format=torch.channels_last):
            dy = dy.contiguous(memory_format=torch.channels_last)
        B, Cc, H, W = dy.shape
        rep = lib.hd_bn_act_replicas()
        dsums = torch.zeros(rep * 2 * Cc, dtype=torch.float32, device=dy.device)
        dx = torch.empty_like(dy, memory_format=torch.channels_last) if ctx.code != 0 else None
        check(lib.hd_bias_act_bwd(_ptr(y), _ptr(dy), 0 if dy.dtype == torch.float32 else 1, B * H * W, Cc, ctx.code, _ptr(dx),
                                  _ptr(dsums), _stream(dy)), "hd_bias_act_bwd")
        db = dsums.view(rep, 2, Cc)[:, 0].sum(0)
        return (dx if dx is not None else dy), (db if ctx.bias_dtype == torch.float32 else db.to(ctx.bias_dtype)), None, None


def bias_act_supported(x: torch.Tensor, act) -> bool:
    return act in (None, "none", "relu") and bn_act_supported(x)


def bias_act(x: torch.Tensor, bias: torch.Tensor, act=None, inplace: bool = False) -> torch.Tensor:
    """act(x + bias[None, :, None, None]) on a dense channels-last float32 / bfloat16 HIP tensor, act None or "relu"; bias float32
    (or x's dtype).  inplace: write over x (x must not be needed by the backward of whatever produced it — a convolution's
    output is not)."""
    if not bias_act_supported(x, act):
        raise _lib.HdError("bias_act: needs a dense channels-last float32/bfloat16 HIP tensor with C % vec == 0, act None or relu")
    if bias.dim() != 1 or bias.shape[0] != x.shape[1] or not bias.is_cuda:
        raise _lib.HdError("bias_act: bias must be a HIP vector of C elements")
    return _BiasActFn.apply(x, bias, _ACT_CODE[act], bool(inplace))


def bn_act_slot(device, channels: int) -> _StatSlot:
    # forward sums + backward sums, each [replicas, 2, C]
    return _arena(torch.device(device)).alloc(4 * channels * _lib.load().hd_bn_act_replicas())


def bn_act(x, weight, bias, running_mean, running_var, eps: float, momentum: float, act: str, training: bool,
           slot: Optional[_StatSlot] = None, sync_group=None, residual: Optional[torch.Tensor] = None,
           residual_after_act: bool = False, out: Optional[torch.Tensor] = None, stats_done: bool = False) -> torch.Tensor:
    """Fused BatchNorm2d + activation on a dense channels-last HIP tensor (see csrc/bn_act.hip).
    sync_group: a torch.distributed process group -> statistics (and the backward sums) are taken over the
    batches of all its ranks (SyncBN; ranks must hold equal shards).  Training mode only, not graph-capturable.
    residual: a tensor shaped like x that is added inside the apply kernel — before the activation
    (`act(bn(x) + r)`, the ResNet bottleneck tail) or after it (`act(bn(x)) + r`, the YOLOv5 shortcut).
    out: a tensor to write the result into, e.g. a CatBuffer slot (a channel slice of a channels-last buffer); or a
    pair (y_first, y_second) of such tensors that split the channels between them (returns the pair)."""
    if not bn_act_supported(x):
        raise _lib.HdError("bn_act: needs a dense channels-last float32/bfloat16 HIP tensor with C % vec == 0")
    code = _ACT_CODE[act]
    if training:
        if slot is None:
            slot = bn_act_slot(x.device, x.shape[1])
        return _BNActFn.apply(x, weight, bias, running_mean, running_var, eps, momentum, code, slot, sync_group, residual,
                              (2 if residual_after_act else 1) if residual is not None else 0,
                              _OutSlot(out) if out is not None else None, bool(stats_done))
    lib = _lib.load()
    B, Cc, H, W = x.shape
    vec = 4 if x.dtype == torch.float32 else 8
    tail, y2 = None, None
    if isinstance(out, tuple):  # two destinations (merged C3 branch convolutions); no residual in that layer
        if residual is not None:
            raise _lib.HdError("bn_act: split outputs and a residual are not combined")
        y, y2 = out
        ld_y, ld_y2 = _cl_row_stride(y, vec), _cl_row_stride(y2, vec)
        if ld_y is None or ld_y2 is None or y.shape[1] + y2.shape[1] != Cc or y.shape[1] % vec:
            raise _lib.HdError("bn_act: split outputs must be channels-last tensors / channel slices covering x's channels")
        tail = Rows2(y2.data_ptr(), int(ld_y2), int(y.shape[1]))
    else:
        direct = out is not None and residual is None  # inference with a residual finishes with torch ops below
        y = out if direct else torch.empty_like(x)
        ld_y = _cl_row_stride(y, vec)
        if ld_y is None:
            raise _lib.HdError("bn_act: `out` must be a channels-last tensor or channel slice")
    fused_act = code if (residual is None or residual_after_act) else 0
    check(lib.hd_bn_act_infer(_ptr(x), 0 if x.dtype == torch.float32 else 1, B * H * W, Cc, _ptr(weight.detach().float()),
                              _ptr(bias.detach().float()), _ptr(running_mean), _ptr(running_var), float(eps), fused_act,
                              _ptr(y), int(ld_y), C.byref(tail) if tail is not None else None, _stream(x)), "hd_bn_act_infer")
    if y2 is not None:
        return y, y2
    if residual is not None:  # inference: the add (and a pre-add activation) stay plain torch ops
        y = y + residual
        if not residual_after_act:
            y = {1: torch.nn.functional.silu, 2: torch.relu, 0: (lambda t: t)}[code](y)
        if out is not None:
            out.copy_(y)
            y = out
    return y



def time_bn_act_kernels(shape, dtype=torch.bfloat16, act: str = "silu", iters: int = 20, device=None,
                        residual_after_act: bool = False):
    """Measurement helper for bench.py: the four fused-BN kernels on one channels-last tensor of `shape`
    ([B,C,H,W]), each launched `iters` times on the current stream, every launch stamping its own HIP event pair
    with the kernel's begin / end (hd_profile_time_next_launch), i.e. the kernel alone, without host launch gaps.
    Returns {kernel: (median ms, algorithmic bytes)}.
    residual_after_act: the apply kernel also reads a residual tensor (y = act(bn(x)) + r, the YOLOv5 shortcut;
    the backward kernels are the plain ones in that mode)."""
    lib = _lib.load()
    device = torch.device(device or "cuda")
    x = torch.randn(shape, device=device).to(dtype).contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x)
    B, Cc, H, W = shape
    R = B * H * W
    dt = 0 if dtype == torch.float32 else 1
    rep = lib.hd_bn_act_replicas()
    sums = torch.zeros(rep * 4 * Cc, dtype=_F32, device=device)
    dsums = sums[rep * 2 * Cc:]
    w = torch.ones(Cc, dtype=_F32, device=device)
    b = torch.zeros(Cc, dtype=_F32, device=device)
    save = torch.zeros((2, Cc), dtype=_F32, device=device)
    y = torch.empty_like(x)
    res = torch.randn_like(x) if residual_after_act else None
    dg = torch.empty((2, Cc), dtype=_F32, device=device)
    st = _stream(x)
    code = _ACT_CODE[act]
    tb = x.numel() * x.element_size()

    def run(fn):
        for _ in range(3):
            fn()
        evs = []
        for _ in range(iters):
            evs.append(_arm_kernel_timing())  # the launch stamps the pair with the kernel's own begin / end
            fn()
        torch.cuda.synchronize(device)
        ts = sorted(a.elapsed_time(e) for a, e in evs)
        return ts[len(ts) // 2]

    out = {}
    out["bn_stats_kernel"] = (run(lambda: check(lib.hd_bn_act_stats(_ptr(x), dt, R, Cc, _ptr(sums), st), "stats")), tb)
    sums.zero_()
    check(lib.hd_bn_act_stats(_ptr(x), dt, R, Cc, _ptr(sums), st), "stats")
    out["bn_apply_kernel"] = (run(lambda: check(lib.hd_bn_act_apply(
        _ptr(x), dt, R, 0, Cc, _ptr(sums), _ptr(w), _ptr(b), 1e-3, 0.03, None, None, _ptr(save[0]), _ptr(save[1]), code,
        _ptr(res), 2 if res is not None else 0, _ptr(y), 0, None, st), "apply")), (3 if res is not None else 2) * tb)
    out["bn_bwd_reduce_kernel"] = (run(lambda: check(lib.hd_bn_act_bwd_reduce(
        _ptr(x), _ptr(dy), 0, dt, R, Cc, _ptr(save[0]), _ptr(save[1]), _ptr(w), _ptr(b), code, None, _ptr(dsums), None, st), "reduce")), 2 * tb)
    out["bn_bwd_dx_kernel"] = (run(lambda: check(lib.hd_bn_act_bwd_dx(
        _ptr(x), _ptr(dy), 0, dt, R, 0, Cc, _ptr(save[0]), _ptr(save[1]), _ptr(w), _ptr(b), code, None, _ptr(dsums), _ptr(y),
        None, _ptr(dg[0]), _ptr(dg[1]), None, st), "dx")), 3 * tb)
    return out


__all__ = [n_ for n_ in dir() if not n_.startswith("__")]   # private helpers included: the sibling modules and the tests use them
