"""Conv building blocks shared by the backbones and necks (plain torch.nn; the convolutions run
on MIOpen/MFMA through PyTorch-ROCm as BASELINE.json's north_star prescribes)."""
from __future__ import annotations

import torch
import torch.nn as nn


def autopad(k: int, p=None) -> int:
    return k // 2 if p is None else p


class LowpWeights:
    """Switch + bookkeeping for low-precision weight shadows (train.Trainer turns `on` around its own
    forward/backward only, so an evaluation or a hand-rolled forward never sees a stale shadow)."""

    on = False


class ShadowConv2d(nn.Conv2d):
    """nn.Conv2d whose fp32 master weight can be shadowed by a bf16 copy (`weight_lp`, a plain leaf
    tensor owned by train.LowpShadows).  Under autocast a plain Conv2d casts its weight fp32->bf16 on
    every call and autograd casts the bf16 weight gradient back: ~140 launches of ~6 us per YOLOv5s
    step, 0.8 ms of a 9.7 ms step once the step replays from a hipGraph.  With a shadow the cast is
    one multi-tensor launch per step for all convolutions and the weight gradients stay bf16 until
    dist.GradReducer packs them into its fp32 buckets (again multi-tensor)."""

    weight_lp = None

    def lp_weight(self) -> torch.Tensor:
        return self.weight_lp if (LowpWeights.on and self.weight_lp is not None) else self.weight

    def forward(self, x):
        w = self.lp_weight()
        if (self.in_channels <= NO_SEARCH_MAX_CHANNELS and self.out_channels <= NO_SEARCH_MAX_CHANNELS and x.is_cuda
                and torch.backends.cudnn.benchmark and self.padding_mode == "zeros"):
            return _LibConvNoSearch.apply(x, w, self.bias, self.stride, self.padding, self.dilation, self.groups)
        return self._conv_forward(x, w, self.bias)


# Library convolutions whose input AND output channel counts are at most this run outside MIOpen's solver search.
# Root cause of the round-4 / round-5 "Memory access fault by GPU" (profiles/r05_fault_hunt.md): with
# torch.backends.cudnn.benchmark = True PyTorch calls miopenFindConvolution*Algorithm with exhaustiveSearch = true, and for a
# problem without a perf-db record MIOpen then TUNES every applicable solver on the caller's tensors.  For YOLOv5n's
# Conv2d(16, 16, 1) on a [2, 16, 40, 40] bf16 channels-last map the backward-data tuning of
# ConvAsmImplicitGemmGTCDynamicBwdXdlopsNHWC falls back to its "spare" set of 22 kernel configurations, one of which reads
# past the end of an operand; it faults when that operand is the last block of a 2 MiB allocator segment (which is why the
# same test alone passes).  No hand-written kernel is involved.  Such layers are a few microseconds either way, so they take
# the library's immediate-mode choice instead.
NO_SEARCH_MAX_CHANNELS = 16


class _LibConvNoSearch(torch.autograd.Function):
    """F.conv2d / convolution_backward with MIOpen's solver search switched off for the duration of the two calls."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, padding, dilation, groups):
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            xc, wc = x.to(dt), w.to(dt)
            bc = bias.to(dt) if bias is not None else None
        else:
            xc, wc, bc = x, w, bias
        ctx.save_for_backward(xc, wc)
        ctx.conf = (tuple(stride), tuple(padding), tuple(dilation), int(groups), x.dtype, w.dtype,
                    None if bias is None else bias.dtype)
        was = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = False
        try:
            with torch.autocast("cuda", enabled=False):
                return torch.nn.functional.conv2d(xc, wc, bc, stride, padding, dilation, groups)
        finally:
            torch.backends.cudnn.benchmark = was

    @staticmethod
    def backward(ctx, gy):
        xc, wc = ctx.saved_tensors
        stride, padding, dilation, groups, xdt, wdt, bdt = ctx.conf
        mask = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], bdt is not None and ctx.needs_input_grad[2])
        was = torch.backends.cudnn.benchmark
        torch.backends.cudnn.benchmark = False
        try:
            gx, gw, gb = torch.ops.aten.convolution_backward(gy.to(xc.dtype), xc, wc, [wc.shape[0]] if bdt is not None else None,
                                                             stride, padding, dilation, False, (0, 0), groups, mask)
        finally:
            torch.backends.cudnn.benchmark = was
        return (gx.to(xdt) if gx is not None else None, gw.to(wdt) if gw is not None else None,
                gb.to(bdt) if gb is not None else None, None, None, None, None)


class FusedBNAct:
    """Mixin-free helper: `apply(bn, x, act)` runs BatchNorm2d + activation as the fused HIP kernels
    (csrc/bn_act.hip) when `x` is a dense channels-last HIP tensor and `enabled` is set, otherwise the
    plain torch modules.  One statistics slot per BatchNorm2d module is kept in `_slots`."""

    enabled = True
    cat_in_place = True  # producers of a concatenation write into its buffer (ops.CatBuffer) instead of torch.cat copying
    # SyncBN (optional second exchange): set to a torch.distributed process group and every fused
    # BatchNorm takes its batch statistics over the shards of all ranks in the group.  Eager mode only: the
    # all-reduce sits between two kernels of the layer, so such a step is not captured into a hipGraph.
    sync_group = None
    _ACTS = {"silu": torch.nn.functional.silu, "relu": torch.relu, None: (lambda t: t)}

    @staticmethod
    def slot_of(bn: nn.BatchNorm2d, device):
        slot = getattr(bn, "_hd_slot", None)
        if slot is None and bn.training:
            from . import ops
            slot = ops.bn_act_slot(device, bn.num_features)
            bn._hd_slot = slot
        return slot

    @staticmethod
    def apply(bn: nn.BatchNorm2d, x: torch.Tensor, act, residual=None, residual_after_act: bool = False, out=None,
              stats_done: bool = False):
        """residual: added inside the fused kernel, before (`act(bn(x) + r)`) or after (`act(bn(x)) + r`) the
        activation; the unfused fallback computes the same expression with torch ops.
        out: a CatBuffer slot — the fused kernel writes its result there (the caller's CatBuffer.join then has nothing
        to copy); the unfused fallback ignores it and returns an ordinary tensor (join copies it in, like torch.cat)."""
        # momentum=None (cumulative moving average) needs the step counter: that case stays on torch's BatchNorm
        if FusedBNAct.enabled and x.is_cuda and bn.momentum is not None:
            from . import ops
            if ops.bn_act_supported(x):
                slot = FusedBNAct.slot_of(bn, x.device)
                training = bn.training or bn.running_mean is None
                return ops.bn_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, act, training, slot,
                                  FusedBNAct.sync_group if training else None, residual, residual_after_act, out, stats_done)
        if stats_done:
            raise RuntimeError("FusedBNAct.apply(stats_done=True) needs the fused HIP path")
        if residual is None:
            return FusedBNAct._ACTS[act](bn(x))
        if residual_after_act:
            return FusedBNAct._ACTS[act](bn(x)) + residual
        return FusedBNAct._ACTS[act](bn(x) + residual)


def cat_buffer(like: torch.Tensor, channels, hw=None):
    """An ops.CatBuffer for a channel concatenation whose parts come out of fused BN+act kernels, or None when the
    fused HIP path is not in play (CPU tensors, fusion switched off, layouts the kernels do not take): callers then
    fall back to torch.cat.  `like` supplies batch size, dtype (autocast's under autocast) and device."""
    if not (ConvBNAct.fused and FusedBNAct.enabled and like.is_cuda and FusedBNAct.cat_in_place):
        return None
    dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else like.dtype
    if dtype not in (torch.float32, torch.bfloat16):
        return None
    vec = 4 if dtype == torch.float32 else 8
    if any(int(c) % vec for c in channels) or any(int(c) // vec > 256 for c in channels):
        return None
    from . import ops
    H, W = hw if hw is not None else like.shape[2:]
    return ops.CatBuffer(_Like(like.shape[0], int(H), int(W), dtype, like.device), channels)


class _Like:
    """Shape / dtype / device carrier for ops.CatBuffer (which only reads .shape, .dtype, .device)."""

    def __init__(self, B, H, W, dtype, device):
        self.shape, self.dtype, self.device = (B, 0, H, W), dtype, device


class ConvBNAct(nn.Module):
    """Conv2d(bias=False) + BatchNorm2d + activation (SiLU for the YOLOv5 family, ReLU for ResNet)."""

    fused = True  # class-wide switch: run BN+activation as the fused HIP kernels on HIP tensors
    pointwise = True  # 1x1 convolutions on the MFMA kernel of csrc/pointwise_conv.hip, BatchNorm statistics in its epilogue
    # 3x3 / stride 1 forward (+ BatchNorm statistics) and data gradient on hd_conv3x3_fwd instead of MIOpen: 20-28 us per
    # layer in the step against 25-38 us (+ a 6 us statistics launch) for the library's kernels.  The 3x3 WEIGHT gradient
    # (hd_conv3x3_wgrad) does not depend on this switch.
    conv3x3_forward = True
    stem = True  # the 6x6 / stride 2 stem on csrc/stem_conv.hip (forward + statistics, weight gradient)
    fork_shortcut = True  # a shortcut's gradient is added inside the bypassed 1x1 convolution's data-gradient kernel

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act="silu"):
        super().__init__()
        self.conv = ShadowConv2d(c1, c2, k, s, autopad(k, p), groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=1e-3, momentum=0.03)
        self.act = {"silu": nn.SiLU(inplace=True), "relu": nn.ReLU(inplace=True), None: nn.Identity()}[act]
        self.act_name = act

    def pointwise_mode(self, x) -> int:
        """How this layer's convolution runs on `x` in training (see pointwise_mode below)."""
        return pointwise_mode(self.conv, self.bn, x) if self.fused else 0

    def forward(self, x, residual=None, out=None, fork=False):
        """residual: added after the activation inside the fused kernel (the YOLOv5 bottleneck shortcut).
        out: a CatBuffer slot for the result (see FusedBNAct.apply).  fork: see conv_bn_act."""
        if self.fused:
            return conv_bn_act(self.conv, self.bn, self.act_name, x, residual, True, out, fork)
        y = self.act(self.bn(self.conv(x)))
        y = y if residual is None else y + residual
        return (y, x) if fork else y


def pointwise_mode(conv: nn.Conv2d, bn: nn.BatchNorm2d, x: torch.Tensor) -> int:
    """How a Conv2d(bias=False) -> BatchNorm2d pair runs on `x` in training: 2 = hd_pointwise_conv with the BatchNorm
    statistics in its epilogue (+ our backward kernels), 1 = library forward + our weight / data gradients (reductions
    too long for the resident-weight kernel), 0 = the library throughout (k x k, strided, grouped, biased, eval, fp32)."""
    if not (ConvBNAct.pointwise and conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1
            and conv.bias is None and conv.padding == (0, 0) and x.is_cuda and FusedBNAct.enabled and bn.training
            and bn.momentum is not None and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.bfloat16 and bn.num_features // 8 <= 256
            and conv.weight.requires_grad):
        return 0
    from . import ops
    w = conv.lp_weight() if isinstance(conv, ShadowConv2d) else conv.weight
    if w.dtype != torch.bfloat16 or x.dtype != torch.bfloat16:  # decide on what the kernels would see
        w = w.detach().to(torch.bfloat16)
        x = x.detach().to(torch.bfloat16)
    if ops.pointwise_conv_usable(x, w):
        return 2
    return 1 if ops.pointwise_backward_usable(x, w) else 0


def conv_bias_act(conv: nn.Conv2d, x: torch.Tensor, act=None) -> torch.Tensor:
    """act(conv(x)) for a BIASED convolution followed by ReLU or nothing (the RPN head's 3x3 -> ReLU): on HIP the library runs
    the convolution without its bias and one pass of ours adds the bias and applies the activation in place
    (ops.bias_act; backward: mask + bias-gradient sums in one pass instead of a mask pass and a strided reduction)."""
    if (conv.bias is not None and FusedBNAct.enabled and x.is_cuda and act in (None, "relu")
            and conv.out_channels % 8 == 0 and conv.out_channels // 4 <= 256):
        from . import ops
        w = conv.lp_weight() if isinstance(conv, ShadowConv2d) else conv.weight
        y = conv._conv_forward(x, w, None)
        if ops.bias_act_supported(y, act):
            return ops.bias_act(y, conv.bias, act, inplace=True)
        y = y + conv.bias.to(y.dtype)[None, :, None, None]
        return torch.relu(y) if act == "relu" else y
    y = conv(x)
    return torch.relu(y) if act == "relu" else y


def conv_bn_act(conv: nn.Conv2d, bn: nn.BatchNorm2d, act, x, residual=None, residual_after_act: bool = False, out=None,
                fork: bool = False):
    """Conv2d(bias=False) -> BatchNorm2d -> activation [-> / <- residual] on the fused HIP kernels: the 1x1 convolutions on
    csrc/pointwise_conv.hip where pointwise_mode says so, the BatchNorm + activation always (FusedBNAct.apply).
    fork: return (result, x') where x' stands for x in a shortcut around this layer: on the one-node 1x1 path x' is an alias
    whose gradient is added inside the layer's data-gradient kernel (ops._PwConvBNActFn), elsewhere it is x itself."""
    if fork:
        if pointwise_mode(conv, bn, x) and FusedBNAct.sync_group is None and x.dtype == torch.bfloat16 and x.requires_grad:
            from . import ops
            w = conv.lp_weight() if isinstance(conv, ShadowConv2d) else conv.weight
            w = w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)
            if ops.bn_act_supported(_like_conv_out(x, w)):
                return ops.pointwise_conv_bn_act(x, w, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, act,
                                                 FusedBNAct.slot_of(bn, x.device), residual, residual_after_act, out,
                                                 pointwise_mode(conv, bn, x) == 2, True)
        return conv_bn_act(conv, bn, act, x, residual, residual_after_act, out), x
    mode = pointwise_mode(conv, bn, x)
    if mode:
        from . import ops
        w = conv.lp_weight() if isinstance(conv, ShadowConv2d) else conv.weight
        w = w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)
        xb = x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16)
        if FusedBNAct.sync_group is None and ops.bn_act_supported(_like_conv_out(xb, w)):
            # one autograd node for convolution + BatchNorm + activation: its backward folds the BatchNorm's elementwise
            # half into the convolution's gradient kernels (ops._PwConvBNActFn); mode 2 also takes the statistics in the
            # forward kernel's epilogue
            return ops.pointwise_conv_bn_act(xb, w, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, act,
                                             FusedBNAct.slot_of(bn, x.device), residual, residual_after_act, out, mode == 2)
        if mode == 2:  # the convolution's epilogue leaves the batch statistics in the BatchNorm's slot: 3 fused-BN launches
            y = ops.pointwise_conv(xb, w, FusedBNAct.slot_of(bn, x.device))
            return FusedBNAct.apply(bn, y, act, residual, residual_after_act, out, stats_done=True)
        return FusedBNAct.apply(bn, ops.pointwise_conv(xb, w, None), act, residual, residual_after_act, out)
    if _conv3x3_ours(conv, bn, x):
        from . import ops
        w = conv.lp_weight() if isinstance(conv, ShadowConv2d) else conv.weight
        w = w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)  # what autocast would hand the library
        if ConvBNAct.conv3x3_forward and FusedBNAct.enabled and bn.momentum is not None and bn.num_features // 8 <= 256 \
                and ops.conv3x3_takes_stats(x, w, conv.stride[0]):
            # hand-written forward: the BatchNorm statistics come out of its epilogue
            y = ops.conv3x3(x, w, 1, FusedBNAct.slot_of(bn, x.device))
            return FusedBNAct.apply(bn, y, act, residual, residual_after_act, out, stats_done=True)
        return FusedBNAct.apply(bn, ops.conv3x3(x, w, conv.stride[0], None, ConvBNAct.conv3x3_forward), act, residual,
                                residual_after_act, out)
    if _conv3x3_s2_first(conv, bn, x):
        from . import ops
        w = conv.lp_weight() if isinstance(conv, ShadowConv2d) else conv.weight
        w = w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)
        w = w if w.is_contiguous(memory_format=torch.channels_last) else w.contiguous(memory_format=torch.channels_last)
        # weight gradient with the library; forward (+ the BatchNorm statistics, a 12 us pass of their own after the library's
        # 47 us) on hd_conv3x3s2_fwd, the data gradient (76 us + a 25 us zero-fill there) on hd_conv3x3s2_dgrad
        if ConvBNAct.conv3x3_forward and FusedBNAct.enabled and bn.momentum is not None and ops.conv3x3_takes_stats(x, w, 2):
            y = ops.conv3x3(x, w, 2, FusedBNAct.slot_of(bn, x.device))
            return FusedBNAct.apply(bn, y, act, residual, residual_after_act, out, stats_done=True)
        return FusedBNAct.apply(bn, ops.conv3x3(x, w, 2, None, False), act, residual, residual_after_act, out)
    if _stem_ours(conv, bn, x):
        from . import ops
        w = conv.lp_weight() if isinstance(conv, ShadowConv2d) else conv.weight
        w = w if w.dtype == torch.bfloat16 else w.to(torch.bfloat16)
        w = w if w.is_contiguous(memory_format=torch.channels_last) else w.contiguous(memory_format=torch.channels_last)
        if ops.stem_conv_supported(x, w):
            # the 6x6 / stride 2 stem on 3-channel images: float32 pixels read directly (no cast pass), statistics in the epilogue
            y = ops.stem_conv(x, w, FusedBNAct.slot_of(bn, x.device))
            return FusedBNAct.apply(bn, y, act, residual, residual_after_act, out, stats_done=True)
    return FusedBNAct.apply(bn, conv(x), act, residual, residual_after_act, out)


def _stem_ours(conv: nn.Conv2d, bn: nn.BatchNorm2d, x: torch.Tensor) -> bool:
    """The YOLOv5 v6.x stem (Conv2d(3, 32 | 64, 6, 2, 2), no bias) in bf16 training on HIP images that need no gradient."""
    return (ConvBNAct.stem and conv.kernel_size == (6, 6) and conv.stride == (2, 2) and conv.padding == (2, 2)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and conv.in_channels == 3
            and conv.out_channels in (32, 64) and x.is_cuda and not x.requires_grad and bn.training and bn.momentum is not None
            and FusedBNAct.enabled and torch.is_autocast_enabled("cuda") and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and conv.weight.requires_grad)


class _like_conv_out:
    """Shape / dtype / layout stand-in for the output of a 1x1 convolution of x (what ops.bn_act_supported looks at)."""

    def __init__(self, x, w):
        self.shape = (x.shape[0], w.shape[0], x.shape[2], x.shape[3])
        self.dtype, self.is_cuda = x.dtype, x.is_cuda

    def dim(self):
        return 4

    def is_contiguous(self, memory_format=None):
        return True


def _conv3x3_ours(conv: nn.Conv2d, bn: nn.BatchNorm2d, x: torch.Tensor) -> bool:
    """3x3 / pad 1 layers of stride 1 or 2 whose weight gradient hd_conv3x3_wgrad takes (bf16 rows, output W in {20, 40, 80},
    channel counts multiples of 64): forward and data gradient stay with the library."""
    if not (ConvBNAct.pointwise and conv.kernel_size == (3, 3) and conv.stride in ((1, 1), (2, 2)) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and x.is_cuda and bn.training
            and x.dtype == torch.bfloat16 and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.bfloat16 and conv.weight.requires_grad):
        return False
    from . import ops
    return ops.conv3x3_wgrad_supported(x, conv.out_channels, conv.stride[0])


def _conv3x3_s2_first(conv: nn.Conv2d, bn: nn.BatchNorm2d, x: torch.Tensor) -> bool:
    """The first stride-2 3x3 layer of the YOLOv5 backbone at 640^2 (32 -> 64 channels, 320^2 -> 160^2), whose weight gradient
    hd_conv3x3_wgrad does not take: its data gradient runs on hd_conv3x3s2_dgrad (ops._Conv3x3Fn.backward)."""
    if not (ConvBNAct.pointwise and conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and x.is_cuda and bn.training
            and x.dtype == torch.bfloat16 and x.dim() == 4 and x.requires_grad and torch.is_autocast_enabled("cuda")
            and torch.get_autocast_dtype("cuda") == torch.bfloat16 and conv.weight.requires_grad
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0):
        return False
    from . import _lib
    return bool(_lib.load().hd_conv3x3s2_dgrad_supported(x.shape[0], x.shape[2] // 2, x.shape[3] // 2, conv.in_channels,
                                                         conv.out_channels))


class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = ConvBNAct(c1, c_, 1, 1)
        self.cv2 = ConvBNAct(c_, c2, 3, 1)
        self.add = shortcut and c1 == c2

    def forward(self, x, out=None):
        # the shortcut add rides in the second conv's fused BN+activation kernel, and (backward) the shortcut's gradient in
        # the first convolution's data-gradient kernel
        if self.add and ConvBNAct.fork_shortcut:
            h, xs = self.cv1(x, None, None, True)
            return self.cv2(h, xs, out)
        return self.cv2(self.cv1(x), x if self.add else None, out)


class C3(nn.Module):
    """CSP bottleneck with 3 convolutions (YOLOv5 v6.x).

    The two branch convolutions cv1 and cv2 (both 1x1, same input, same width c_, each followed by its own
    BatchNorm + SiLU) are held as ONE convolution to 2*c_ channels with ONE BatchNorm (`cv12`): BatchNorm statistics
    and affine parameters are per channel, so this is the same function of the same number of parameters with the
    same initialisation — but one MIOpen convolution forward, one data-gradient and one weight-gradient kernel
    instead of two each, four fused-BN launches instead of eight, and no gradient-sum kernel for the shared input.
    Channels [0, c_) are the cv1 branch (into the bottleneck stack), [c_, 2c_) the cv2 branch, which the fused BN
    kernel writes straight into its half of the block's concatenation (ops.CatBuffer; hd_rows2 second place).
    `split_state_dict` / `merge_state_dict` translate to and from the cv1 / cv2 naming of the public v6.x weights."""

    def __init__(self, c1, c2, n=1, shortcut=True, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.c_ = c_
        self.cv12 = ConvBNAct(c1, 2 * c_, 1, 1)
        self.cv3 = ConvBNAct(2 * c_, c2, 1, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, e=1.0) for _ in range(n)))

    def forward(self, x):
        c_ = self.c_
        cb = cat_buffer(x, (c_, c_), x.shape[2:])
        if cb is None:
            y = self.cv12(x)
            return self.cv3(torch.cat((self.m(y[:, :c_]), y[:, c_:]), 1))
        # first half: a dense tensor for the bottleneck stack; second half: straight into the concatenation
        y1 = torch.empty((x.shape[0], c_, x.shape[2], x.shape[3]), dtype=cb.buf.dtype, device=x.device,
                         memory_format=torch.channels_last)
        y = self.cv12(x, None, (y1, cb.slot(1)))
        if isinstance(y, tuple):
            y, y2 = y
        else:  # the unfused fallback returned one tensor
            y, y2 = y[:, :c_], y[:, c_:]
        for k, blk in enumerate(self.m):
            y = blk(y, cb.slot(0) if k == len(self.m) - 1 else None)
        return self.cv3(cb.join((y, y2)))

    @staticmethod
    def split_state_dict(sd: dict, prefix: str) -> dict:
        """`prefix`cv12.* -> `prefix`cv1.* + `prefix`cv2.* (the two halves along the output-channel axis)."""
        out = dict(sd)
        for k in [k for k in sd if k.startswith(prefix + "cv12.")]:
            v = out.pop(k)
            rest = k[len(prefix) + 5:]
            if v.dim() == 0:  # num_batches_tracked
                out[prefix + "cv1." + rest] = v.clone()
                out[prefix + "cv2." + rest] = v.clone()
            else:
                h = v.shape[0] // 2
                out[prefix + "cv1." + rest] = v[:h].clone()
                out[prefix + "cv2." + rest] = v[h:].clone()
        return out

    @staticmethod
    def merge_state_dict(sd: dict, prefix: str) -> dict:
        """The inverse: `prefix`cv1.* + `prefix`cv2.* -> `prefix`cv12.*."""
        out = dict(sd)
        for k in [k for k in sd if k.startswith(prefix + "cv1.")]:
            rest = k[len(prefix) + 4:]
            a, b = out.pop(k), out.pop(prefix + "cv2." + rest)
            out[prefix + "cv12." + rest] = a.clone() if a.dim() == 0 else torch.cat((a, b), 0)
        return out


class C2f(nn.Module):
    """YOLOv8-style CSP block (used by the 'YOLOv8-style' PAFPN of the FasterRCNN family, README.md:8)."""

    def __init__(self, c1, c2, n=1, shortcut=False, e=0.5, act="silu"):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = ConvBNAct(c1, 2 * self.c, 1, 1, act=act)
        self.cv2 = ConvBNAct((2 + n) * self.c, c2, 1, 1, act=act)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, e=1.0) for _ in range(n))

    def forward(self, x):
        c, n = self.c, len(self.m)
        cb = cat_buffer(x, (c,) * (2 + n), x.shape[2:])
        if cb is None:
            y = list(self.cv1(x).chunk(2, 1))
            y.extend(m(y[-1]) for m in self.m)
            return self.cv2(torch.cat(y, 1))
        # every part is written straight into its slice of the concatenation: cv1's fused BN+act puts its two halves into
        # slots 0 and 1 (one launch, an output in two places), bottleneck k's last kernel writes slot 2 + k, and join() copies
        # nothing — a torch.cat of a 384-channel 208 x 208 map is a 1 GB read + write at 832^2 bs16 (FasterRCNN PAFPN)
        y = self.cv1(x, None, (cb.slot(0), cb.slot(1)))
        parts = list(y) if isinstance(y, tuple) else list(y.chunk(2, 1))
        for k, m in enumerate(self.m):
            parts.append(m(parts[-1], cb.slot(2 + k)))
        return self.cv2(cb.join(parts))


class SPPF(nn.Module):
    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = ConvBNAct(c1, c_, 1, 1)
        self.cv2 = ConvBNAct(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def forward(self, x):
        c_ = self.cv1.conv.out_channels
        cb = cat_buffer(x, (c_, c_, c_, c_), x.shape[2:]) if self.m.kernel_size == 5 else None
        if cb is not None:
            from . import ops  # HIP SPPF pool (arg-max tap forward, gather backward); see csrc/pool.hip
            # cv1's activation and the three pools are written straight into their quarter of the concatenation
            x = self.cv1(x, None, cb.slot(0))
            if ops.maxpool5_supported(x):
                y1 = ops.maxpool5(x, cb.slot(1))
                y2 = ops.maxpool5(y1, cb.slot(2))
                return self.cv2(cb.join((x, y1, y2, ops.maxpool5(y2, cb.slot(3)))))
            y1 = self.m(x)
            y2 = self.m(y1)
            return self.cv2(cb.join((x, y1, y2, self.m(y2))))
        x = self.cv1(x)
        pool = self.m
        if self.m.kernel_size == 5 and x.is_cuda:
            from . import ops
            if ops.maxpool5_supported(x):
                pool = ops.maxpool5
        y1 = pool(x)
        y2 = pool(y1)
        return self.cv2(torch.cat((x, y1, y2, pool(y2)), 1))


class DropBlock2d(nn.Module):
    """DropBlock regularisation (the "*_dropBlock0.5" rows of README.md:90,132-133): contiguous
    `block_size` x `block_size` squares of a feature map are zeroed with overall drop probability `p`,
    and the survivors rescaled.  Identity in eval mode.  Device-side only (no host sync); under train.Trainer the uniform
    field comes from the step's StepRandom pool (rng.py), so a step that contains it captures into a hipGraph and still sees
    fresh values at every replay, split or not."""

    def __init__(self, p: float = 0.1, block_size: int = 5):
        super().__init__()
        if not 0.0 <= p < 1.0 or block_size < 1 or block_size % 2 == 0:
            raise ValueError("DropBlock2d: need 0 <= p < 1 and an odd block_size")
        self.p, self.block_size = float(p), int(block_size)

    def forward(self, x):
        if not self.training or self.p == 0.0:
            return x
        B, C, H, W = x.shape
        bs = min(self.block_size, H if H % 2 else H - 1, W if W % 2 else W - 1)
        bs = max(bs, 1)
        # seed rate so that the expected dropped fraction is p (the public DropBlock formula)
        gamma = self.p / (bs * bs) * (H * W) / max((H - bs + 1) * (W - bs + 1), 1)
        from .rng import StepRandom
        pool = StepRandom.active()   # train.Trainer: the field is refilled eagerly before every step (rng.py)
        u = pool.uniform(self, (B, C, H, W)) if pool is not None else torch.rand((B, C, H, W), device=x.device)
        seeds = (u < gamma).to(x.dtype)
        drop = torch.nn.functional.max_pool2d(seeds, bs, 1, bs // 2)
        keep = 1.0 - drop
        scale = keep.numel() / keep.float().sum().clamp(min=1.0)
        return x * keep * scale.to(x.dtype)


def make_drop_block(cfg):
    """None | float p | dict(p=, block_size=) -> module (Identity when disabled)."""
    if not cfg:
        return nn.Identity()
    if isinstance(cfg, (int, float)):
        return DropBlock2d(float(cfg))
    return DropBlock2d(**cfg)
