"""Necks (reference README.md:8-9,65): YOLOv5 PAFPN; FPN and a YOLOv8-style PAFPN (256 output
channels) for FasterRCNN, each with a P2-only variant flag.  Plain torch.nn (MIOpen convs)."""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from .layers import C2f, C3, ConvBNAct, ShadowConv2d, cat_buffer, make_drop_block
from .registry import NECKS


def up_cat(small: torch.Tensor, big: torch.Tensor) -> torch.Tensor:
    """cat(upsample2x(small), big) along channels: one HIP gather into the bf16 concatenation (ops.upsample2x_cat) instead
    of a float32 nearest upsample, a promoting torch.cat and the casts back; other shapes / devices take the torch ops."""
    if ConvBNAct.fused and small.is_cuda:
        from . import ops
        if ops.upsample2x_cat_supported(small, big):
            return ops.upsample2x_cat(small, big)
    return torch.cat((F.interpolate(small, size=big.shape[-2:], mode="nearest"), big), 1)


@NECKS.register()
class YOLOv5PAFPN(nn.Module):
    """v6.1 head layers 10-23: top-down + bottom-up over (P3, P4, P5)."""

    def __init__(self, in_channels: Sequence[int], depth: float = 0.33, drop_block=None):
        super().__init__()
        self.drop = make_drop_block(drop_block)  # "YOLOv5s_dropBlock0.5" (README.md:132): on the neck outputs
        c3, c4, c5 = in_channels
        n = max(round(3 * depth), 1)
        self.lat5 = ConvBNAct(c5, c4, 1, 1)
        self.td4 = C3(2 * c4, c4, n, shortcut=False)
        self.lat4 = ConvBNAct(c4, c3, 1, 1)
        self.td3 = C3(2 * c3, c3, n, shortcut=False)
        self.down3 = ConvBNAct(c3, c3, 3, 2)
        self.bu4 = C3(2 * c3, c4, n, shortcut=False)
        self.down4 = ConvBNAct(c4, c4, 3, 2)
        self.bu5 = C3(2 * c4, c5, n, shortcut=False)
        self.out_channels = (c3, c4, c5)

    @staticmethod
    def _down_cat(down: ConvBNAct, x: torch.Tensor, lateral: torch.Tensor, cb=None) -> torch.Tensor:
        """cat(down(x), lateral): the strided conv's fused BN+act writes its half of the concatenation in place; the
        lateral (which has other consumers) is copied in, as torch.cat would — unless the caller made the buffer `cb` before
        the lateral was computed and had it written into slot 1 then (YOLOv5PAFPN.forward): join() finds it in place."""
        if cb is None:
            cb = cat_buffer(x, (down.conv.out_channels, lateral.shape[1]), lateral.shape[2:])
        if cb is None:
            return torch.cat((down(x), lateral), 1)
        return cb.join((down(x, None, cb.slot(0)), lateral))

    @staticmethod
    def _lateral_into_cat(lat: ConvBNAct, down: ConvBNAct, x: torch.Tensor):
        """lat(x) written straight into slot 1 of the bottom-up concatenation cat(down(.), lat(x)) it will join later — the
        top-down path reads it there as a channel slice (hd_upsample2x_cat takes row strides), so the copy torch.cat would make
        (6.9 + 5.4 us per YOLOv5s step) never runs.  -> (lateral, buffer or None)."""
        cb = cat_buffer(x, (down.conv.out_channels, lat.conv.out_channels), x.shape[2:])
        if cb is None:
            return lat(x), None
        return lat(x, None, cb.slot(1)), cb

    _up_cat = staticmethod(lambda small, big: up_cat(small, big))

    def forward(self, feats: List[torch.Tensor]) -> List[torch.Tensor]:
        p3, p4, p5 = feats
        l5, cb5 = self._lateral_into_cat(self.lat5, self.down4, p5)
        t4 = self.td4(self._up_cat(l5, p4))
        l4, cb4 = self._lateral_into_cat(self.lat4, self.down3, t4)
        o3 = self.td3(self._up_cat(l4, p3))
        o4 = self.bu4(self._down_cat(self.down3, o3, l4, cb4))
        o5 = self.bu5(self._down_cat(self.down4, o4, l5, cb5))
        return [self.drop(o3), self.drop(o4), self.drop(o5)]


@NECKS.register()
class FPN(nn.Module):
    """Feature Pyramid Network over (C2..C5) -> (P2..P5) [+P6 by stride-2 subsampling for the RPN]."""

    def __init__(self, in_channels: Sequence[int], out_channels: int = 256, extra_level: bool = True,
                 p2_only: bool = False, drop_block=None):
        super().__init__()
        self.drop = make_drop_block(drop_block)
        self.lateral = nn.ModuleList(ShadowConv2d(c, out_channels, 1) for c in in_channels)
        self.output = nn.ModuleList(ShadowConv2d(out_channels, out_channels, 3, padding=1) for _ in in_channels)
        self.extra_level = extra_level
        self.p2_only = p2_only  # "P2" variants: only level P2 reaches the heads (README.md:65)
        self.out_channels = out_channels

    def forward(self, feats):
        lat = [l(f) for l, f in zip(self.lateral, feats)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[-2:], mode="nearest")
        outs = [self.drop(o(x)) for o, x in zip(self.output, lat)]
        if self.p2_only:
            return outs[:1]
        if self.extra_level:
            outs.append(F.max_pool2d(outs[-1], 1, 2, 0))
        return outs


@NECKS.register()
class PAFPN(nn.Module):
    """YOLOv8-style PAFPN with a fixed output width (256; README.md:8): 1x1 laterals to `out_channels`,
    C2f top-down fusion, strided-conv + C2f bottom-up fusion."""

    def __init__(self, in_channels: Sequence[int], out_channels: int = 256, n: int = 1, extra_level: bool = True,
                 p2_only: bool = False, act: str = "silu", drop_block=None):
        super().__init__()
        self.drop = make_drop_block(drop_block)  # "..._dropBlock0.5" (README.md:90)
        L = len(in_channels)
        self.lateral = nn.ModuleList(ConvBNAct(c, out_channels, 1, 1, act=act) for c in in_channels)
        self.td = nn.ModuleList(C2f(2 * out_channels, out_channels, n, act=act) for _ in range(L - 1))
        self.down = nn.ModuleList(ConvBNAct(out_channels, out_channels, 3, 2, act=act) for _ in range(L - 1))
        self.bu = nn.ModuleList(C2f(2 * out_channels, out_channels, n, act=act) for _ in range(L - 1))
        self.extra_level = extra_level
        self.p2_only = p2_only
        self.out_channels = out_channels

    def lateral_fork(self, i: int, feat):
        """Lateral i on a backbone stage output, as a fork: -> (lateral, alias of feat for the backbone's next stage)."""
        return self.lateral[i](feat, fork=True)

    def forward(self, feats, laterals=None):
        """laterals: the lateral maps when they were already taken through lateral_fork (backbone taps)."""
        lat = list(laterals) if laterals is not None else [l(f) for l, f in zip(self.lateral, feats)]
        L = len(lat)
        td = [None] * L
        td[L - 1] = lat[L - 1]
        for i in range(L - 2, -1, -1):
            td[i] = self.td[i](up_cat(td[i + 1], lat[i]))   # upsample + concatenation as one gather
        outs = [td[0]]
        for i in range(1, L):
            # the strided convolution's fused BN+act writes its half of the concatenation in place
            outs.append(self.bu[i - 1](YOLOv5PAFPN._down_cat(self.down[i - 1], outs[-1], td[i])))
        outs = [self.drop(o) for o in outs]
        if self.p2_only:
            return outs[:1]
        if self.extra_level:
            outs.append(F.max_pool2d(outs[-1], 1, 2, 0))
        return outs
