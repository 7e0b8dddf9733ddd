"""Backbones (reference README.md:8-9,65,120): CSPDarknet for YOLOv5, ResNet50 for FasterRCNN.
Plain torch.nn — the conv FLOPs go to MIOpen/MFMA; nothing here is hand-written HIP."""
from __future__ import annotations

import math
from typing import List

import torch
import torch.nn as nn

from . import ops
from .layers import C3, SPPF, ConvBNAct, FusedBNAct, ShadowConv2d, conv_bn_act
from .registry import BACKBONES


def _w(c, width):  # channel scaling, multiples of 8
    return int(math.ceil(c * width / 8) * 8)


def _d(n, depth):
    return max(round(n * depth), 1)


@BACKBONES.register()
class CSPDarknet(nn.Module):
    """YOLOv5 v6.1 backbone; returns (P3/8, P4/16, P5/32). s: depth .33 width .5; l: 1.0 / 1.0."""

    PRESETS = {"n": (0.33, 0.25), "s": (0.33, 0.50), "m": (0.67, 0.75), "l": (1.0, 1.0), "x": (1.33, 1.25)}

    def __init__(self, variant: str = "s", depth: float | None = None, width: float | None = None, in_ch: int = 3,
                 frozen: bool = False):
        super().__init__()
        self.frozen = bool(frozen)  # "..._frozeBackbone" (README.md:133): no gradients, BatchNorm on running stats
        d0, w0 = self.PRESETS[variant]
        depth, width = depth or d0, width or w0
        c = [_w(v, width) for v in (64, 128, 256, 512, 1024)]
        self.out_channels = (c[2], c[3], c[4])
        self.stem = ConvBNAct(in_ch, c[0], 6, 2, 2)
        self.stage1 = nn.Sequential(ConvBNAct(c[0], c[1], 3, 2), C3(c[1], c[1], _d(3, depth)))
        self.stage2 = nn.Sequential(ConvBNAct(c[1], c[2], 3, 2), C3(c[2], c[2], _d(6, depth)))
        self.stage3 = nn.Sequential(ConvBNAct(c[2], c[3], 3, 2), C3(c[3], c[3], _d(9, depth)))
        self.stage4 = nn.Sequential(ConvBNAct(c[3], c[4], 3, 2), C3(c[4], c[4], _d(3, depth)), SPPF(c[4], c[4], 5))
        self.depth, self.width = depth, width
        if self.frozen:
            for p in self.parameters():
                p.requires_grad_(False)
            self.train(False)

    def train(self, mode: bool = True):
        super().train(mode and not self.frozen)
        return self

    def forward(self, x) -> List[torch.Tensor]:
        x = self.stage1(self.stem(x))
        p3 = self.stage2(x)
        p4 = self.stage3(p3)
        p5 = self.stage4(p4)
        return [p3, p4, p5]


class _ResBottleneck(nn.Module):
    expansion = 4

    def __init__(self, c_in, width, stride=1, downsample=None):
        super().__init__()
        self.conv1 = ShadowConv2d(c_in, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = ShadowConv2d(width, width, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = ShadowConv2d(width, width * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(width * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        # the 1x1 convolutions (conv1, conv3, a stride-1 downsample) take the MFMA kernels of csrc/pointwise_conv.hip; the
        # identity shortcut's gradient is added inside conv1's data-gradient kernel (`fork`)
        if self.downsample is None:
            if ConvBNAct.fork_shortcut:
                y, idt = conv_bn_act(self.conv1, self.bn1, "relu", x, fork=True)
            else:
                y, idt = conv_bn_act(self.conv1, self.bn1, "relu", x), x
        else:
            # the block input feeds conv1 and the projection shortcut: conv1 hands on an alias whose gradient (the projection's
            # data gradient) is added inside conv1's own data-gradient kernel — no separate add of two 100+ MB maps
            if ConvBNAct.fork_shortcut:
                y, x1 = conv_bn_act(self.conv1, self.bn1, "relu", x, fork=True)
            else:
                y, x1 = conv_bn_act(self.conv1, self.bn1, "relu", x), x
            idt = conv_bn_act(self.downsample[0], self.downsample[1], None, x1)
        y = conv_bn_act(self.conv2, self.bn2, "relu", y)
        # relu(bn3(conv3) + identity) in one fused kernel (residual added before the activation)
        return conv_bn_act(self.conv3, self.bn3, "relu", y, idt, False)


@BACKBONES.register()
class ResNet50(nn.Module):
    """ResNet-50 (stride on the 3x3, v1.5); returns (C2/4, C3/8, C4/16, C5/32)."""

    def __init__(self, in_ch: int = 3, frozen_stages: int = 0):
        super().__init__()
        self.conv1 = ShadowConv2d(in_ch, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self._c = 64
        self.layer1 = self._make(64, 3, 1)
        self.layer2 = self._make(128, 4, 2)
        self.layer3 = self._make(256, 6, 2)
        self.layer4 = self._make(512, 3, 2)
        self.out_channels = (256, 512, 1024, 2048)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self.frozen_stages = frozen_stages
        if frozen_stages > 0:  # "frozen backbone" experiments, README.md:132
            for mod in [self.conv1, self.bn1, self.layer1, self.layer2, self.layer3, self.layer4][: frozen_stages + 1]:
                for p in mod.parameters():
                    p.requires_grad_(False)

    def _make(self, width, n, stride):
        ds = None
        if stride != 1 or self._c != width * 4:
            ds = nn.Sequential(ShadowConv2d(self._c, width * 4, 1, stride, bias=False), nn.BatchNorm2d(width * 4))
        layers = [_ResBottleneck(self._c, width, stride, ds)]
        self._c = width * 4
        layers += [_ResBottleneck(self._c, width) for _ in range(n - 1)]
        return nn.Sequential(*layers)

    def forward(self, x, tap=None):
        """tap(i, feature) -> feature: called on every stage output BEFORE the next stage reads it; a neck uses it to run its
        lateral 1x1 convolution as a `fork` (layers.conv_bn_act) and hand back the alias, so that the stage output's gradient is
        accumulated inside the data-gradient kernels of its consumers instead of by add launches (detectors.FasterRCNN)."""
        x = FusedBNAct.apply(self.bn1, self.conv1(x), "relu")
        if FusedBNAct.enabled and ops.maxpool3s2_supported(x):
            x = ops.maxpool3s2(x)  # one byte of tap index per output, gather backward (csrc/pool.hip)
        else:
            x = self.maxpool(x)
        feats = []
        for i, layer in enumerate((self.layer1, self.layer2, self.layer3, self.layer4)):
            x = layer(x)
            if tap is not None:
                x = tap(i, x)
            feats.append(x)
        return feats
