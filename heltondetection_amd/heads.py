"""Detection heads (reference README.md:8-9).  The heads own the calls into the HIP hot path:
YOLOv5Head -> decode / build-targets / fused loss / per-class NMS;  RPNHead -> anchor grid,
IoU assignment, proposals (top-k + NMS);  RoIHead -> RoIAlign over pyramid levels."""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .layers import ConvBNAct
from .registry import ASSIGNERS, HEADS

YOLOV5_ANCHORS = (((10, 13), (16, 30), (33, 23)), ((30, 61), (62, 45), (59, 119)), ((116, 90), (156, 198), (373, 326)))


@HEADS.register()
class YOLOv5Head(nn.Module):
    """Detect layer (1x1 convs to na*(5+nc) channels per level) + loss + post-processing.

    The conv output stays in channels-last memory; the HIP kernels read it through the
    [B, na, H, W, no] strided view, so there is no permute/contiguous copy in either direction.
    """

    def __init__(self, in_channels: Sequence[int], num_classes: int = 80, anchors=YOLOV5_ANCHORS,
                 strides: Sequence[int] = (8, 16, 32), assigner=None, box_gain: float = 0.05, obj_gain: float = 1.0,
                 cls_gain: float = 0.5, balance: Sequence[float] = (4.0, 1.0, 0.4), img_size: int = 640):
        super().__init__()
        self.nc, self.no = int(num_classes), int(num_classes) + 5
        self.anchors = tuple(tuple(tuple(float(v) for v in a) for a in lv) for lv in anchors)
        self.na, self.nl = len(self.anchors[0]), len(self.anchors)
        self.strides = tuple(float(s) for s in strides)
        self.m = nn.ModuleList(nn.Conv2d(c, self.na * self.no, 1) for c in in_channels)
        self.assigner = ASSIGNERS.build(assigner or dict(type="YOLOv5Assigner", anchor_t=4.0))
        # gain scaling as in the public v6.1 training script
        self.box_gain = box_gain * 3.0 / self.nl
        self.cls_gain = cls_gain * self.nc / 80.0 * 3.0 / self.nl
        self.obj_gain = obj_gain * (img_size / 640.0) ** 2 * 3.0 / self.nl
        self.balance = tuple(balance)
        self.bias_grad_in_loss = True
        self._init_bias()

    def _init_bias(self):
        for conv, s in zip(self.m, self.strides):
            b = conv.bias.view(self.na, -1)
            with torch.no_grad():
                b[:, 4] += math.log(8 / (640 / s) ** 2)
                b[:, 5:5 + self.nc] += math.log(0.6 / (self.nc - 0.99999))

    def forward(self, feats: List[torch.Tensor]) -> List[torch.Tensor]:
        views = []
        for conv, f in zip(self.m, feats):
            # bias.detach(): the bias gradient comes from the loss kernels (ops.yolo_loss(biases=...)),
            # not from a framework reduction over the dense [B, na*no, H, W] gradient
            bias = conv.bias.detach() if (self.training and self.bias_grad_in_loss) else conv.bias
            y = F.conv2d(f, conv.weight, bias).float()
            if not y.is_contiguous(memory_format=torch.channels_last):
                y = y.contiguous(memory_format=torch.channels_last)
            B, _, H, W = y.shape
            views.append(y.view(B, self.na, self.no, H, W).permute(0, 1, 3, 4, 2))
        return views

    def loss(self, preds, gt_boxes, gt_cls, m_valid):
        shapes = [(self.na, p.shape[2], p.shape[3], self.no) for p in preds]
        slots = self.assigner(shapes, self.strides, self.anchors, gt_boxes, gt_cls, m_valid)
        biases = [c.bias for c in self.m] if (self.training and self.bias_grad_in_loss) else None
        return ops.yolo_loss(preds, self.strides, self.anchors, slots, gt_cls.shape[1], self.box_gain, self.obj_gain,
                             self.cls_gain, self.balance, biases=biases)

    @torch.no_grad()
    def predict(self, preds, conf_thr=0.25, iou_thr=0.45, max_nms=30000, max_det=300):
        dec = ops.yolo_decode(preds, self.strides, self.anchors)
        return ops.yolo_nms(dec, conf_thr, iou_thr, max_nms, max_det)
