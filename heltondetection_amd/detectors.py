"""Detectors = backbone + neck + head(s), assembled from registry configs (reference
README.md:8-9,12).  forward_train returns (loss, parts); forward_test returns padded detections."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import backbones, heads, necks, assigners  # noqa: F401  (registers the classes)
from .registry import BACKBONES, DETECTORS, HEADS, NECKS


@DETECTORS.register()
class YOLOv5(nn.Module):
    def __init__(self, backbone: Optional[Dict] = None, neck: Optional[Dict] = None, head: Optional[Dict] = None,
                 variant: str = "s", num_classes: int = 80, img_size: int = 640):
        super().__init__()
        self.backbone = BACKBONES.build(backbone or dict(type="CSPDarknet", variant=variant))
        self.neck = NECKS.build(neck or dict(type="YOLOv5PAFPN", depth=self.backbone.depth),
                                in_channels=self.backbone.out_channels)
        self.head = HEADS.build(head or dict(type="YOLOv5Head", num_classes=num_classes, img_size=img_size),
                                in_channels=self.neck.out_channels)
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eps, m.momentum = 1e-3, 0.03

    def extract(self, images):
        return self.neck(self.backbone(images))

    def forward_train(self, images, gt_boxes, gt_cls, m_valid):
        preds = self.head(self.extract(images))
        return self.head.loss(preds, gt_boxes, gt_cls, m_valid)

    @torch.no_grad()
    def forward_test(self, images, conf_thr=0.25, iou_thr=0.45, max_nms=30000, max_det=300):
        preds = self.head(self.extract(images))
        return self.head.predict(preds, conf_thr, iou_thr, max_nms, max_det)

    def forward(self, images, gt_boxes=None, gt_cls=None, m_valid=None):
        if gt_boxes is not None:
            return self.forward_train(images, gt_boxes, gt_cls, m_valid)
        return self.forward_test(images)


def build_detector(cfg: Dict) -> nn.Module:
    return DETECTORS.build(cfg)
