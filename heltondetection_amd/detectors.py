"""Detectors = backbone + neck + head(s), assembled from registry configs (reference
README.md:8-9,12).  forward_train returns (loss, parts); forward_test returns padded detections."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import backbones, heads, layers, necks, assigners  # noqa: F401  (registers the classes)
from .registry import BACKBONES, DETECTORS, HEADS, NECKS


@DETECTORS.register()
class YOLOv5(nn.Module):
    def __init__(self, backbone: Optional[Dict] = None, neck: Optional[Dict] = None, head: Optional[Dict] = None,
                 variant: str = "s", num_classes: int = 80, img_size: int = 640):
        super().__init__()
        self.backbone = BACKBONES.build(backbone or dict(type="CSPDarknet", variant=variant))
        self.neck = NECKS.build(neck or dict(type="YOLOv5PAFPN", depth=self.backbone.depth),
                                in_channels=self.backbone.out_channels)
        head = dict(head) if head else dict(type="YOLOv5Head")
        head.setdefault("num_classes", num_classes)  # a head config without its own class count takes the detector's
        head.setdefault("img_size", img_size)
        self.head = HEADS.build(head, in_channels=self.neck.out_channels)
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                # num_batches_tracked stays a buffer (checkpoints of stock torch BatchNorm2d modules load with
                # strict=True, and ours load there); the fused HIP path never touches it — momentum is fixed, so
                # the counter is not read, and there is no per-layer "add 1" launch
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


@DETECTORS.register()
class FasterRCNN(nn.Module):
    """Two-stage detector (README.md:8,65): ResNet50 + FPN | PAFPN (256 ch) [+ P2-only variants] + RPN +
    RoIAlign head (standard | Decoupled)."""

    def __init__(self, backbone: Optional[Dict] = None, neck: Optional[Dict] = None, rpn_head: Optional[Dict] = None,
                 roi_head: Optional[Dict] = None, num_classes: int = 80):
        super().__init__()
        self.backbone = BACKBONES.build(backbone or dict(type="ResNet50"))
        self.neck = NECKS.build(neck or dict(type="PAFPN", out_channels=256), in_channels=self.backbone.out_channels)
        ch = self.neck.out_channels
        p2_only = getattr(self.neck, "p2_only", False)
        if rpn_head is None:
            rpn_head = dict(type="RPNHead", in_channels=ch)
            if p2_only:  # the single stride-4 level carries all five anchor scales
                rpn_head.update(strides=(4,), sizes=[[32.0, 64.0, 128.0, 256.0, 512.0]])
        self.rpn_head = HEADS.build(rpn_head)
        if roi_head is None:
            roi_head = dict(type="RoIHead", in_channels=ch, num_classes=num_classes, decoupled=True)
            if p2_only:
                roi_head.update(featmap_strides=(4,))
        roi_head = dict(roi_head)
        roi_head.setdefault("num_classes", num_classes)
        self.roi_head = HEADS.build(roi_head)

    def extract(self, images):
        if (self.training and layers.ConvBNAct.fork_shortcut and hasattr(self.neck, "lateral_fork")
                and isinstance(self.backbone, backbones.ResNet50)):
            # the neck's laterals run as taps on the stage outputs: each output's gradient then flows through one chain
            # (lateral -> next stage's conv1 -> projection shortcut), every link adding the next one's gradient in its
            # data-gradient epilogue
            lats = [None] * len(self.backbone.out_channels)

            def tap(i, f):
                lats[i], alias = self.neck.lateral_fork(i, f)
                return alias

            feats = self.backbone(images, tap)
            return self.neck(feats, lats)
        return self.neck(self.backbone(images))

    def forward_train(self, images, gt_boxes, gt_cls, m_valid):
        feats = self.extract(images)
        img_hw = images.shape[-2:]
        logits, deltas = self.rpn_head(feats)
        anchors, level_sizes = self.rpn_head.anchors_for(feats)
        l_rpn_cls, l_rpn_reg = self.rpn_head.loss(logits, deltas, anchors, gt_boxes, m_valid)
        rois, _, count = self.rpn_head.proposals(logits, deltas, anchors, level_sizes, img_hw, True)
        rois5, cls_t, reg_t, ok, is_fg = self.roi_head.sample(rois, count, gt_boxes, gt_cls, m_valid)
        cls_logits, box_deltas = self.roi_head(feats, rois5)
        l_cls, l_reg = self.roi_head.loss(cls_logits, box_deltas, cls_t, reg_t, ok, is_fg)
        total = l_rpn_cls + l_rpn_reg + l_cls + l_reg
        return total, torch.stack([total.detach(), l_rpn_cls.detach(), l_rpn_reg.detach(), l_cls.detach(), l_reg.detach()])

    @torch.no_grad()
    def forward_test(self, images):
        feats = self.extract(images)
        img_hw = images.shape[-2:]
        logits, deltas = self.rpn_head(feats)
        anchors, level_sizes = self.rpn_head.anchors_for(feats)
        rois, _, count = self.rpn_head.proposals(logits, deltas, anchors, level_sizes, img_hw, False)
        return self.roi_head.predict(feats, rois, count, img_hw)

    def forward(self, images, gt_boxes=None, gt_cls=None, m_valid=None):
        if gt_boxes is not None:
            return self.forward_train(images, gt_boxes, gt_cls, m_valid)
        return self.forward_test(images)
