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
from .layers import ConvBNAct, ShadowConv2d, conv_bias_act
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
                 cls_gain: float = 0.5, balance: Sequence[float] = (4.0, 1.0, 0.4), img_size: int = 640,
                 focal: Optional[dict] = None):
        super().__init__()
        # focal: dict(gamma=1.5, alpha=0.25, terms=("cls", "obj")) — the "YOLOv5s-focalloss(...)" rows of
        # README.md:118-154; which BCE terms it wraps is the `terms` subset (e.g. ("cls",) only)
        self.focal = dict(focal) if focal else None
        self.nc, self.no = int(num_classes), int(num_classes) + 5
        self.anchors = tuple(tuple(tuple(float(v) for v in a) for a in lv) for lv in anchors)
        self.na, self.nl = len(self.anchors[0]), len(self.anchors)
        self.strides = tuple(float(s) for s in strides)
        self.m = nn.ModuleList(ShadowConv2d(c, self.na * self.no, 1) for c in in_channels)
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
        self._roots = self._head_in = None
        if (self.training and self.bias_grad_in_loss
                and all(ops.head_conv_supported(f, conv.lp_weight(), conv.bias) for conv, f in zip(self.m, feats))):
            # training on the HIP Detect convolution (ops.head_conv): float32 rows padded to a 16-byte pitch (255 -> 256 channels),
            # read by the loss through [B, na, H, W, no] views; inside ops.yolo_loss autograd differentiates the padded rows
            # themselves (`_roots`, handed over by loss() below), so no cast, bias add, permute or slice runs in either
            # direction (a caller that differentiates the views directly still gets gradients, through as_strided's backward)
            roots, self._head_in = [], []
            for conv, f in zip(self.m, feats):
                y = ops.head_conv(f, conv.lp_weight(), conv.bias)
                B, P, H, W = y.shape
                roots.append(y)
                self._head_in.append((f, conv.lp_weight()))
                views.append(y.as_strided((B, self.na, H, W, self.no), (H * W * P, self.no, W * P, P, 1), y.storage_offset()))
            self._roots = roots
            return views
        for conv, f in zip(self.m, feats):
            # bias.detach(): the bias gradient comes from the loss kernels (ops.yolo_loss(biases=...)),
            # not from a framework reduction over the dense [B, na*no, H, W] gradient
            bias = conv.bias.detach() if (self.training and self.bias_grad_in_loss) else conv.bias
            y = F.conv2d(f, conv.lp_weight(), bias).float()
            if not y.is_contiguous(memory_format=torch.channels_last):
                y = y.contiguous(memory_format=torch.channels_last)
            B, _, H, W = y.shape
            views.append(y.view(B, self.na, self.no, H, W).permute(0, 1, 3, 4, 2))
        return views

    def loss(self, preds, gt_boxes, gt_cls, m_valid):
        shapes = [(self.na, p.shape[2], p.shape[3], self.no) for p in preds]
        slots = self.assigner(shapes, self.strides, self.anchors, gt_boxes, gt_cls, m_valid)
        biases = [c.bias for c in self.m] if (self.training and self.bias_grad_in_loss) else None
        roots, self._roots = getattr(self, "_roots", None), None
        if roots is not None and not all(p.data_ptr() == r.data_ptr() for p, r in zip(preds, roots)):
            roots = None   # predictions that did not come from the last forward (a caller's own tensors)
        extra = {}
        if roots is not None:
            # bf16 logit gradient straight into the Detect convolutions' backward (ops.yolo_loss(heads=...)), or — with
            # HD_HEAD_GRAD_BF16=0 — a float32 gradient of the padded rows that autograd hands to ops.head_conv's backward
            if ops.HeadConvSwitch.bf16_grad and all(r.shape[1] % 8 == 0 for r in roots):
                extra = dict(heads=self._head_in)
            else:
                extra = dict(roots=roots)
        self._head_in = None
        return ops.yolo_loss(preds, self.strides, self.anchors, slots, gt_cls.shape[1], self.box_gain, self.obj_gain,
                             self.cls_gain, self.balance, biases=biases, focal=self.focal, **extra)

    @torch.no_grad()
    def predict(self, preds, conf_thr=0.25, iou_thr=0.45, max_nms=30000, max_det=300):
        # decode + per-class NMS in one pass over the raw head outputs (no decoded [B, n, no] tensor)
        return ops.yolo_detect(preds, self.strides, self.anchors, conf_thr, iou_thr, max_nms, max_det)


# =============================================================================== FasterRCNN heads
def _sample_mask(mask: torch.Tensor, k: int, gen_like: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Uniformly samples up to k True entries per row of `mask` [B,N] with fixed-size outputs and no
    host sync: returns (idx [B,k], valid [B,k])."""
    r = torch.rand(mask.shape, device=mask.device, dtype=torch.float32)
    key = torch.where(mask, r, torch.full_like(r, -1.0))
    k = min(k, mask.shape[1])
    val, idx = torch.topk(key, k, dim=1)
    return idx, val >= 0


def smooth_l1(x: torch.Tensor, beta: float) -> torch.Tensor:
    ax = x.abs()
    return torch.where(ax < beta, 0.5 * ax * ax / beta, ax - 0.5 * beta)


@HEADS.register()
class RPNHead(nn.Module):
    """Region Proposal Network over pyramid levels (README.md:8).  HIP hot path: anchor grid,
    fused IoU max/argmax assignment, delta encode, and proposals = per-level top-k + decode + NMS."""

    def __init__(self, in_channels: int = 256, strides: Sequence[int] = (4, 8, 16, 32, 64),
                 anchor_scale: float = 8.0, ratios: Sequence[float] = (0.5, 1.0, 2.0),
                 scales_per_octave: int = 1, sizes: Optional[Sequence[Sequence[float]]] = None,
                 pre_nms_topk: int = 2000, post_nms_topk: int = 1000,
                 nms_thr: float = 0.7, min_size: float = 0.0, batch_per_image: int = 256,
                 pos_fraction: float = 0.5, assigner=None):
        super().__init__()
        self.strides = tuple(float(s) for s in strides)
        self.ratios = tuple(ratios)
        # anchor sizes per level: explicit, or anchor_scale * stride * 2^(i/scales_per_octave)
        self.sizes = [list(map(float, z)) for z in sizes] if sizes is not None else \
            [[anchor_scale * s * 2 ** (i / scales_per_octave) for i in range(scales_per_octave)] for s in self.strides]
        self.A = len(self.ratios) * len(self.sizes[0])
        self.conv = ShadowConv2d(in_channels, in_channels, 3, padding=1)
        self.obj = ShadowConv2d(in_channels, self.A, 1)
        self.reg = ShadowConv2d(in_channels, self.A * 4, 1)
        for m in (self.conv, self.obj, self.reg):
            nn.init.normal_(m.weight, std=0.01)
            nn.init.zeros_(m.bias)
        self.pre, self.post, self.nms_thr, self.min_size = pre_nms_topk, post_nms_topk, nms_thr, min_size
        self.batch_per_image, self.pos_fraction = batch_per_image, pos_fraction
        self.assigner = ASSIGNERS.build(assigner or dict(type="MaxIoUAssigner", pos_iou_thr=0.7, neg_iou_thr=0.3,
                                                         match_low_quality=True))
        self._anchor_cache = {}

    def anchors_for(self, feats: List[torch.Tensor]):
        hw = tuple((f.shape[2], f.shape[3]) for f in feats)
        key = (hw, feats[0].device)
        if key not in self._anchor_cache:
            L = len(feats)
            base = torch.stack([ops.anchor_base(self.sizes[l], self.ratios) for l in range(L)])
            anchors = ops.anchor_grid(hw, self.strides[:L], base, 0.5, feats[0].device)
            self._anchor_cache[key] = (anchors, [h * w * self.A for h, w in hw])
        return self._anchor_cache[key]

    def forward(self, feats: List[torch.Tensor]):
        logits, deltas = [], []
        # objectness and box deltas come out of ONE 1x1 convolution with the two weights stacked (A + 4A outputs): the hidden
        # map is read once forward, and backward it receives one data gradient instead of two 256-channel maps plus their sum
        # The stack is padded to a multiple of 8 channels and run WITHOUT bias; ops.bias_act adds it in place and takes its
        # gradient in one pass: the library's bias gradient of a 15-channel channels-last map is a strided reduction that
        # took 1.1 ms at the 208 x 208 level of config 3 (profiles/r03_frcnn_step_order.txt history in DESIGN.md section 8).
        A = self.A
        wo, wr = self.obj.lp_weight(), self.reg.lp_weight()
        pad = (-5 * A) % 8
        ws = [wo, wr] + ([wo.new_zeros((pad,) + tuple(wo.shape[1:]))] if pad else [])
        bs = [self.obj.bias, self.reg.bias] + ([self.obj.bias.new_zeros(pad)] if pad else [])
        w, b = torch.cat(ws, 0), torch.cat(bs, 0)
        for f in feats:
            t = conv_bias_act(self.conv, f, "relu")  # bias + ReLU in one in-place pass after the bias-free convolution
            B = t.shape[0]
            o = F.conv2d(t, w.to(t.dtype) if w.dtype != t.dtype and not torch.is_autocast_enabled() else w)
            if ops.bias_act_supported(o, None):
                o = ops.bias_act(o, b, None, inplace=True)
            else:
                o = o + b.to(o.dtype)[None, :, None, None]
            o = o.float()
            # channels-last conv outputs: permute(0,2,3,1) is a free view; order = (y, x, anchor)
            logits.append(o[:, :A].permute(0, 2, 3, 1).reshape(B, -1))
            deltas.append(o[:, A:5 * A].permute(0, 2, 3, 1).reshape(B, -1, 4))
        return torch.cat(logits, 1), torch.cat(deltas, 1)

    @torch.no_grad()
    def proposals(self, logits, deltas, anchors, level_sizes, img_hw, training: bool):
        pre = self.pre
        post = self.post
        return ops.rpn_proposals(logits.detach().contiguous(), deltas.detach().contiguous(), anchors, level_sizes, pre,
                                 post, self.nms_thr, self.min_size, float(img_hw[0]), float(img_hw[1]))

    def loss(self, logits, deltas, anchors, gt_boxes, m_valid):
        B, NA = logits.shape
        with torch.no_grad():
            labels, matched, _ = self.assigner(anchors, gt_boxes, m_valid)
            n_pos_max = int(self.batch_per_image * self.pos_fraction)
            pos_idx, pos_ok = _sample_mask(labels == 1, n_pos_max, logits)
            n_pos = pos_ok.sum(1, keepdim=True)
            neg_idx, neg_ok = _sample_mask(labels == 0, self.batch_per_image, logits)
            rank = torch.arange(neg_idx.shape[1], device=logits.device)[None, :]
            neg_ok = neg_ok & (rank < (self.batch_per_image - n_pos))
            gt_idx = torch.gather(matched, 1, pos_idx).clamp(min=0).long()
            gt_sel = torch.gather(gt_boxes, 1, gt_idx[..., None].expand(-1, -1, 4))
            an_sel = anchors[pos_idx.reshape(-1)]
            tgt = ops.box_encode(an_sel.contiguous(), gt_sel.reshape(-1, 4).contiguous()).view(B, -1, 4)
            tgt = torch.where(pos_ok[..., None], tgt, torch.zeros_like(tgt))
            norm = (pos_ok.sum() + neg_ok.sum()).clamp(min=1).float()
        lp = torch.gather(logits, 1, pos_idx)
        ln = torch.gather(logits, 1, neg_idx)
        cls = (F.binary_cross_entropy_with_logits(lp, torch.ones_like(lp), reduction="none") * pos_ok).sum() + \
              (F.binary_cross_entropy_with_logits(ln, torch.zeros_like(ln), reduction="none") * neg_ok).sum()
        dp = torch.gather(deltas, 1, pos_idx[..., None].expand(-1, -1, 4))
        reg = (smooth_l1(dp - tgt, 1.0 / 9.0) * pos_ok[..., None]).sum()
        return cls / norm, reg / norm


@HEADS.register()
class RoIHead(nn.Module):
    """Second stage: RoIAlign (default) or quantised RoI max-pooling (`roi_op="pool"`, the
    "-RoIPooling" variant of README.md:73) over pyramid levels (HIP) + box head.  `decoupled=True` gives
    the Decoupled-Head of README.md:65,76 (separate classification FC tower and regression conv tower)."""

    def __init__(self, in_channels: int = 256, num_classes: int = 80, roi_size: int = 7, sampling_ratio: int = 2,
                 featmap_strides: Sequence[int] = (4, 8, 16, 32), k_min: int = 2, decoupled: bool = False,
                 fc_dim: int = 1024, batch_per_image: int = 128, pos_fraction: float = 0.25, assigner=None,
                 delta_std: Sequence[float] = (0.1, 0.1, 0.2, 0.2), score_thr: float = 0.05, nms_thr: float = 0.5,
                 max_det: int = 100, pre_nms_candidates: int = 4096, roi_op: str = "align"):
        super().__init__()
        if roi_op not in ("align", "pool"):
            raise ValueError(f"roi_op must be 'align' or 'pool', got {roi_op!r}")
        self.roi_op = roi_op
        self.nc, self.P, self.sr = num_classes, roi_size, sampling_ratio
        self.scales = tuple(1.0 / s for s in featmap_strides)
        self.k_min = k_min
        self.decoupled = decoupled
        flat = in_channels * roi_size * roi_size
        if not decoupled:
            self.shared = nn.Sequential(nn.Flatten(1), nn.Linear(flat, fc_dim), nn.ReLU(inplace=True),
                                        nn.Linear(fc_dim, fc_dim), nn.ReLU(inplace=True))
            self.cls_tower, self.reg_tower = nn.Identity(), nn.Identity()
            cls_in = reg_in = fc_dim
        else:
            self.shared = nn.Identity()
            self.cls_tower = nn.Sequential(nn.Flatten(1), nn.Linear(flat, fc_dim), nn.ReLU(inplace=True),
                                           nn.Linear(fc_dim, fc_dim), nn.ReLU(inplace=True))
            self.reg_tower = nn.Sequential(*[ConvBNAct(in_channels, in_channels, 3, 1, act="relu") for _ in range(4)],
                                           nn.AdaptiveAvgPool2d(1), nn.Flatten(1))
            cls_in, reg_in = fc_dim, in_channels
        self.cls = nn.Linear(cls_in, num_classes + 1)   # last index = background
        self.reg = nn.Linear(reg_in, num_classes * 4)
        nn.init.normal_(self.cls.weight, std=0.01)
        nn.init.normal_(self.reg.weight, std=0.001)
        nn.init.zeros_(self.cls.bias)
        nn.init.zeros_(self.reg.bias)
        self.batch_per_image, self.pos_fraction = batch_per_image, pos_fraction
        self.assigner = ASSIGNERS.build(assigner or dict(type="MaxIoUAssigner", pos_iou_thr=0.5, neg_iou_thr=0.5,
                                                         match_low_quality=False))
        self.register_buffer("delta_std", torch.tensor(delta_std, dtype=torch.float32), persistent=False)
        # placeholder box of padded samples; a buffer, not a literal: a host->device copy inside
        # sample()/predict() would be illegal while the step is being captured into a hipGraph
        self.register_buffer("unit_box", torch.tensor([0.0, 0.0, 1.0, 1.0], dtype=torch.float32), persistent=False)
        self.score_thr, self.nms_thr, self.max_det, self.pre_nms_candidates = score_thr, nms_thr, max_det, pre_nms_candidates

    # ---- RoI features ------------------------------------------------------------------------
    def extract(self, feats: List[torch.Tensor], rois5: torch.Tensor) -> torch.Tensor:
        fs = list(feats[: len(self.scales)])
        if self.roi_op == "pool":  # the quantised max-pool variant is float32-only
            return ops.roi_pool([f.float() for f in fs], rois5, self.P, self.scales[: len(fs)], self.k_min, 224.0, 4,
                                channels_last_out=self.decoupled)
        # the pyramid is consumed in its own dtype (bf16 under autocast: no float32 copy of P2..P5, bf16 RoI features
        # for the bf16 towers); accumulation inside the kernel is float32
        if fs[0].dtype not in (torch.float32, torch.bfloat16):
            fs = [f.float() for f in fs]
        return ops.roi_align(fs, rois5, self.P, self.scales[: len(fs)], self.sr, True, self.k_min, 224.0, 4,
                             channels_last_out=self.decoupled)

    def forward(self, feats, rois5):
        x = self.extract(feats, rois5)
        if not self.decoupled:
            h = self.shared(x)
            return self.cls(h), self.reg(h)
        return self.cls(self.cls_tower(x)), self.reg(self.reg_tower(x))

    # ---- training ----------------------------------------------------------------------------
    @torch.no_grad()
    def sample(self, rois, count, gt_boxes, gt_cls, m_valid):
        """rois [B,K,4] (+ the gt boxes appended) -> fixed-size sample per image (no host sync)."""
        B, K, _ = rois.shape
        M = gt_boxes.shape[1]
        cand = torch.cat([rois, gt_boxes], 1)                                   # [B, K+M, 4]
        ar = torch.arange(K + M, device=rois.device)[None, :]
        cand_ok = torch.where(ar < K, ar < count[:, None], (ar - K) < m_valid[:, None])
        labels, matched, _ = self.assigner(cand.contiguous(), gt_boxes, m_valid)
        fg = (labels == 1) & cand_ok
        bg = (labels == 0) & cand_ok
        n_fg_max = int(self.batch_per_image * self.pos_fraction)
        fg_idx, fg_ok = _sample_mask(fg, n_fg_max, rois)
        n_fg = fg_ok.sum(1, keepdim=True)
        bg_idx, bg_ok = _sample_mask(bg, self.batch_per_image, rois)
        rank = torch.arange(bg_idx.shape[1], device=rois.device)[None, :]
        bg_ok = bg_ok & (rank < (self.batch_per_image - n_fg))
        idx = torch.cat([fg_idx, bg_idx], 1)                                    # [B, S]
        ok = torch.cat([fg_ok, bg_ok], 1)
        is_fg = torch.cat([fg_ok, torch.zeros_like(bg_ok)], 1)
        boxes = torch.gather(cand, 1, idx[..., None].expand(-1, -1, 4))
        gt_i = torch.gather(matched, 1, idx).clamp(min=0).long()
        gt_b = torch.gather(gt_boxes, 1, gt_i[..., None].expand(-1, -1, 4))
        cls = torch.gather(gt_cls.long(), 1, gt_i)
        cls = torch.where(is_fg, cls, torch.full_like(cls, self.nc))
        S = idx.shape[1]
        tgt = ops.box_encode(boxes.reshape(-1, 4).contiguous(), gt_b.reshape(-1, 4).contiguous()).view(B, S, 4)
        tgt = torch.where(is_fg[..., None], tgt / self.delta_std, torch.zeros_like(tgt))
        bi = torch.arange(B, device=rois.device, dtype=torch.float32)[:, None, None].expand(B, S, 1)
        # invalid samples keep a harmless 1x1 box so RoIAlign stays in range; their loss weight is 0
        boxes = torch.where(ok[..., None], boxes, self.unit_box.expand_as(boxes))
        rois5 = torch.cat([bi, boxes], 2).reshape(B * S, 5).contiguous()
        return rois5, cls.reshape(-1), tgt.reshape(-1, 4), ok.reshape(-1), is_fg.reshape(-1)

    def loss(self, cls_logits, deltas, cls_t, reg_t, ok, is_fg):
        okf = ok.float()
        n = okf.sum().clamp(min=1)
        lcls = (F.cross_entropy(cls_logits.float(), cls_t, reduction="none") * okf).sum() / n
        R = deltas.shape[0]
        d = deltas.float().view(R, self.nc, 4)
        sel = d[torch.arange(R, device=d.device), cls_t.clamp(max=self.nc - 1)]
        lreg = (smooth_l1(sel - reg_t, 1.0) * is_fg[:, None].float()).sum() / n
        return lcls, lreg

    # ---- inference ---------------------------------------------------------------------------
    @torch.no_grad()
    def predict(self, feats, rois, count, img_hw):
        B, K, _ = rois.shape
        bi = torch.arange(B, device=rois.device, dtype=torch.float32)[:, None, None].expand(B, K, 1)
        ar = torch.arange(K, device=rois.device)[None, :]
        valid = ar < count[:, None]
        safe = torch.where(valid[..., None], rois, self.unit_box.expand_as(rois))
        cls_logits, deltas = self.forward(feats, torch.cat([bi, safe], 2).reshape(B * K, 5).contiguous())
        prob = F.softmax(cls_logits.float(), 1)[:, : self.nc].reshape(B, K, self.nc)
        prob = torch.where(valid[..., None], prob, torch.zeros_like(prob))
        d = deltas.float().view(B * K, self.nc, 4) * self.delta_std
        src = safe.reshape(B * K, 1, 4).expand(-1, self.nc, -1).reshape(-1, 4).contiguous()
        boxes = ops.box_decode(src, d.reshape(-1, 4).contiguous(), float(img_hw[0]), float(img_hw[1])).view(B, K * self.nc, 4)
        scores = prob.reshape(B, K * self.nc).contiguous()
        cls = torch.arange(self.nc, device=rois.device, dtype=torch.int32).repeat(K)[None, :].expand(B, -1).contiguous()
        keep, cnt = ops.nms_batched(boxes.contiguous(), scores, cls, None, self.score_thr, self.nms_thr,
                                    self.pre_nms_candidates, self.max_det)
        k = keep.clamp(min=0).long()
        det = torch.cat([torch.gather(boxes, 1, k[..., None].expand(-1, -1, 4)), torch.gather(scores, 1, k)[..., None],
                         torch.gather(cls, 1, k).float()[..., None]], 2)
        det = torch.where((keep >= 0)[..., None], det, torch.zeros_like(det))
        return det, cnt
