"""Generates tests/golden/hotpath_v1.npz: seeded inputs + expected outputs of every hot-path op.

The reference snapshot has no source and no fixtures, so these vectors are
produced by THIS repo's CPU oracle (oracle/hd_oracle.c) and cross-checked at generation time
against the independent restatement (oracle/torch_ref.py); generation aborts if the two disagree.
They pin the oracle against silent drift and give the GPU tests a data-only fixture.
Run: python tests/golden/make_golden.py   (seed 0 everywhere, as SURVEY.md §8d prescribes)
"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c_oracle as co, torch_ref as tr  # noqa: E402
from tests import helpers as H  # noqa: E402
from heltondetection_amd import synthetic as syn  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_v1.npz")


def agree(a, b, tol=0.0, what=""):
    a, b = np.asarray(a), np.asarray(b)
    if tol == 0.0:
        assert np.array_equal(a, b), f"oracles disagree (exact) on {what}"
    else:
        assert np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= tol * (np.abs(b).max() + 1e-12), \
            f"oracles disagree on {what}"


def main():
    d = {}
    # anchors
    strides = [8, 16, 32]
    hw = [(12, 12), (6, 6), (3, 3)]
    sizes = [[32.0], [64.0], [128.0]]
    base = np.stack([co.anchor_base(s, [0.5, 1.0, 2.0]) for s in sizes])
    agree(base, torch.stack([tr.anchor_base(s, [0.5, 1.0, 2.0]) for s in sizes]).numpy(), 0.0, "anchor_base")
    anchors = co.anchor_grid([h for h, _ in hw], [w for _, w in hw], strides, base, 0.5)
    agree(anchors, tr.anchor_grid([h for h, _ in hw], [w for _, w in hw], strides, list(base), 0.5).numpy(), 0.0, "anchor_grid")
    d["anchor_base"], d["anchor_grid"] = base, anchors
    # IoU + assign
    a = H.boxes_batch(2, 200, 96, 0).numpy()
    b = H.boxes_batch(2, 9, 96, 1).numpy()
    mv = np.array([9, 5], np.int32)
    iou = co.pairwise_iou(a, b, mv)
    agree(iou, tr.pairwise_iou(torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(mv)).numpy(), 0.0, "pairwise_iou")
    asg = co.iou_assign(a, b, mv)
    for x, y in zip(asg, tr.iou_assign(torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(mv))):
        agree(x, y.numpy(), 0.0, "iou_assign")
    d.update(iou_a=a, iou_b=b, iou_mv=mv, iou_out=iou, asg_max=asg[0], asg_arg=asg[1], asg_gmax=asg[2], asg_garg=asg[3])
    # NMS
    nb = H.clustered_boxes(2, 300, 128, 2, n_centers=12, jitter=4.0).numpy()
    ns = H.distinct_scores(2, 300, 3).numpy()
    ncls = torch.randint(0, 3, (2, 300), generator=syn.make_generator(4), dtype=torch.int32).numpy()
    keep, cnt = co.nms_batched(nb, ns, ncls, None, 0.1, 0.5, 0, 64)
    k2, c2 = tr.nms_batched(torch.from_numpy(nb), torch.from_numpy(ns), torch.from_numpy(ncls), None, 0.1, 0.5, 0, 64)
    agree(keep, k2.numpy(), 0.0, "nms keep"); agree(cnt, c2.numpy(), 0.0, "nms count")
    d.update(nms_boxes=nb, nms_scores=ns, nms_cls=ncls, nms_keep=keep, nms_count=cnt)
    # RPN proposals (anchors above)
    g = syn.make_generator(5)
    NA = anchors.shape[0]
    logits = torch.randn((2, NA), generator=g).numpy()
    deltas = (torch.randn((2, NA, 4), generator=g) * 0.2).numpy()
    lsz = [h * w * 3 for h, w in hw]
    rois, rsc, rcnt, ridx = co.rpn_proposals(logits, deltas, anchors, lsz, 100, 40, 0.7, 2.0, 96.0, 96.0, return_index=True)
    r2 = tr.rpn_proposals(torch.from_numpy(logits), torch.from_numpy(deltas), torch.from_numpy(anchors), lsz, 100, 40, 0.7, 2.0, 96.0, 96.0)
    agree(rcnt, r2[2].numpy(), 0.0, "rpn count"); agree(rois, r2[0].numpy(), 1e-5, "rpn rois"); agree(rsc, r2[1].numpy(), 1e-5, "rpn scores")
    agree(ridx, r2[3].numpy(), 0.0, "rpn kept anchor indices")
    d.update(rpn_logits=logits, rpn_deltas=deltas, rpn_rois=rois, rpn_scores=rsc, rpn_count=rcnt, rpn_index=ridx)
    # RoIAlign
    g = syn.make_generator(6)
    feats = [torch.randn((2, 8, 96 // s, 96 // s), generator=g).numpy() for s in strides]
    rb = H.boxes_batch(1, 12, 96, 7, 4.0, 80.0)[0]
    rr = torch.cat([torch.randint(0, 2, (12, 1), generator=g).float(), rb], 1).numpy()
    scales = [1.0 / s for s in strides]
    ra = co.roi_align_fwd(feats, rr, 3, scales, 2, True, 3, 56.0, 4)
    ra2 = tr.roi_align([torch.from_numpy(f) for f in feats], torch.from_numpy(rr), 3, scales, 2, True, 3, 56.0, 4).numpy()
    agree(ra, ra2, 1e-5, "roi_align")
    lv = co.roi_level_map(rr, 3, 3, 56.0, 4)
    agree(lv, tr.roi_level_map(torch.from_numpy(rr), 3, 3, 56.0, 4).numpy(), 0.0, "roi levels")
    rp, rarg = co.roi_pool_fwd(feats, rr, 3, scales, 3, 56.0, 4)
    rp2 = tr.roi_pool([torch.from_numpy(f) for f in feats], torch.from_numpy(rr), 3, scales, 3, 56.0, 4).numpy()
    agree(rp, rp2, 0.0, "roi_pool")
    d.update(roi_f0=feats[0], roi_f1=feats[1], roi_f2=feats[2], roi_rois=rr, roi_out=ra, roi_levels=lv,
             roi_pool_out=rp, roi_pool_arg=rarg)
    # YOLOv5
    preds = [p.numpy() for p in H.yolo_preds(2, 64, 9, 8)]
    gtb, gtc, gmv = syn.make_targets(2, 64, syn.make_generator(9), "coco", 4, 12)
    gmv = torch.clamp(gmv, max=12)
    dec = co.yolo_decode(preds, H.YOLO_STRIDES, H.YOLO_ANCHORS)
    agree(dec, tr.yolo_decode([torch.from_numpy(p) for p in preds], H.YOLO_STRIDES, H.YOLO_ANCHORS).numpy(), 1e-6, "decode")
    shapes = H.yolo_shapes(64, 9)
    cell, tbox, scls = co.yolo_assign(shapes, H.YOLO_STRIDES, H.YOLO_ANCHORS, gtb.numpy(), gtc.numpy(), gmv.numpy(), 4.0)
    loss, grads, _, _ = co.yolo_loss(preds, H.YOLO_STRIDES, H.YOLO_ANCHORS, cell, tbox, scls, 12)
    pt = [torch.from_numpy(p).requires_grad_(True) for p in preds]
    lt = tr.yolo_loss(pt, H.YOLO_STRIDES, H.YOLO_ANCHORS, gtb, gtc, gmv)
    lt[0].backward()
    agree(loss, np.array([float(x) for x in lt], np.float32), 1e-5, "yolo loss")
    for gg, p in zip(grads, pt):
        agree(gg, p.grad.numpy(), 1e-4, "yolo grad")
    fl_loss, fl_grads, _, _ = co.yolo_loss(preds, H.YOLO_STRIDES, H.YOLO_ANCHORS, cell, tbox, scls, 12,
                                           focal=(1.5, 0.25, ("cls", "obj")))
    pf = [torch.from_numpy(p).requires_grad_(True) for p in preds]
    lf = tr.yolo_loss(pf, H.YOLO_STRIDES, H.YOLO_ANCHORS, gtb, gtc, gmv, focal=(1.5, 0.25, ("cls", "obj")))
    lf[0].backward()
    agree(fl_loss, np.array([float(x.detach()) for x in lf], np.float32), 1e-5, "focal yolo loss")
    for gg, p in zip(fl_grads, pf):
        agree(gg, p.grad.numpy(), 1e-4, "focal yolo grad")
    d.update(y_focal_loss=fl_loss, y_focal_g0=fl_grads[0], y_focal_g2=fl_grads[2])
    det, dcnt = co.yolo_nms(dec, 0.3, 0.45, 500, 20)
    d2, c2 = tr.yolo_nms(torch.from_numpy(dec), 0.3, 0.45, 500, 20)
    agree(det, d2.numpy(), 0.0, "yolo nms"); agree(dcnt, c2.numpy(), 0.0, "yolo nms count")
    d.update(y_p0=preds[0], y_p1=preds[1], y_p2=preds[2], y_gtb=gtb.numpy(), y_gtc=gtc.numpy(), y_mv=gmv.numpy(),
             y_dec=dec, y_cell=cell, y_tbox=tbox, y_cls=scls, y_loss=loss, y_g0=grads[0], y_g1=grads[1], y_g2=grads[2],
             y_det=det, y_dcnt=dcnt)
    # CIoU
    g = syn.make_generator(10)
    c = torch.rand((64, 2), generator=g) * 60 + 20
    wh = torch.rand((64, 2), generator=g) * 30 + 4
    p = torch.cat([c - wh / 2, c + wh / 2], 1)
    c2_ = c + torch.randn((64, 2), generator=g) * 6
    wh2 = wh * torch.exp(torch.randn((64, 2), generator=g) * 0.3)
    t = torch.cat([c2_ - wh2 / 2, c2_ + wh2 / 2], 1)
    ci = co.ciou_fwd(p.numpy(), t.numpy())
    cg = co.ciou_bwd(p.numpy(), t.numpy(), np.ones(64, np.float32))
    pp = p.clone().double().requires_grad_(True)
    cc = tr.ciou_xyxy(pp, t.double()); cc.sum().backward()
    agree(ci, cc.detach().numpy(), 1e-5, "ciou"); agree(cg, pp.grad.numpy(), 1e-3, "ciou grad")
    d.update(ciou_p=p.numpy(), ciou_t=t.numpy(), ciou_out=ci, ciou_grad=cg)
    # WBF
    g = syn.make_generator(11)
    wb0 = H.clustered_boxes(2, 40, 128, 12, n_centers=10, jitter=1.5)
    wb = torch.cat([wb0 + torch.randn(wb0.shape, generator=g) for _ in range(3)], 1).clamp(0, 128)
    ws_ = torch.rand((2, 120), generator=g)
    wl = torch.randint(0, 3, (2, 40), generator=g, dtype=torch.int32).repeat(1, 3)
    wm = torch.ones((2, 120))
    wo, wc = co.wbf(wb.numpy(), ws_.numpy(), wl.numpy(), wm.numpy(), None, 3.0, 0.55, 0.05)
    for bi in range(2):
        o2 = tr.wbf_single(wb[bi], ws_[bi], wl[bi], wm[bi], 3.0, 0.55, 0.05).numpy()
        assert o2.shape[0] == wc[bi], "oracles disagree on wbf count"
        agree(wo[bi, : wc[bi]], o2, 1e-5, "wbf")
    d.update(wbf_boxes=wb.numpy(), wbf_scores=ws_.numpy(), wbf_labels=wl.numpy(), wbf_out=wo, wbf_count=wc)
    # Mosaic-4
    g = syn.make_generator(12)
    mi = syn.make_images(4, 32, g)
    mb, mc, mm = syn.make_targets(4, 32, g, "coco", 5, 6)
    msrc = torch.stack([torch.tensor([b] + torch.randint(0, 4, (3,), generator=g).tolist()) for b in range(4)]).to(torch.int32)
    mctr = torch.randint(16, 49, (4, 2), generator=g, dtype=torch.int32)
    mapl = torch.tensor([1, 0, 1, 1], dtype=torch.int32)
    mo = co.mosaic4(mi.numpy(), mb.numpy(), mc.numpy(), mm.numpy(), msrc.numpy(), mctr.numpy(), mapl.numpy(), min_wh=1.0)
    mo2 = tr.mosaic4(mi, mb, mc, mm, msrc, mctr, mapl, min_wh=1.0)
    for a_, b_, nm in zip(mo, mo2, ("image", "boxes", "cls", "count")):
        agree(a_, b_.numpy(), 0.0, "mosaic " + nm)
    d.update(mos_img=mi.numpy(), mos_boxes=mb.numpy(), mos_cls=mc.numpy(), mos_mv=mm.numpy(), mos_src=msrc.numpy(),
             mos_center=mctr.numpy(), mos_apply=mapl.numpy(), mos_out=mo[0], mos_out_boxes=mo[1], mos_out_cls=mo[2],
             mos_out_count=mo[3])
    np.savez_compressed(OUT, **d)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(d), "arrays")


if __name__ == "__main__":
    main()
