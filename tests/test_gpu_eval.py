"""mAP50 of the HIP post-processing path vs the oracle's on identical synthetic boxes (north_star:
"mAP50 within +-0.2 of the reference on identical synthetic boxes" — here the checker is our CPU
restatement, parity unpinned), and the TTA -> WBF pipeline end to end."""
import numpy as np
import pytest
import torch

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu


def _synthetic_decoded(B, n, nc, size, seed):
    """Decoded head rows: a few rows sit on ground-truth boxes (noisy, confident), the rest are clutter."""
    from heltondetection_amd import synthetic as syn
    g = syn.make_generator(seed)
    gtb, gtc, mv = syn.make_targets(B, size, g, "coco", nc, 40)
    dec = torch.zeros((B, n, 5 + nc))
    clutter = syn.random_boxes(B * n, float(size), g, 8.0, size / 3).view(B, n, 4)
    dec[..., 0] = (clutter[..., 0] + clutter[..., 2]) / 2
    dec[..., 1] = (clutter[..., 1] + clutter[..., 3]) / 2
    dec[..., 2] = clutter[..., 2] - clutter[..., 0]
    dec[..., 3] = clutter[..., 3] - clutter[..., 1]
    dec[..., 4] = torch.rand((B, n), generator=g) * 0.35
    dec[..., 5:] = torch.rand((B, n, nc), generator=g) * 0.6
    for b in range(B):
        for t in range(int(mv[b])):
            for rep in range(3):  # three noisy copies per gt: NMS has to merge them
                r = int(torch.randint(0, n, (1,), generator=g))
                box = gtb[b, t] + torch.randn(4, generator=g) * (2.0 + 3.0 * rep)
                dec[b, r, 0] = (box[0] + box[2]) / 2
                dec[b, r, 1] = (box[1] + box[3]) / 2
                dec[b, r, 2] = (box[2] - box[0]).clamp(min=2)
                dec[b, r, 3] = (box[3] - box[1]).clamp(min=2)
                dec[b, r, 4] = 0.6 + 0.39 * torch.rand(1, generator=g)
                dec[b, r, 5:] = torch.rand(nc, generator=g) * 0.2
                dec[b, r, 5 + int(gtc[b, t])] = 0.7 + 0.29 * torch.rand(1, generator=g)
    return dec, gtb, gtc, mv


def test_map50_hip_postprocess_equals_oracle(gpu):
    from heltondetection_amd import ops
    from heltondetection_amd.evaluator import MeanAP
    B, n, nc, size = 4, 3000, 12, 640
    dec, gtb, gtc, mv = _synthetic_decoded(B, n, nc, size, 7)
    det, cnt = ops.yolo_nms(dec.to(gpu), 0.05, 0.5, 3000, 200)
    rdet, rcnt = co.yolo_nms(dec.numpy(), 0.05, 0.5, 3000, 200)
    ev_h, ev_o = MeanAP(nc), MeanAP(nc)
    ev_h.add_batch(det, cnt, gtb, gtc, mv)
    ev_o.add_batch(rdet, rcnt, gtb.numpy(), gtc.numpy(), mv.numpy())
    h, o = ev_h.compute(), ev_o.compute()
    assert o["mAP50"] > 0.5                      # the synthetic task is meaningful
    assert abs(100 * h["mAP50"] - 100 * o["mAP50"]) <= 0.2
    assert np.array_equal(det.cpu().numpy(), rdet)  # in fact the detections are identical


def test_tta_wbf_pipeline(gpu):
    """A noisy fake detector: each TTA view sees the gt boxes with independent jitter; WBF fusion must give
    at least the single-view mAP50-95 (averaging the jitter tightens localisation)."""
    from heltondetection_amd import synthetic as syn
    from heltondetection_amd.evaluator import MeanAP
    from heltondetection_amd.tta import tta_detect
    B, nc, size, D = 4, 6, 320, 64
    g = syn.make_generator(11)
    gtb, gtc, mv = syn.make_targets(B, size, g, "coco", nc, 20)
    images = torch.rand((B, 3, size, size), generator=g).to(gpu)
    state = {"calls": 0}

    def detect(x):
        state["calls"] += 1
        gen = torch.Generator().manual_seed(100 + state["calls"])
        h, w = x.shape[-2:]
        # recover whether this view is flipped by comparing with the (resized) original
        ref = torch.nn.functional.interpolate(images, size=(h, w), mode="bilinear", align_corners=False) if (h, w) != (size, size) else images
        flipped = not torch.allclose(x, ref, atol=1e-5)
        det = torch.zeros((B, D, 6))
        cnt = torch.zeros((B,), dtype=torch.int32)
        for b in range(B):
            m = int(mv[b])
            bx = gtb[b, :m] * (w / size) + torch.randn((m, 4), generator=gen) * 4.0
            if flipped:
                bx = torch.stack([w - bx[:, 2], bx[:, 1], w - bx[:, 0], bx[:, 3]], 1)
            det[b, :m, :4] = bx
            det[b, :m, 4] = 0.5 + 0.4 * torch.rand(m, generator=gen)
            det[b, :m, 5] = gtc[b, :m].float()
            cnt[b] = m
        return det.to(gpu), cnt.to(gpu)

    single, scnt = detect(images)
    state["calls"] = 0
    fused, fcnt = tta_detect(detect, images, scales=(1.0, 0.75), flips=(False, True), iou_thr=0.55, skip_thr=0.05)
    assert state["calls"] == 4
    ev_s, ev_f = MeanAP(nc), MeanAP(nc)
    ev_s.add_batch(single, scnt, gtb, gtc, mv)
    ev_f.add_batch(fused, fcnt, gtb, gtc, mv)
    s, f = ev_s.compute(), ev_f.compute()
    assert f["mAP50"] >= s["mAP50"] - 0.02
    assert f["mAP50_95"] > s["mAP50_95"]
    assert int(fcnt.sum()) < 0.75 * 4 * int(mv.sum())  # the four views were merged, not concatenated


def test_end_to_end_example_trains_and_scores(gpu, tmp_path):
    """examples/train_eval_synthetic.py: COCO-format dataset on disk -> reader -> graph-replayed training ->
    graph-replayed inference -> results json -> mAP.  Overfitting 16 noise images with 3 box classes must lift
    mAP50 well above zero and lower the loss; this is the whole chain a user of the reference would run."""
    import importlib.util, os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "train_eval_synthetic.py")
    spec = importlib.util.spec_from_file_location("hd_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = mod.main(["--workdir", str(tmp_path), "--images", "16", "--size", "128", "--classes", "3", "--batch", "8",
                    "--epochs", "100", "--mosaic", "0.0", "--lr", "3e-3"])
    assert out["hip_graph"] and out["last_loss"] < 0.6 * out["first_loss"]
    assert out["detections"] > 0 and out["params_M"] > 5
    assert out["mAP50"] > 0.15, out   # typically 0.3-0.5 after 200 steps; an untrained model scores ~0
