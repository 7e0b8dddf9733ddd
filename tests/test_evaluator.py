import numpy as np
import pytest

from heltondetection_amd.evaluator import MeanAP


def test_perfect_and_empty_detections():
    ev = MeanAP(3)
    gt = np.array([[10, 10, 50, 50], [60, 60, 100, 120]], np.float32)
    cls = np.array([0, 2])
    det = np.concatenate([gt, np.array([[0.9, 0], [0.8, 2]], np.float32)], 1)
    ev.add_image(det, gt, cls)
    r = ev.compute()
    assert abs(r["mAP50"] - 1.0) < 1e-9 and abs(r["mAP50_95"] - 1.0) < 1e-9 and r["classes_with_gt"] == 2
    ev2 = MeanAP(3)
    ev2.add_image(np.zeros((0, 6), np.float32), gt, cls)
    assert ev2.compute()["mAP50"] == 0.0


def test_known_precision_recall_curve():
    # one class, 2 gt; detections by score: TP, FP, TP  ->  P/R points (1, .5) (.5, .5) (.67, 1)
    ev = MeanAP(1, iou_thresholds=(0.5,))
    gt = np.array([[0, 0, 10, 10], [20, 20, 30, 30]], np.float32)
    det = np.array([[0, 0, 10, 10, 0.9, 0], [50, 50, 60, 60, 0.8, 0], [20, 20, 30, 30, 0.7, 0]], np.float32)
    ev.add_image(det, gt, np.array([0, 0]))
    # interpolated precision: 1.0 for recall <= 0.5 (51 points), 2/3 for recall in (0.5, 1] (50 points)
    expect = (51 * 1.0 + 50 * (2 / 3)) / 101
    assert abs(ev.compute()["mAP50"] - expect) < 1e-9


def test_localisation_quality_separates_map50_from_map5095():
    ev = MeanAP(1)
    gt = np.array([[0, 0, 100, 100]], np.float32)
    det = np.array([[0, 0, 100, 80, 0.9, 0]], np.float32)  # IoU 0.8
    ev.add_image(det, gt, np.array([0]))
    r = ev.compute()
    assert abs(r["mAP50"] - 1.0) < 1e-9
    assert abs(r["mAP50_95"] - 0.7) < 1e-9  # passes 0.50..0.80 (7 of 10 thresholds)


def test_duplicate_detections_count_as_false_positives():
    ev = MeanAP(1, iou_thresholds=(0.5,))
    gt = np.array([[0, 0, 10, 10]], np.float32)
    det = np.array([[0, 0, 10, 10, 0.9, 0], [0, 0, 10, 10, 0.8, 0]], np.float32)
    ev.add_image(det, gt, np.array([0]))
    assert abs(ev.compute()["mAP50"] - 1.0) < 1e-9  # recall 1 reached at precision 1; the later FP does not lower AP


def test_coco_format_round_trip_and_json_evaluation(tmp_path):
    """COCO-format writer -> reader -> padded batches -> results json -> evaluator (README.md:10,61)."""
    import json
    import torch
    from heltondetection_amd import coco_io, synthetic as syn
    ann = coco_io.write_synthetic_coco(str(tmp_path), n_images=5, size=64, num_classes=6, seed=3, max_boxes=7)
    ds = coco_io.CocoDetection(ann)
    assert len(ds) == 5 and ds.num_classes == 6
    # the reader reproduces what the generator produced (same seed, same order)
    g = syn.make_generator(3)
    for i in range(5):
        px_ref = (syn.make_images(1, 64, g)[0] * 255.0).round().clamp(0, 255) / 255.0
        gtb, gtc, mv = syn.make_targets(1, 64, g, "coco", 6, 7)
        px, b, c = ds[i]
        assert torch.equal(px, px_ref)
        n = int(mv[0])
        assert b.shape[0] == n and torch.allclose(b, gtb[0, :n], atol=1e-4) and torch.equal(c, gtc[0, :n])
    batches = list(ds.batches(2))
    assert [bt[0].shape[0] for bt in batches] == [2, 2, 1]
    imgs, gb, gc, mv, ids = batches[0]
    assert imgs.shape == (2, 3, 64, 64) and gb.shape[1] == gc.shape[1] == ds.max_boxes and ids == [1, 2]
    # perfect detections (ground truth echoed as results, padded detector format) -> AP 1; shifted ones -> lower
    results, shifted = [], []
    for imgs, gb, gc, mv, ids in ds.batches(2):
        det = torch.zeros((imgs.shape[0], ds.max_boxes, 6))
        det[:, :, :4], det[:, :, 4], det[:, :, 5] = gb, 0.9, gc.float()
        results += coco_io.detections_to_coco(det, mv, ids, ds.cat_ids)
        det2 = det.clone()
        det2[:, :, [0, 2]] += 40.0
        shifted += coco_io.detections_to_coco(det2, mv, ids, ds.cat_ids)
    res_path = tmp_path / "results.json"
    res_path.write_text(json.dumps(results))
    good = coco_io.evaluate_coco_json(ann, str(res_path))
    bad = coco_io.evaluate_coco_json(ann, shifted)
    assert good["mAP50"] == pytest.approx(1.0) and good["mAP50_95"] == pytest.approx(1.0)
    assert bad["mAP50"] < 0.5
    assert all(r["category_id"] >= 1 and len(r["bbox"]) == 4 for r in results)
