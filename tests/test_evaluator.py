import numpy as np

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
