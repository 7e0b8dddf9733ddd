"""The config-file CLI on the HIP ops (reference README.md:12-14): train -> eval -> test driven by
one yaml each, for both detector families."""
import json
import os

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


def _cfg(tmp_path, name, d):
    p = tmp_path / name
    with open(p, "w") as f:
        yaml.safe_dump(d, f)
    return str(p)


def test_cli_yolov5_train_eval_test_on_hip(gpu, tmp_path):
    from heltondetection_amd import cli
    from heltondetection_amd.logger import read_scalars
    cfg = _cfg(tmp_path, "y.yaml", {
        "model": {"type": "YOLOv5", "variant": "s"},
        "data": {"type": "synthetic_coco", "n_images": 32, "val_images": 8, "size": 160, "num_classes": 4, "max_boxes": 6,
                 "paint": True, "batch": 8},
        "train": {"epochs": 10, "lr": 0.002, "warmup_steps": 2, "amp": "bf16", "hip_graph": True, "mosaic": 0.5, "log_every": 4,
                  "eval_every": 10},
        "eval": {"conf_thr": 0.001, "iou_thr": 0.6, "max_det": 100, "max_nms": 3000},
    })
    work = str(tmp_path / "run")
    out = cli.main(["train", "--config", cfg, "--work-dir", work])
    assert out["steps"] == 40 and out["hip_graph"] is True, out
    losses = [v for _, v in read_scalars(os.path.join(work, "log.jsonl"), "loss")]
    # Mosaic (p = 0.5) makes single readings noisy: judge the trend on the ends of the curve
    assert len(losses) == 10 and min(losses[-3:]) < max(losses[:3]) and all(l == l for l in losses), losses
    assert out["metrics"] is not None and 0.0 <= out["metrics"]["mAP50"] <= 1.0        # eval_every ran inside the loop
    m = cli.main(["eval", "--config", cfg, "--work-dir", work])
    assert m["images"] == 8 and os.path.exists(m["results"])
    # the in-loop evaluation and the stand-alone one see the same weights -> the same detections, up to the bf16
    # convolution solvers MIOpen's find mode picks per process (a low-confidence box can change rank)
    assert m["detections"] == out["metrics"]["detections"] and abs(m["mAP50"] - out["metrics"]["mAP50"]) < 1e-3
    t = cli.main(["test", "--config", cfg, "--work-dir", work, "--set", "test.conf_thr=0.05"])
    assert t["images"] == 8 and os.path.exists(t["path"])
    dets = json.load(open(t["path"]))
    assert all(all(0.0 <= r[4] <= 1.0 and 0 <= r[5] < 4 for r in d["detections"]) for d in dets)


def test_cli_faster_rcnn_train_eval_on_hip(gpu, tmp_path):
    from heltondetection_amd import cli
    cfg = _cfg(tmp_path, "f.yaml", {
        "model": {"type": "FasterRCNN", "neck": {"type": "PAFPN", "out_channels": 64},
                  "rpn_head": {"type": "RPNHead", "in_channels": 64, "pre_nms_topk": 300, "post_nms_topk": 100},
                  "roi_head": {"type": "RoIHead", "in_channels": 64, "decoupled": True, "batch_per_image": 32, "fc_dim": 128}},
        "data": {"type": "synthetic_coco", "n_images": 8, "val_images": 4, "size": 256, "num_classes": 3, "max_boxes": 5,
                 "paint": True, "batch": 2},
        "train": {"epochs": 2, "lr": 0.0005, "warmup_steps": 1, "amp": "bf16", "hip_graph": False, "log_every": 2},
        "eval": {"conf_thr": 0.01, "iou_thr": 0.5, "max_det": 50},
    })
    work = str(tmp_path / "run")
    out = cli.main(["train", "--config", cfg, "--work-dir", work])
    assert out["steps"] == 8 and out["last_loss"] == out["last_loss"]
    m = cli.main(["eval", "--config", cfg, "--work-dir", work])
    assert m["images"] == 4 and 0.0 <= m["mAP50"] <= 1.0
    from heltondetection_amd.detectors import build_detector
    state = torch.load(os.path.join(work, "last.pt"), map_location="cpu")
    assert state["meta"]["step"] == 8 and "optimizer" in state
    assert any(k.endswith("num_batches_tracked") for k in state["model"])   # stock BatchNorm keys are present
