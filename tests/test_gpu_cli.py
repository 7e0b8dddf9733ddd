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
    base = {
        "model": {"type": "YOLOv5", "variant": "s"},
        # 8 training images = ONE fixed batch per epoch (the same 8 images every step, order aside), no Mosaic: the loss
        # of a working train -> log chain must come down, not merely stay finite
        "data": {"type": "synthetic_coco", "n_images": 8, "val_images": 8, "size": 160, "num_classes": 4, "max_boxes": 6,
                 "paint": True, "batch": 8},
        "train": {"epochs": 40, "lr": 0.002, "warmup_steps": 2, "amp": "bf16", "hip_graph": True, "mosaic": 0.0, "log_every": 4,
                  "eval_every": 40},
        "eval": {"conf_thr": 0.001, "iou_thr": 0.6, "max_det": 100, "max_nms": 3000},
    }
    cfg = _cfg(tmp_path, "y.yaml", base)
    work = str(tmp_path / "run")
    out = cli.main(["train", "--config", cfg, "--work-dir", work])
    assert out["steps"] == 40 and out["hip_graph"] is True, out
    losses = [v for _, v in read_scalars(os.path.join(work, "log.jsonl"), "loss")]
    assert len(losses) == 10 and all(l == l for l in losses), losses
    assert losses[-1] < 0.9 * losses[0], losses                 # a real trend on a fixed batch
    # the Mosaic (p = 0.5) + hipGraph path through the same CLI: short run, replayed, finite
    mcfg = dict(base, train=dict(base["train"], epochs=12, mosaic=0.5, eval_every=0, log_every=3))
    mout = cli.main(["train", "--config", _cfg(tmp_path, "ym.yaml", mcfg), "--work-dir", str(tmp_path / "run_mosaic")])
    assert mout["steps"] == 12 and mout["hip_graph"] is True and mout["last_loss"] == mout["last_loss"], mout
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
