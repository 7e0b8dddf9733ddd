"""config files -> registry, the one-line CLI (train / eval / test), the eval runner and the logger
(reference README.md:12-14).  CPU: the product has no CPU ops, so — exactly like tests/test_plumbing_cpu.py — the
three op calls the YOLOv5 head makes are monkeypatched with the oracle (test infrastructure); everything else
(config resolution, data set on disk, training loop, checkpoint, sharded evaluation, results / metrics / log files)
is the product code.  The same flow on the HIP ops is tests/test_gpu_cli.py."""
import json
import os

import pytest
import torch
import yaml

from oracle import torch_ref as tr


def _write(path, text):
    with open(path, "w") as f:
        f.write(text)
    return str(path)


def test_config_formats_inheritance_and_overrides(tmp_path):
    from heltondetection_amd.config import ConfigError, load_config
    base = _write(tmp_path / "base.yaml", "data: {size: 320, batch: 4}\ntrain: {lr: 0.01, epochs: 3}\nmodel: {type: YOLOv5, variant: n}\n")
    child = _write(tmp_path / "child.yaml", "_base_: base.yaml\ntrain: {lr: 0.02}\nmodel: {head: {type: YOLOv5Head, num_classes: 7}}\n")
    cfg = load_config(child, ["train.epochs=5", "data.paint=true", "model.head.box_gain=0.1"])
    assert cfg["train"]["lr"] == 0.02 and cfg["train"]["epochs"] == 5            # child wins, override wins over both
    assert cfg["data"]["size"] == 320 and cfg["data"]["paint"] is True and cfg["data"]["max_boxes"] == 40   # inherited / defaulted
    assert cfg["model"] == {"type": "YOLOv5", "variant": "n", "head": {"type": "YOLOv5Head", "num_classes": 7, "box_gain": 0.1}}
    # the same content as .json and as .py
    js = _write(tmp_path / "c.json", json.dumps({"_base_": "base.yaml", "train": {"lr": 0.02}}))
    py = _write(tmp_path / "c.py", "_base_ = 'base.yaml'\ntrain = dict(lr=0.02)\n")
    assert load_config(js)["train"] == load_config(py)["train"]
    assert load_config(py)["model"]["variant"] == "n"
    # _delete_ drops inherited keys
    d = _write(tmp_path / "d.yaml", "_base_: base.yaml\nmodel: {_delete_: true, type: FasterRCNN}\n")
    assert load_config(d)["model"] == {"type": "FasterRCNN"}
    # errors are loud and specific
    for text, msg in (("model: {type: Nope}\n", "not a registered detector"), ("train: {lr: 1}\n", "must be a registry config"),
                      ("model: {type: YOLOv5}\ntrain: {learning_rate: 1}\n", "unknown key"),
                      ("model: {type: YOLOv5}\noptimizer: {}\n", "unknown top-level")):
        with pytest.raises(ConfigError, match=msg):
            load_config(_write(tmp_path / "bad.yaml", text))
    with pytest.raises(ConfigError, match="cycle"):
        _write(tmp_path / "x.yaml", "_base_: y.yaml\n")
        _write(tmp_path / "y.yaml", "_base_: x.yaml\n")
        load_config(str(tmp_path / "x.yaml"))
    # the shipped configs resolve and build
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")
    from heltondetection_amd.detectors import build_detector
    y = load_config(os.path.join(root, "yolov5s_synth.yaml"))
    assert y["data"]["size"] == 640 and y["data"]["batch"] == 16 and y["train"]["mosaic"] == 0.5
    f = load_config(os.path.join(root, "frcnn_r50_pafpn_decoupled_synth.yaml"))
    m = build_detector(dict(f["model"], num_classes=80))
    assert m.roi_head.decoupled and m.rpn_head.pre == 2000 and m.rpn_head.post == 1000 and type(m.neck).__name__ == "PAFPN"


def test_run_logger_jsonl_roundtrip(tmp_path, capsys):
    from heltondetection_amd.logger import RunLogger, read_events, read_scalars
    lg = RunLogger(str(tmp_path), rank=0)
    lg.event("start", mode="train", steps=3)
    for s in range(3):
        lg.scalars(s + 1, loss=torch.tensor(1.0 / (s + 1)), lr=1e-3)
    lg.close()
    quiet = RunLogger(str(tmp_path), rank=1)   # other ranks write nothing by default
    quiet.event("start")
    quiet.close()
    path = os.path.join(str(tmp_path), "log.jsonl")
    assert [e["event"] for e in read_events(path)] == ["start", "scalars", "scalars", "scalars"]
    assert list(read_scalars(path, "loss")) == [(1, 1.0), (2, 0.5), (3, pytest.approx(1 / 3))]
    assert not os.path.exists(os.path.join(str(tmp_path), "log.rank1.jsonl"))
    assert "scalars: step=3" in capsys.readouterr().out


def _patch_ops_with_oracle(monkeypatch):
    from heltondetection_amd import ops
    captured = {}

    def fake_assign(shapes, strides, anchors, gt_boxes, gt_cls, m_valid, anchor_t=4.0):
        captured["t"] = (gt_boxes, gt_cls, m_valid, anchor_t)
        return None

    def fake_loss(preds, strides, anchors, slots, M, box_gain, obj_gain, cls_gain, balance, biases=None, focal=None):
        gtb, gtc, mv, at = captured["t"]
        total, lb, lo, lc = tr.yolo_loss(list(preds), strides, anchors, gtb, gtc, mv, at, box_gain, obj_gain, cls_gain, balance)
        return total, torch.stack([total.detach(), lb, lo, lc])

    def fake_detect(preds, strides, anchors, conf_thr=0.25, iou_thr=0.45, max_nms=30000, max_det=300):
        dec = tr.yolo_decode([p.float() for p in preds], strides, anchors)
        return tr.yolo_nms(dec, conf_thr, iou_thr, max_nms, max_det)

    monkeypatch.setattr(ops, "yolo_assign", fake_assign)
    monkeypatch.setattr(ops, "yolo_loss", fake_loss)
    monkeypatch.setattr(ops, "yolo_detect", fake_detect)


@pytest.mark.timeout(600)
def test_cli_train_eval_test_cpu_plumbing(tmp_path, monkeypatch, capsys):
    """BASELINE configs[0] in small: `cli train` -> checkpoint + log, `cli eval` -> results json + mAP, `cli test` ->
    detections json, all from one config file."""
    from heltondetection_amd import cli
    from heltondetection_amd.logger import read_events, read_scalars
    _patch_ops_with_oracle(monkeypatch)
    torch.set_num_threads(4)
    cfg = _write(tmp_path / "tiny.yaml", yaml.safe_dump({
        "model": {"type": "YOLOv5", "variant": "n"},
        "data": {"type": "synthetic_coco", "n_images": 8, "val_images": 4, "size": 64, "num_classes": 3, "max_boxes": 4,
                 "paint": True, "batch": 2},
        "train": {"epochs": 1, "lr": 0.001, "warmup_steps": 1, "amp": "fp32", "hip_graph": False, "log_every": 1},
        "eval": {"conf_thr": 0.001, "max_det": 20, "max_nms": 200},
        "test": {"conf_thr": 0.001, "max_det": 5},
    }))
    work = str(tmp_path / "run")
    # the head's bias gradient normally comes from the HIP loss kernels; on the patched CPU path autograd owns it
    from heltondetection_amd.heads import YOLOv5Head
    monkeypatch.setattr(YOLOv5Head, "bias_grad_in_loss", False, raising=False)
    orig_init = YOLOv5Head.__init__

    def init(self, *a, **k):
        orig_init(self, *a, **k)
        self.bias_grad_in_loss = False
    monkeypatch.setattr(YOLOv5Head, "__init__", init)

    out = cli.main(["train", "--config", cfg, "--work-dir", work, "--device", "cpu"])
    assert out["steps"] == 4 and out["last_loss"] is not None and os.path.exists(out["checkpoint"])
    log = os.path.join(work, "log.jsonl")
    assert [e["event"] for e in read_events(log)][:2] == ["start", "scalars"]
    losses = [v for _, v in read_scalars(log, "loss")]
    assert len(losses) == 4 and all(l == l for l in losses)
    assert yaml.safe_load(open(os.path.join(work, "config.yaml")))["data"]["size"] == 64      # resolved config kept with the run
    assert os.path.exists(os.path.join(work, "data", "train", "annotations.json"))
    # resume continues from the stored step with the optimizer state
    out2 = cli.main(["train", "--config", cfg, "--work-dir", work, "--device", "cpu", "--resume", out["checkpoint"],
                     "--set", "train.epochs=2"])
    assert out2["steps"] == 8
    m = cli.main(["eval", "--config", cfg, "--work-dir", work, "--device", "cpu"])   # picks up <work-dir>/last.pt
    assert 0.0 <= m["mAP50"] <= 1.0 and m["images"] == 4 and os.path.exists(m["results"])
    res = json.load(open(m["results"]))
    assert all({"image_id", "category_id", "bbox", "score"} <= set(r) for r in res)
    t = cli.main(["test", "--config", cfg, "--work-dir", work, "--device", "cpu"])
    dets = json.load(open(t["path"]))
    assert t["images"] == 4 and len(dets) == 4 and all(len(d["detections"]) <= 5 for d in dets)
    # a directory of arbitrary-size images: padded to a multiple of 32, boxes stay in the image frame
    src = tmp_path / "imgs"
    src.mkdir()
    from PIL import Image
    import numpy as np
    Image.fromarray((np.random.RandomState(0).rand(50, 70, 3) * 255).astype("uint8")).save(src / "a.png")
    t2 = cli.main(["test", "--config", cfg, "--work-dir", work, "--device", "cpu", "--source", str(src)])
    assert t2["images"] == 1
    last = capsys.readouterr().out.strip().splitlines()[-1]
    assert json.loads(last)["cmd"] == "test"


def test_cli_rejects_bad_config(tmp_path):
    from heltondetection_amd import cli
    bad = _write(tmp_path / "bad.yaml", "model: {type: YOLOv5}\ntrain: {lr: 1, nope: 2}\n")
    with pytest.raises(SystemExit, match="config error"):
        cli.main(["train", "--config", bad, "--work-dir", str(tmp_path / "w"), "--device", "cpu"])
