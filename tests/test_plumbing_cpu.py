"""BASELINE.json configs[0]: YOLOv5s 640x640 bs=2 on PyTorch-CPU, synthetic images + random boxes —
plumbing only (registry -> backbone -> neck -> head -> loss), no GPU.  The product has no CPU ops,
so the test monkeypatches the two op calls the head makes with the oracle (test infrastructure)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as tr


def test_registry_builds_all_detectors():
    from heltondetection_amd.detectors import build_detector
    from heltondetection_amd.registry import ASSIGNERS, BACKBONES, DETECTORS, HEADS, NECKS
    assert {"YOLOv5", "FasterRCNN"} <= set(DETECTORS.names())
    assert {"CSPDarknet", "ResNet50"} <= set(BACKBONES.names())
    assert {"YOLOv5PAFPN", "FPN", "PAFPN"} <= set(NECKS.names())
    assert {"YOLOv5Head", "RPNHead", "RoIHead"} <= set(HEADS.names())
    assert {"YOLOv5Assigner", "MaxIoUAssigner"} <= set(ASSIGNERS.names())
    m = build_detector(dict(type="YOLOv5", variant="s"))
    assert abs(sum(p.numel() for p in m.parameters()) / 1e6 - 7.235) < 0.01   # nominal YOLOv5s size
    with pytest.raises(KeyError):
        build_detector(dict(type="NoSuchDetector"))

    from heltondetection_amd.registry import NECKS as N

    @N.register("IdentityNeck")
    class IdentityNeck(torch.nn.Module):  # user-defined component, selectable by name
        def __init__(self, in_channels, depth=0.33):
            super().__init__()
            self.out_channels = tuple(in_channels)

        def forward(self, feats):
            return feats

    m2 = build_detector(dict(type="YOLOv5", variant="n", neck=dict(type="IdentityNeck")))
    assert isinstance(m2.neck, IdentityNeck)


@pytest.mark.timeout(600)
def test_yolov5s_640_bs2_cpu_plumbing(monkeypatch):
    from heltondetection_amd import ops, synthetic as syn
    from heltondetection_amd.detectors import build_detector

    captured = {}

    def fake_assign(shapes, strides, anchors, gt_boxes, gt_cls, m_valid, anchor_t=4.0):
        captured["targets"] = (gt_boxes, gt_cls, m_valid, anchor_t)
        return None

    def fake_loss(preds, strides, anchors, slots, M, box_gain, obj_gain, cls_gain, balance, biases=None, focal=None):
        gtb, gtc, mv, at = captured["targets"]
        ft = None if focal is None else (focal.get("gamma", 1.5), focal.get("alpha", 0.25), focal.get("terms", ("cls", "obj")))
        total, lb, lo, lc = tr.yolo_loss(list(preds), strides, anchors, gtb, gtc, mv, at, box_gain, obj_gain, cls_gain, balance,
                                         focal=ft)
        return total, torch.stack([total.detach(), lb, lo, lc])

    monkeypatch.setattr(ops, "yolo_assign", fake_assign)
    monkeypatch.setattr(ops, "yolo_loss", fake_loss)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    model = build_detector(dict(type="YOLOv5", variant="s", num_classes=80, img_size=640))
    model = model.to(memory_format=torch.channels_last).train()
    model.head.bias_grad_in_loss = False  # CPU plumbing: autograd owns the bias gradient
    g = syn.make_generator(0)
    img = syn.make_images(2, 640, g).contiguous(memory_format=torch.channels_last)
    gtb, gtc, mv = syn.make_targets(2, 640, g, "coco", 80, 40)
    loss, parts = model.forward_train(img, gtb, gtc, mv)
    assert torch.isfinite(loss) and parts.shape == (4,)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
    preds = model.head(model.extract(img))
    assert [tuple(p.shape) for p in preds] == [(2, 3, 80, 80, 85), (2, 3, 40, 40, 85), (2, 3, 20, 20, 85)]
    assert all(p.stride(4) == 1 for p in preds)  # channels-last view, no copy


def test_drop_block_statistics_and_eval_identity():
    from heltondetection_amd.layers import DropBlock2d
    torch.manual_seed(0)
    db = DropBlock2d(p=0.3, block_size=5).train()
    x = torch.ones((4, 16, 40, 40))
    y = db(x)
    dropped = float((y == 0).float().mean())
    assert 0.2 < dropped < 0.4                       # expected dropped fraction ~ p
    assert abs(float(y.mean()) - 1.0) < 1e-3         # survivors rescaled: the mean is preserved
    # zeros come in block_size squares: the dropped set is a union of (border-clipped) 5x5 windows, i.e. it is
    # unchanged by a morphological opening with that window
    d = (y == 0).float()
    eroded = 1.0 - torch.nn.functional.max_pool2d(1.0 - d, 5, 1, 2)
    assert torch.equal(torch.nn.functional.max_pool2d(eroded, 5, 1, 2), d)
    assert torch.equal(db.eval()(x), x)
    with pytest.raises(ValueError):
        DropBlock2d(p=0.1, block_size=4)


def test_frozen_backbone_and_drop_block_configs():
    """ "YOLOv5s_dropBlock0.5_frozeBackbone" (README.md:133): no gradient and no statistics update in the backbone."""
    from heltondetection_amd.detectors import build_detector
    from heltondetection_amd.profiling import count_flops, count_params, model_summary
    torch.manual_seed(0)
    model = build_detector(dict(type="YOLOv5", variant="n", num_classes=4, img_size=64,
                                backbone=dict(type="CSPDarknet", variant="n", frozen=True),
                                neck=dict(type="YOLOv5PAFPN", depth=0.33, drop_block=dict(p=0.5, block_size=3))))
    model.train()
    assert not any(p.requires_grad for p in model.backbone.parameters())
    assert all(not m.training for m in model.backbone.modules())
    assert model.neck.training and model.neck.drop.training
    rm0 = [m.running_mean.clone() for m in model.backbone.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    x = torch.rand((2, 3, 64, 64))
    feats = model.extract(x)
    sum(f.float().mean() for f in feats).backward()
    assert all(p.grad is None for p in model.backbone.parameters())
    assert any(p.grad is not None for p in model.neck.parameters())
    rm1 = [m.running_mean for m in model.backbone.modules() if isinstance(m, torch.nn.BatchNorm2d)]
    assert all(torch.equal(a, b) for a, b in zip(rm0, rm1))
    assert any(bool((f == 0).any()) for f in feats)                 # DropBlock acted on the neck outputs
    # FLOPs / Params (README.md:24): a 3x3 conv 8->16 on 10x10 = 2*16*10*10*8*9 FLOPs
    conv = torch.nn.Conv2d(8, 16, 3, padding=1, bias=False)
    flops, _ = count_flops(conv, lambda: conv(torch.zeros(1, 8, 10, 10)))
    assert flops == 2 * 16 * 10 * 10 * 8 * 9 and count_params(conv) == 16 * 8 * 9
    s = model_summary(model, lambda: model.extract(x))
    assert s["params_M"] > 1.0 and s["gflops"] > 0.01
    assert count_params(model, trainable_only=True) < count_params(model)


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """`bench.py --gpus N` must never print a line for a different number of ranks.  With a launcher
    in the environment that disagrees it exits non-zero with the launch line; with no launcher and too few devices it
    exits non-zero too (on a multi-GPU node it starts the N ranks itself) — all before the first GPU call."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr and "--nproc-per-node 2" in r.stderr and r.stdout.strip() == ""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and r.stdout.strip() == "" and "--gpus 8" in r.stderr


def test_shipped_find_db_is_selected_only_when_the_caller_chose_nothing(monkeypatch, tmp_path):
    """tuning.use_shipped_find_db hands MIOpen a PRIVATE COPY of heltondetection_amd/miopen_db (MIOpen appends to its user
    db: the tracked directory itself is never written, and two processes never share one copy) unless MIOPEN_USER_DB_PATH is
    already set or the switch HD_NO_SHIPPED_FIND_DB=1 is; the shipped directory holds text find-db / perf-db files for gfx950."""
    import os
    from heltondetection_amd import tuning
    shipped = sorted(f for f in os.listdir(tuning._DB) if f.endswith(".txt"))
    assert os.path.isdir(tuning._DB) and any(f.endswith(".ufdb.txt") and f.startswith("gfx950") for f in shipped)
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("HD_NO_SHIPPED_FIND_DB", raising=False)
    monkeypatch.delenv("HD_FIND_DB_DIR", raising=False)
    monkeypatch.setattr(tuning, "_private", None)
    assert tuning.use_shipped_find_db() is True
    first = os.environ["MIOPEN_USER_DB_PATH"]
    assert first != tuning._DB and first == tuning.find_db_dir() and sorted(os.listdir(first)) == shipped
    assert tuning.use_shipped_find_db() is True and os.environ["MIOPEN_USER_DB_PATH"] == first   # idempotent in one process
    # another process (simulated): its own directory
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.setattr(tuning, "_private", None)
    assert tuning.use_shipped_find_db() is True and os.environ["MIOPEN_USER_DB_PATH"] != first
    # a persistent private directory, one per rank under a launcher
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.setattr(tuning, "_private", None)
    monkeypatch.setenv("HD_FIND_DB_DIR", str(tmp_path / "db"))
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert tuning.use_shipped_find_db() is True and os.environ["MIOPEN_USER_DB_PATH"] == str(tmp_path / "db" / "rank1")
    assert sorted(os.listdir(tmp_path / "db" / "rank1")) == shipped
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
    monkeypatch.setattr(tuning, "_private", None)
    assert tuning.use_shipped_find_db() is False and os.environ["MIOPEN_USER_DB_PATH"] == "/somewhere/else"
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.setenv("HD_NO_SHIPPED_FIND_DB", "1")
    assert tuning.use_shipped_find_db() is False and "MIOPEN_USER_DB_PATH" not in os.environ


def test_bucket_plan_overlaps_most_of_the_gradient_bytes_and_keeps_the_exposed_tail_small():
    """with 32 MB buckets YOLOv5s (27.6 MB of float32 gradients) had two buckets, 4.4 and 23.2 MB, and the big
    one held the stem, i.e. 84 % of the bytes were reduced with nothing left to overlap.  dist.plan_buckets sizes buckets from
    the model: a partition in arrival order, >= 6 buckets for YOLOv5s, the LAST bucket (complete only when backward ends)
    <= 2 MB, and no bucket but an oversized single parameter above twice the target."""
    import torch
    from heltondetection_amd.detectors import build_detector
    from heltondetection_amd.dist import GradReducer, plan_buckets
    for variant, min_buckets in (("s", 6), ("l", 8)):
        model = build_detector(dict(type="YOLOv5", variant=variant, num_classes=80, img_size=640))
        params = [p for p in model.parameters() if p.requires_grad]
        sizes = [p.numel() * 4 for p in reversed(params)]
        plan = plan_buckets(sizes)
        assert [i for b in plan for i in b] == list(range(len(sizes)))           # a partition, in arrival order
        nbytes = [sum(sizes[i] for i in b) for b in plan]
        total, target = sum(sizes), max(4 << 20, sum(sizes) // 16)
        assert len(plan) >= min_buckets, nbytes
        assert nbytes[-1] <= 2 << 20 and nbytes[-1] <= 0.05 * total, nbytes       # the exposed tail
        assert max(nbytes) < 2 * target + max(sizes), nbytes
        red = GradReducer(model)                                                  # the shipped default takes this plan
        assert [len(b.params) for b in red.buckets] == [len(b) for b in plan]
        assert red.buckets[-1].params[-1] is params[0]                            # the stem's weight is in the last bucket
    # degenerate inputs
    assert plan_buckets([]) == [] and plan_buckets([10]) == [[0]]
    assert plan_buckets([4 << 20, 4 << 20, 100]) == [[0], [1], [2]]
    assert plan_buckets([1, 2, 3], 2, 2, 3) == [[0, 1], [2]]
    # explicit sizes still work (tests, configs): 1 MB buckets over 10 x 0.5 MB
    plan = plan_buckets([1 << 19] * 10, 1 << 20, 1 << 19, 1 << 19)
    assert plan == [[0], [1, 2], [3, 4], [5, 6], [7, 8], [9]]


def _load_bench():
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("hd_bench_under_test", os.path.join(root, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, root


def test_bench_stdout_line_is_compact_and_round_trips():
    """Round 4's driver record was lost because the one JSON line had grown to 35 KB and the driver keeps a ~14 KB tail.
    bench.compact_line must stay under 4096 bytes whatever the detail record holds, parse back, and keep every key the bench
    contract names plus `roofline` and `cpu_baseline`.  Fed with the committed full record of round 4 and with a bloated one."""
    import json
    import os
    bench, root = _load_bench()
    full = json.loads(open(os.path.join(root, "profiles", "r04_bench_default.log")).read().strip().splitlines()[-1])
    assert len(json.dumps(full)) > 30000
    fat = json.loads(json.dumps(full))
    fat["config"]["workload"] = "w" * 5000
    fat["roofline"]["kernel"] = "k" * 5000
    fat["cpu_baseline"]["sample"] = "s" * 5000
    fat["roofline_secondary"] = fat["roofline_secondary"] * 40
    fat["frcnn_832"] = {"error": "e" * 3000}
    for rec in (full, fat):
        line = bench.compact_line(rec, "bench_detail.json")
        assert "\n" not in line and len(line.encode()) < bench.COMPACT_LIMIT <= 4096
        back = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in back, k
        assert back["value"] == rec["value"] and back["ms_per_step"] == rec["ms_per_step"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert back["roofline"][k] == rec["roofline"][k]
        assert back["roofline"]["frac"] == round(back["roofline"]["achieved"] / back["roofline"]["peak"], 4)
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in back["cpu_baseline"]
        assert "workload" in back["config"]
    legs = json.loads(bench.compact_line(full))["legs"]
    assert set(legs) == set(bench.LEG_NAMES) and all("ms_per_step" in v for v in legs.values())
