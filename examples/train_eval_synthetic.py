"""End-to-end template on synthetic data: COCO-format dataset on disk -> reader -> hipGraph-replayed training with
device-side Mosaic -> hipGraph-replayed inference -> COCO results json -> mAP50 / mAP50-95.

    python examples/train_eval_synthetic.py --workdir /tmp/hd_demo --images 32 --size 160 --epochs 30

Everything model-side is selected by registry configs (dicts), as the reference selects components by config
file (README.md:12); swap `MODEL` for e.g. dict(type="FasterRCNN", num_classes=...)."""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from heltondetection_amd import coco_io                      # noqa: E402
from heltondetection_amd.augment import Mosaic               # noqa: E402
from heltondetection_amd.detectors import build_detector     # noqa: E402
from heltondetection_amd.profiling import model_summary      # noqa: E402
from heltondetection_amd.serve import GraphedDetector        # noqa: E402
from heltondetection_amd.train import Trainer                # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--workdir", default="/tmp/hd_demo")
    ap.add_argument("--images", type=int, default=32)
    ap.add_argument("--size", type=int, default=160)
    ap.add_argument("--classes", type=int, default=4)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--mosaic", type=float, default=0.5)
    ap.add_argument("--lr", type=float, default=2e-3)
    ap.add_argument("--amp", choices=("bf16", "fp32", "torch_amp"), default="bf16",
                    help="bf16: the product step (bf16 autocast, weight shadows, hand-written kernels); fp32: the same step without "
                         "autocast; torch_amp: plain torch modules under bf16 autocast (tools/convergence_ab.py compares them)")
    ap.add_argument("--seed", type=int, default=0, help="weight initialisation + the step's random draws (Mosaic)")
    args = ap.parse_args(argv)
    dev = torch.device("cuda", 0)
    torch.manual_seed(args.seed)
    if args.amp == "torch_amp":
        from heltondetection_amd.layers import ConvBNAct
        ConvBNAct.fused = False

    # paint=True: boxes are drawn as class-coloured rectangles, so there is something to learn
    ann = coco_io.write_synthetic_coco(args.workdir, args.images, args.size, args.classes, seed=0, max_boxes=6, paint=True)
    ds = coco_io.CocoDetection(ann)
    MODEL = dict(type="YOLOv5", variant="s", num_classes=ds.num_classes, img_size=args.size)
    model = build_detector(MODEL).to(dev).to(memory_format=torch.channels_last).train()
    batches = [tuple(t.to(dev) if torch.is_tensor(t) else t for t in b) for b in ds.batches(args.batch, drop_last=True)]
    batches = [(b[0].contiguous(memory_format=torch.channels_last),) + b[1:] for b in batches]
    steps = args.epochs * len(batches)
    trainer = Trainer(model, lr=args.lr, total_steps=steps, warmup_steps=max(steps // 20, 1), device=dev, use_graph=True,
                      amp_dtype=None if args.amp == "fp32" else torch.bfloat16, seed=args.seed,
                      augment=Mosaic(p=args.mosaic, max_boxes=2 * ds.max_boxes) if args.mosaic > 0 else None)
    first = last = None
    for ep in range(args.epochs):
        for imgs, gb, gc, mv, _ in batches:
            loss, parts = trainer.step(imgs, gb, gc, mv)
        last = float(loss)
        first = last if first is None else first
    summary = model_summary(model.eval(), lambda: model.extract(batches[0][0][:1]))
    detector = GraphedDetector(model, batches[0][0], conf_thr=0.001, iou_thr=0.6, max_nms=3000, max_det=100)
    results = []
    for imgs, _, _, _, ids in batches:
        det, cnt = detector(imgs)
        results += coco_io.detections_to_coco(det, cnt, ids, ds.cat_ids)
    res_path = os.path.join(args.workdir, "results.json")
    with open(res_path, "w") as f:
        json.dump(results, f)
    metrics = coco_io.evaluate_coco_json(ann, res_path)
    out = {"amp": args.amp, "seed": args.seed, "first_loss": first, "last_loss": last, "steps": steps, "hip_graph": trainer._graph is not None,
           "mAP50": metrics["mAP50"], "mAP50_95": metrics["mAP50_95"], "detections": len(results), **summary}
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
