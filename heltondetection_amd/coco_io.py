"""COCO-format data either side of the hot path (reference README.md:10 "COCO-format dataset reading",
README.md:61 "evaluation through the pycocotools interface"): a synthetic dataset writer (config 1 of
BASELINE.json: "synthetic COCO-format images + random boxes"), a reader that yields the padded batch tensors
the detectors consume, and the results-json <-> evaluator bridge.  pycocotools is not in the image, so
`evaluate_coco_json` feeds evaluator.MeanAP (COCO's 101-point AP over IoU 0.50:0.95) from the two json files.

COCO conventions kept: boxes are [x, y, w, h] in pixels, `category_id` is an arbitrary positive id listed
under "categories" (mapped to contiguous 0..nc-1 class indices in id order), `iscrowd` annotations are
skipped on reading."""
from __future__ import annotations

import json
import os
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import synthetic as syn
from .evaluator import MeanAP


def write_synthetic_coco(root: str, n_images: int, size: int, num_classes: int = 80, seed: int = 0,
                         style: str = "coco", max_boxes: int = 40, write_pixels: bool = True,
                         paint: bool = False) -> str:
    """Writes `root/images/%06d.png` (uniform-noise RGB, the synthetic images of SURVEY.md §8d) and
    `root/annotations.json`; returns the annotation path.  Deterministic in `seed`.
    paint=True draws every box as a rectangle in its class colour over the noise (later boxes on top), which
    gives a detector something to learn; the default keeps the pure-noise images of the measurement spec."""
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    g = syn.make_generator(seed)
    images, annotations = [], []
    ann_id = 1
    for i in range(n_images):
        name = f"{i:06d}.png"
        noise = syn.make_images(1, size, g)[0] if write_pixels else None
        gtb, gtc, mv = syn.make_targets(1, size, g, style, num_classes, max_boxes)
        if write_pixels:
            from PIL import Image
            if paint:
                for k in range(int(mv[0])):
                    x1, y1, x2, y2 = [int(round(float(v))) for v in gtb[0, k]]
                    c = int(gtc[0, k])
                    colour = torch.tensor([((c * 37) % 97) / 96.0, ((c * 59 + 31) % 89) / 88.0, ((c * 83 + 7) % 101) / 100.0])
                    noise[:, y1:y2, x1:x2] = 0.85 * colour[:, None, None] + 0.15 * noise[:, y1:y2, x1:x2]
            px = (noise.permute(1, 2, 0) * 255.0).round().clamp(0, 255).to(torch.uint8).numpy()
            Image.fromarray(px, "RGB").save(os.path.join(root, "images", name))
        images.append({"id": i + 1, "file_name": name, "width": size, "height": size})
        for k in range(int(mv[0])):
            x1, y1, x2, y2 = [float(v) for v in gtb[0, k]]
            annotations.append({"id": ann_id, "image_id": i + 1, "category_id": int(gtc[0, k]) + 1,
                                "bbox": [x1, y1, x2 - x1, y2 - y1], "area": (x2 - x1) * (y2 - y1), "iscrowd": 0})
            ann_id += 1
    cats = [{"id": c + 1, "name": f"class_{c}"} for c in range(num_classes)]
    path = os.path.join(root, "annotations.json")
    with open(path, "w") as f:
        json.dump({"images": images, "annotations": annotations, "categories": cats}, f)
    return path


class CocoDetection:
    """Minimal COCO-format reader: `len()`, `[i] -> (image [3,H,W] float in [0,1], boxes [n,4] xyxy, cls [n])`,
    and `batches(B)` -> padded (images, gt_boxes [B,M,4], gt_cls [B,M] int32, m_valid [B] int32, image_ids)."""

    def __init__(self, ann_path: str, image_dir: Optional[str] = None, max_boxes: Optional[int] = None):
        with open(ann_path) as f:
            js = json.load(f)
        self.image_dir = image_dir or os.path.join(os.path.dirname(ann_path), "images")
        self.images = sorted(js["images"], key=lambda im: im["id"])
        self.cat_ids = sorted(c["id"] for c in js["categories"])
        self.cat_to_cls = {cid: k for k, cid in enumerate(self.cat_ids)}
        self.num_classes = len(self.cat_ids)
        self._ann: Dict[int, List[dict]] = {im["id"]: [] for im in self.images}
        for a in js["annotations"]:
            if a.get("iscrowd", 0) or a["image_id"] not in self._ann:
                continue
            self._ann[a["image_id"]].append(a)
        self.max_boxes = max_boxes or max([len(v) for v in self._ann.values()] + [1])

    def __len__(self) -> int:
        return len(self.images)

    def targets(self, i: int) -> Tuple[np.ndarray, np.ndarray]:
        anns = self._ann[self.images[i]["id"]]
        b = np.array([[a["bbox"][0], a["bbox"][1], a["bbox"][0] + a["bbox"][2], a["bbox"][1] + a["bbox"][3]]
                      for a in anns], np.float32).reshape(-1, 4)
        c = np.array([self.cat_to_cls[a["category_id"]] for a in anns], np.int32)
        return b, c

    def __getitem__(self, i: int):
        from PIL import Image
        im = Image.open(os.path.join(self.image_dir, self.images[i]["file_name"])).convert("RGB")
        px = torch.from_numpy(np.asarray(im, np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        b, c = self.targets(i)
        return px, torch.from_numpy(b), torch.from_numpy(c)

    def batches(self, batch: int, drop_last: bool = False) -> Iterator[tuple]:
        M = self.max_boxes
        for s in range(0, len(self), batch):
            idx = list(range(s, min(s + batch, len(self))))
            if drop_last and len(idx) < batch:
                return
            items = [self[i] for i in idx]
            imgs = torch.stack([it[0] for it in items])
            gb = torch.zeros((len(idx), M, 4))
            gc = torch.zeros((len(idx), M), dtype=torch.int32)
            mv = torch.zeros((len(idx),), dtype=torch.int32)
            for k, (_, b, c) in enumerate(items):
                n = min(b.shape[0], M)
                gb[k, :n], gc[k, :n], mv[k] = b[:n], c[:n], n
            yield imgs, gb, gc, mv, [self.images[i]["id"] for i in idx]


def detections_to_coco(det, det_count, image_ids: Sequence[int], cat_ids: Sequence[int]) -> List[dict]:
    """Padded detector output ([B,K,6] = x1 y1 x2 y2 score cls, counts [B]) -> COCO results entries."""
    det = det.detach().float().cpu().numpy() if hasattr(det, "detach") else np.asarray(det)
    cnt = det_count.detach().cpu().numpy() if hasattr(det_count, "detach") else np.asarray(det_count)
    out = []
    for b, img_id in enumerate(image_ids):
        for r in det[b, : int(cnt[b])]:
            out.append({"image_id": int(img_id), "category_id": int(cat_ids[int(r[5])]),
                        "bbox": [float(r[0]), float(r[1]), float(r[2] - r[0]), float(r[3] - r[1])], "score": float(r[4])})
    return out


def evaluate_coco_json(ann_path: str, results: Sequence[dict] | str) -> Dict[str, object]:
    """mAP50 / mAP50-95 of COCO-format results (a list or a json path) against COCO-format annotations."""
    ds = CocoDetection(ann_path)
    if isinstance(results, str):
        with open(results) as f:
            results = json.load(f)
    per_img: Dict[int, List[List[float]]] = {im["id"]: [] for im in ds.images}
    for r in results:
        if r["image_id"] in per_img:
            x, y, w, h = r["bbox"]
            per_img[r["image_id"]].append([x, y, x + w, y + h, r["score"], ds.cat_to_cls[r["category_id"]]])
    ev = MeanAP(ds.num_classes)
    for i, im in enumerate(ds.images):
        b, c = ds.targets(i)
        ev.add_image(np.asarray(per_img[im["id"]], np.float32).reshape(-1, 6), b, c)
    return ev.compute()
