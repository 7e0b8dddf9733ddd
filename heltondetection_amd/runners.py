"""Train / eval / test runners driven by a config file (reference README.md:12-14: "one-line CLI to train, evaluate
and test; hyper-parameters come from the config; logger + tensorboard record the run").

    train     data-parallel training loop around train.Trainer (hipGraph step, bf16 autocast, warmup + cosine LR,
              device-side Mosaic), sharded by image across ranks, periodic evaluation and checkpoints.
    evaluate  images sharded across ranks ("eval shards images across ranks and gathers detections
              once at the end"), COCO results json + mAP50 / mAP50-95 through evaluator.MeanAP.
    test      inference on a directory of images (or the val split) -> detections json; no ground truth needed.

The runners own no kernels: everything device-side goes through the detectors and heltondetection_amd.ops.
"""
from __future__ import annotations

import json
import math
import os
import time
from typing import Any, Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import coco_io
from .config import dump as dump_config
from .detectors import build_detector
from .dist import init_distributed
from .logger import RunLogger

_AMP = {"bf16": torch.bfloat16, "fp32": None, None: None}


# ------------------------------------------------------------------------------------------- set-up helpers
def _device(local: int, want: Optional[str]) -> torch.device:
    if want:
        return torch.device(want)
    return torch.device("cuda", local) if torch.cuda.is_available() else torch.device("cpu")


def _barrier(world: int) -> None:
    if world > 1 and dist.is_initialized():
        dist.barrier()


def prepare_data(cfg: Dict[str, Any], work_dir: str, rank: int, world: int) -> Tuple[str, str]:
    """-> (train annotation path, val annotation path).  `synthetic_coco` data sets are written once by rank 0
    (COCO-format json + png images, the "synthetic COCO-format images + random boxes" of BASELINE configs[0])."""
    d = cfg["data"]
    if d["type"] == "coco":
        return d["train_ann"], d.get("val_ann", d["train_ann"])
    root = d.get("root") or os.path.join(work_dir, "data")
    tr, va = os.path.join(root, "train"), os.path.join(root, "val")
    if rank == 0:
        for path, n, seed in ((tr, d["n_images"], d["seed"]), (va, d["val_images"], d["seed"] + 1)):
            if not os.path.exists(os.path.join(path, "annotations.json")):
                coco_io.write_synthetic_coco(path, int(n), int(d["size"]), int(d["num_classes"]), seed=int(seed),
                                             style=d["style"], max_boxes=int(d["max_boxes"]), paint=bool(d["paint"]))
    _barrier(world)
    return os.path.join(tr, "annotations.json"), os.path.join(va, "annotations.json")


def iter_batches(ds: coco_io.CocoDetection, indices: Sequence[int], batch: int, max_boxes: int,
                 drop_last: bool) -> Iterator[tuple]:
    """Padded (images, gt_boxes [B,M,4], gt_cls [B,M], m_valid [B], image ids) over `indices` in order."""
    for s in range(0, len(indices), batch):
        idx = list(indices[s:s + batch])
        if len(idx) < batch and drop_last:
            return
        items = [ds[i] for i in idx]
        imgs = torch.stack([it[0] for it in items])
        gb = torch.zeros((len(idx), max_boxes, 4))
        gc = torch.zeros((len(idx), max_boxes), dtype=torch.int32)
        mv = torch.zeros((len(idx),), dtype=torch.int32)
        for k, (_, b, c) in enumerate(items):
            n = min(b.shape[0], max_boxes)
            gb[k, :n], gc[k, :n], mv[k] = b[:n], c[:n], n
        yield imgs, gb, gc, mv, [ds.images[i]["id"] for i in idx]


def _to_device(batch: tuple, device: torch.device) -> tuple:
    imgs, gb, gc, mv, ids = batch
    imgs = imgs.to(device).contiguous(memory_format=torch.channels_last)
    return imgs, gb.to(device), gc.to(device), mv.to(device), ids


def build_model(cfg: Dict[str, Any], num_classes: int, device: torch.device) -> torch.nn.Module:
    mcfg = dict(cfg["model"])
    mcfg.setdefault("num_classes", num_classes)
    if mcfg["type"] == "YOLOv5":
        mcfg.setdefault("img_size", int(cfg["data"]["size"]))
    model = build_detector(mcfg).to(device)
    return model.to(memory_format=torch.channels_last)


def save_checkpoint(path: str, model: torch.nn.Module, trainer=None, **meta) -> None:
    state = {"model": model.state_dict(), "meta": meta}
    if trainer is not None:
        state["optimizer"] = trainer.opt.state_dict()
        state["scheduler"] = trainer.sched.state_dict()
    tmp = path + ".tmp"
    torch.save(state, tmp)
    os.replace(tmp, path)


def load_checkpoint(path: str, model: torch.nn.Module, trainer=None) -> Dict[str, Any]:
    state = torch.load(path, map_location="cpu")
    model.load_state_dict(state["model"], strict=True)
    if trainer is not None:
        trainer.weights_changed()  # the bf16 shadows of the convolution weights follow the loaded masters
    if trainer is not None and "optimizer" in state:
        trainer.opt.load_state_dict(state["optimizer"])
        trainer.sched.load_state_dict(state["scheduler"])
    return state.get("meta", {})


# ------------------------------------------------------------------------------------------- train
def train(cfg: Dict[str, Any], work_dir: str, device: Optional[str] = None, resume: Optional[str] = None) -> Dict[str, Any]:
    from .train import Trainer
    rank, local, world = init_distributed()
    dev = _device(local, device)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    t = cfg["train"]
    torch.manual_seed(int(t["seed"]))  # identical init on every rank (GradReducer also broadcasts rank 0's parameters)
    torch.backends.cudnn.benchmark = bool(t["miopen_find"])
    if t["miopen_find"]:
        from . import tuning
        tuning.use_shipped_find_db()  # the library convolutions' solver search of one MI355X run (tuning.py)
    log = RunLogger(work_dir, rank, **cfg["log"])
    if rank == 0:
        os.makedirs(work_dir, exist_ok=True)
        with open(os.path.join(work_dir, "config.yaml"), "w") as f:
            f.write(dump_config(cfg))
    train_ann, val_ann = prepare_data(cfg, work_dir, rank, world)
    ds = coco_io.CocoDetection(train_ann)
    batch = int(cfg["data"]["batch"])
    model = build_model(cfg, ds.num_classes, dev).train()
    steps_per_epoch = len(ds) // (batch * world)
    if steps_per_epoch < 1:
        raise ValueError(f"{len(ds)} training images < one global batch ({batch} x {world} ranks)")
    total = steps_per_epoch * int(t["epochs"])
    if t["max_steps"]:
        total = min(total, int(t["max_steps"]))
    augment = None
    if float(t["mosaic"]) > 0 and dev.type == "cuda":
        from .augment import Mosaic
        augment = Mosaic(p=float(t["mosaic"]), max_boxes=2 * ds.max_boxes)
    trainer = Trainer(model, lr=float(t["lr"]), weight_decay=float(t["weight_decay"]), amp_dtype=_AMP[t["amp"]],
                      total_steps=total, warmup_steps=int(t["warmup_steps"]), bucket_bytes=(int(t["bucket_mb"]) << 20) if t["bucket_mb"] else None,
                      device=dev, use_graph=bool(t["hip_graph"]), augment=augment, seed=int(t["seed"]))
    step0 = 0
    if resume:
        meta = load_checkpoint(resume, model, trainer)
        step0 = int(meta.get("step", 0))
        log.event("resume", checkpoint=resume, step=step0)
    log.event("start", mode="train", model=cfg["model"]["type"], params_M=round(sum(p.numel() for p in model.parameters()) / 1e6, 3),
              device=str(dev), world=world, images=len(ds), batch_per_rank=batch, steps=total, hip_graph=trainer.use_graph)
    step, first, last, metrics = step0, None, None, None
    t_win, n_win = time.perf_counter(), 0
    for epoch in range(step0 // steps_per_epoch, math.ceil(total / steps_per_epoch)):
        g = torch.Generator().manual_seed(int(t["seed"]) * 1000003 + epoch)  # same permutation on every rank
        perm = torch.randperm(len(ds), generator=g).tolist()
        mine = [perm[k * world + rank] for k in range(steps_per_epoch * batch)]  # rank r takes every world-th image
        if epoch == step0 // steps_per_epoch and step0 % steps_per_epoch:
            # resumed in the middle of this epoch (a checkpoint written at max_steps): its first batches were already trained on
            mine = mine[(step0 % steps_per_epoch) * batch:]
        model.train()
        for b in iter_batches(ds, mine, batch, ds.max_boxes, drop_last=True):
            if step >= total:
                break
            imgs, gb, gc, mv, _ = _to_device(b, dev)
            loss, parts = trainer.step(imgs, gb, gc, mv)
            step += 1
            n_win += 1
            if step % int(t["log_every"]) == 0 or step == total:
                lv = float(loss)  # the one host sync of the window
                first = lv if first is None else first
                last = lv
                dt = time.perf_counter() - t_win
                log.scalars(step, loss=lv, lr=trainer.opt.param_groups[0]["lr"], images_per_s=round(n_win * batch * world / dt, 2),
                            epoch=epoch, parts=[round(float(v), 5) for v in parts.detach().float().cpu().tolist()])
                t_win, n_win = time.perf_counter(), 0
        done = epoch + 1
        if rank == 0 and t["save_every"] and done % int(t["save_every"]) == 0:
            save_checkpoint(os.path.join(work_dir, "last.pt"), model, trainer, step=step, epoch=done)
        if t["eval_every"] and done % int(t["eval_every"]) == 0:
            metrics = evaluate(cfg, work_dir, model=model, _ctx=(rank, world, dev, log), ann=val_ann, tag=f"epoch{done}")
        if step >= total:
            break
    if rank == 0:
        save_checkpoint(os.path.join(work_dir, "last.pt"), model, trainer, step=step, epoch=math.ceil(step / steps_per_epoch))
    out = {"steps": step, "first_loss": first, "last_loss": last, "hip_graph": trainer._graph is not None,
           "hip_graph_error": trainer.graph_error, "checkpoint": os.path.join(work_dir, "last.pt"), "world": world,
           "metrics": metrics}
    log.event("done", **{k: v for k, v in out.items() if k != "metrics"})
    if rank == 0:
        with open(os.path.join(work_dir, "train_summary.json"), "w") as f:
            json.dump(out, f)
    log.close()
    _barrier(world)
    return out


# ------------------------------------------------------------------------------------------- eval / test
@torch.no_grad()
def _detect(model, imgs, section: Dict[str, Any], amp) -> Tuple[torch.Tensor, torch.Tensor]:
    use_amp = amp is not None and imgs.is_cuda
    with torch.autocast(device_type=imgs.device.type, dtype=amp or torch.bfloat16, enabled=use_amp):
        if hasattr(model, "head"):  # single-stage: thresholds are arguments
            return model.forward_test(imgs, conf_thr=float(section["conf_thr"]), iou_thr=float(section["iou_thr"]),
                                      max_nms=int(section["max_nms"]), max_det=int(section["max_det"]))
        rh = model.roi_head       # two-stage: thresholds live on the RoI head
        old = (rh.score_thr, rh.nms_thr, rh.max_det)
        rh.score_thr, rh.nms_thr, rh.max_det = float(section["conf_thr"]), float(section["iou_thr"]), int(section["max_det"])
        try:
            return model.forward_test(imgs)
        finally:
            rh.score_thr, rh.nms_thr, rh.max_det = old


def _gather_results(local: List[dict], world: int) -> List[dict]:
    if world <= 1 or not dist.is_initialized():
        return local
    parts: List[Optional[List[dict]]] = [None] * world
    dist.all_gather_object(parts, local)  # ONE exchange at the end
    return [r for p in parts for r in (p or [])]


def evaluate(cfg: Dict[str, Any], work_dir: str, ckpt: Optional[str] = None, device: Optional[str] = None, model=None,
             ann: Optional[str] = None, tag: str = "eval", _ctx=None) -> Dict[str, Any]:
    if _ctx is None:
        rank, local, world = init_distributed()
        dev = _device(local, device)
        if dev.type == "cuda":
            torch.cuda.set_device(dev)
        log = RunLogger(work_dir, rank, **cfg["log"])
    else:
        rank, world, dev, log = _ctx
    if ann is None:
        ann = prepare_data(cfg, work_dir, rank, world)[1]
    ds = coco_io.CocoDetection(ann)
    if model is None:
        model = build_model(cfg, ds.num_classes, dev)
        if ckpt:
            load_checkpoint(ckpt, model)
    was_training = model.training
    model.eval()
    e = cfg["eval"]
    batch = int(e["batch"] or cfg["data"]["batch"])
    mine = list(range(rank, len(ds), world))  # image i belongs to rank i % world
    results: List[dict] = []
    t0 = time.perf_counter()
    for b in iter_batches(ds, mine, batch, ds.max_boxes, drop_last=False):
        imgs, _, _, _, ids = _to_device(b, dev)
        det, cnt = _detect(model, imgs, e, _AMP[e["amp"]])
        results += coco_io.detections_to_coco(det, cnt, ids, ds.cat_ids)
    results = _gather_results(results, world)
    metrics: Dict[str, Any] = {}
    if rank == 0:
        os.makedirs(work_dir, exist_ok=True)
        res_path = os.path.join(work_dir, f"results_{tag}.json")
        with open(res_path, "w") as f:
            json.dump(results, f)
        m = coco_io.evaluate_coco_json(ann, results)
        metrics = {"mAP50": float(m["mAP50"]), "mAP50_95": float(m["mAP50_95"]), "detections": len(results), "images": len(ds),
                   "seconds": round(time.perf_counter() - t0, 2), "results": res_path}
        with open(os.path.join(work_dir, f"metrics_{tag}.json"), "w") as f:
            json.dump(metrics, f)
        log.event("eval", tag=tag, **{k: v for k, v in metrics.items() if k != "results"})
    if world > 1 and dist.is_initialized():
        box = [metrics]
        dist.broadcast_object_list(box, src=0)
        metrics = box[0]
    model.train(was_training)
    if _ctx is None:
        log.close()
    return metrics


def _load_image(path: str) -> torch.Tensor:
    from PIL import Image
    im = Image.open(path).convert("RGB")
    return torch.from_numpy(np.asarray(im, np.uint8).copy()).permute(2, 0, 1).float() / 255.0


def test(cfg: Dict[str, Any], work_dir: str, ckpt: Optional[str] = None, device: Optional[str] = None,
         source: Optional[str] = None) -> Dict[str, Any]:
    """Inference without ground truth.  `source`: a directory of images (png / jpg; each is padded at the bottom and
    right with 114/255 to a multiple of 32, so box coordinates stay in the image's own pixel frame), default: the val
    split of the config.  Writes `<work_dir>/detections.json`: [{"file": ..., "detections": [[x1,y1,x2,y2,score,cls]]}]."""
    rank, local, world = init_distributed()
    dev = _device(local, device)
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    log = RunLogger(work_dir, rank, **cfg["log"])
    tcfg = cfg["test"]
    if source is None:
        ann = prepare_data(cfg, work_dir, rank, world)[1]
        ds = coco_io.CocoDetection(ann)
        files = [os.path.join(ds.image_dir, im["file_name"]) for im in ds.images]
        nc = ds.num_classes
    else:
        files = sorted(os.path.join(source, f) for f in os.listdir(source) if f.lower().endswith((".png", ".jpg", ".jpeg", ".bmp")))
        nc = int(cfg["model"].get("num_classes", cfg["data"]["num_classes"]))
    model = build_model(cfg, nc, dev).eval()
    if ckpt:
        load_checkpoint(ckpt, model)
    batch = int(tcfg["batch"] or cfg["data"]["batch"])
    mine = files[rank::world]
    out: List[dict] = []
    for s in range(0, len(mine), batch):
        chunk = mine[s:s + batch]
        ims = [_load_image(p) for p in chunk]
        H = max(int(math.ceil(i.shape[1] / 32) * 32) for i in ims)
        W = max(int(math.ceil(i.shape[2] / 32) * 32) for i in ims)
        x = torch.full((len(ims), 3, H, W), 114.0 / 255.0)
        for k, i in enumerate(ims):
            x[k, :, : i.shape[1], : i.shape[2]] = i
        det, cnt = _detect(model, x.to(dev).contiguous(memory_format=torch.channels_last), tcfg, _AMP[tcfg["amp"]])
        det, cnt = det.float().cpu(), cnt.cpu()
        for k, p in enumerate(chunk):
            out.append({"file": os.path.basename(p), "detections": [[round(float(v), 3) for v in r] for r in det[k, : int(cnt[k])]]})
    out = _gather_results(out, world)
    summary = {"images": len(out), "detections": sum(len(o["detections"]) for o in out)}
    if rank == 0:
        os.makedirs(work_dir, exist_ok=True)
        path = os.path.join(work_dir, "detections.json")
        with open(path, "w") as f:
            json.dump(sorted(out, key=lambda o: o["file"]), f)
        summary["path"] = path
        log.event("test", **summary)
    log.close()
    return summary
