"""One-line command interface (reference README.md:12-14: train / evaluate / test from a config file).

    python -m heltondetection_amd.cli train --config configs/yolov5s_synth.yaml --work-dir runs/y5s [--set train.lr=0.002 ...]
    python -m heltondetection_amd.cli eval  --config ... --work-dir runs/y5s --ckpt runs/y5s/last.pt
    python -m heltondetection_amd.cli test  --config ... --work-dir runs/y5s --ckpt runs/y5s/last.pt [--source DIR]
    python -m heltondetection_amd.cli show  --config ...                     (prints the resolved config)

Multi-GPU: launch the same command under `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1
-m heltondetection_amd.cli train ...`; ranks are read from the environment (one process per GPU, RCCL).
The last line printed by rank 0 is a JSON summary of the run.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from typing import List, Optional

from .config import ConfigError, dump, load_config


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="python -m heltondetection_amd.cli", description=__doc__.split("\n\n")[0])
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name, hlp in (("train", "train the configured detector"), ("eval", "mAP50 / mAP50-95 on the val split"),
                      ("test", "inference on images, no ground truth"), ("show", "print the resolved config")):
        p = sub.add_parser(name, help=hlp)
        p.add_argument("--config", required=True, help=".yaml / .json / .py config file (may inherit through _base_)")
        p.add_argument("--set", nargs="*", default=[], metavar="section.key=value", help="override config entries")
        if name != "show":
            p.add_argument("--work-dir", default=None, help="run directory (default: runs/<config name>)")
            p.add_argument("--device", default=None, help="torch device (default: cuda:<LOCAL_RANK>, cpu without a GPU)")
        if name == "train":
            p.add_argument("--resume", default=None, help="checkpoint to continue from")
        if name in ("eval", "test"):
            p.add_argument("--ckpt", default=None, help="checkpoint to load (default: <work-dir>/last.pt when it exists)")
        if name == "test":
            p.add_argument("--source", default=None, help="directory of images (default: the val split)")
    return ap


def main(argv: Optional[List[str]] = None) -> dict:
    args = build_parser().parse_args(argv)
    try:
        cfg = load_config(args.config, args.set)
    except ConfigError as e:
        raise SystemExit(f"config error: {e}")
    if args.cmd == "show":
        print(dump(cfg))
        return cfg
    from . import runners
    work = args.work_dir or os.path.join("runs", os.path.splitext(os.path.basename(args.config))[0])
    ckpt = getattr(args, "ckpt", None)
    if args.cmd in ("eval", "test") and ckpt is None and os.path.exists(os.path.join(work, "last.pt")):
        ckpt = os.path.join(work, "last.pt")
    if args.cmd == "train":
        out = runners.train(cfg, work, device=args.device, resume=args.resume)
    elif args.cmd == "eval":
        out = runners.evaluate(cfg, work, ckpt=ckpt, device=args.device)
    else:
        out = runners.test(cfg, work, ckpt=ckpt, device=args.device, source=args.source)
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"cmd": args.cmd, "work_dir": work, **{k: v for k, v in out.items()}}, default=str), flush=True)
    return out


if __name__ == "__main__":
    main(sys.argv[1:])
