"""Convergence A/B on the GPU box: does the bf16 product step train like the float32 step?

`examples/train_eval_synthetic.py` (synthetic COCO-format set with painted, learnable objects -> hipGraph-replayed training
with device-side Mosaic -> hipGraph-replayed inference -> mAP50) is run `--seeds` times per arm, each in a fresh process:
  fp32       the step without autocast (library float32 convolutions, fused float32 BatchNorm)
  bf16       the product step: bf16 autocast, bf16 weight shadows, the hand-written kernels
  torch_amp  plain torch modules under bf16 autocast (what `torch.autocast` alone gives)
and mAP50 / mAP50-95 / last loss are tabulated as mean and min-max spread.  README.md:130's "+-0.2" (quoted by north_star) is
the bar the arms are compared at — on the 0-100 scale of the README's tables; this tool reports mAP on the 0-1 scale its
evaluator uses and converts in the table.

    python tools/convergence_ab.py [--size 160] [--images 64] [--steps 600] [--seeds 3] [--out profiles/r04_convergence_ab.md]
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=160)
    ap.add_argument("--images", type=int, default=64)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=600)
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--arms", default="fp32,bf16,torch_amp")
    ap.add_argument("--mosaic", type=float, default=0.5)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    epochs = max(args.steps // max(args.images // args.batch, 1), 1)
    rows = []
    for arm in args.arms.split(","):
        for seed in range(args.seeds):
            wd = tempfile.mkdtemp(prefix="hd_ab_")
            cmd = [sys.executable, os.path.join(ROOT, "examples", "train_eval_synthetic.py"), "--workdir", wd, "--images", str(args.images),
                   "--size", str(args.size), "--batch", str(args.batch), "--epochs", str(epochs), "--mosaic", str(args.mosaic),
                   "--amp", arm, "--seed", str(seed)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not line:
                print(f"[ab] {arm} seed {seed} FAILED rc={r.returncode}\n{r.stderr[-2000:]}", flush=True)
                rows.append({"amp": arm, "seed": seed, "error": r.stderr[-300:]})
                continue
            rec = json.loads(line[-1])
            rows.append(rec)
            print(f"[ab] {arm} seed {seed}: mAP50 {rec['mAP50']:.4f} mAP50-95 {rec['mAP50_95']:.4f} loss {rec['first_loss']:.3f} -> "
                  f"{rec['last_loss']:.3f} ({rec['steps']} steps, hipGraph {rec['hip_graph']})", flush=True)
    L = [f"## YOLOv5s {args.size}² on {args.images} painted synthetic images, batch {args.batch}, {epochs} epochs "
         f"({epochs * (args.images // args.batch)} steps), Mosaic p={args.mosaic}, {args.seeds} seeds per arm (weight init + Mosaic draws)\n",
         "| arm | mAP50 ×100: per seed | mean | min – max | mAP50-95 ×100 mean | last loss mean |", "|---|---|---|---|---|---|"]
    summary = {}
    for arm in args.arms.split(","):
        ok = [r for r in rows if r.get("amp") == arm and "error" not in r]
        if not ok:
            continue
        m = [100 * r["mAP50"] for r in ok]
        m95 = [100 * r["mAP50_95"] for r in ok]
        ll = [r["last_loss"] for r in ok]
        summary[arm] = {"mAP50_x100": m, "mean": sum(m) / len(m), "min": min(m), "max": max(m)}
        L.append(f"| {arm} | {' '.join(f'{v:.2f}' for v in m)} | {sum(m) / len(m):.2f} | {min(m):.2f} – {max(m):.2f} | "
                 f"{sum(m95) / len(m95):.2f} | {sum(ll) / len(ll):.4f} |")
    print(json.dumps(summary), flush=True)
    if args.out:
        with open(args.out, "a") as f:
            f.write("\n".join(L) + "\n\n")


if __name__ == "__main__":
    main()
