"""How faithful is the bf16 product train step's gradient to the float32 gradient?  

Two measurements, same weights, same batch (YOLOv5s, default 640^2 bs16):

A. WHOLE STEP.  float32 reference = plain torch modules (no fused kernels, no autocast) + the HIP loss; bf16 product =
   train.Trainer._fwd_bwd (weight shadows, hand-written kernels, chained weight gradients), `--runs` repetitions at fixed
   weights; bf16 torch-AMP = plain torch modules under autocast (what `torch.autocast` gives any PyTorch user: the
   comparison class for "is OUR bf16 step worse than bf16 as such").  Per parameter tensor: relative L2 and cosine of
   one run against float32, of the MEAN of the runs against float32 (bias vs noise), and run 1 against run 0.
B. PER LAYER, IDENTICAL INPUTS.  During one product step every Conv-BN-act layer's input x and incoming gradient dy are
   recorded; then each layer is re-run ALONE on exactly those tensors — product kernels vs a float32 torch restatement
   (conv2d -> batch_norm(training) -> activation) — and dx, dW, dgamma, dbeta are compared.  This takes the chaotic
   amplification of a random-init BatchNorm network out of the comparison: a wrong or missing single-layer gradient
   cannot hide in it.  The same pass measures DESIGN section 9's common-mode hypothesis per layer: the share of the
   pre-activation gradient g = dy * act'(z) that BatchNorm backward projects out (its per-channel mean and its
   component along x_hat), i.e. by how much a relative rounding error of dy is amplified in what survives:
   amplification = ||g|| / ||g - mean(g) - x_hat * mean(g * x_hat)||.

PREDICTIONS, written before the first run (2026-10-04; the test bounds in tests/test_gpu_grad_fidelity.py are these):
  * B, every layer: cosine >= 0.999 and relative L2 <= 3e-2 for dx, dW, dgamma, dbeta (two bf16 roundings of
    operands with ~2^-9 relative error each, float32 accumulation).
  * A at random init: one bf16 run vs float32 global cosine >= 0.8; mean of 8 runs >= 0.95 (noise, not bias);
    the product step no worse than torch-AMP by more than 0.05 in global cosine.
  * A after `--pretrain` steps (weights off the chaotic random-init point): single-run global cosine >= 0.9.

    python tools/grad_fidelity.py --miopen-find [--size 640] [--batch 16] [--runs 8] [--pretrain 0] [--out profiles/r04_grad_fidelity.md]
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel_l2(a, b):
    den = float(b.double().norm())
    return float((a.double() - b.double()).norm()) / den if den > 0 else float("nan")


def cosine(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    den = float(a.norm() * b.norm())
    return float(a @ b) / den if den > 0 else float("nan")


def global_stats(xs, ys):
    num = sum(float(((a.double() - b.double()) ** 2).sum()) for a, b in zip(xs, ys))
    den = sum(float((b.double() ** 2).sum()) for b in ys)
    dot = sum(float((a.double() * b.double()).sum()) for a, b in zip(xs, ys))
    na = sum(float((a.double() ** 2).sum()) for a in xs)
    return math.sqrt(num / den), dot / math.sqrt(na * den)


def build(size, classes, dev, seed=0):
    from heltondetection_amd.detectors import build_detector
    torch.manual_seed(seed)
    m = build_detector(dict(type="YOLOv5", variant="s", num_classes=classes, img_size=size)).to(dev)
    return m.to(memory_format=torch.channels_last).train()


def plain_torch_grads(model, batch, amp: bool):
    """Plain torch modules (ConvBNAct.fused off): float32 reference, or the torch-AMP bf16 step."""
    from heltondetection_amd import ops
    from heltondetection_amd.layers import ConvBNAct
    ConvBNAct.fused = False
    try:
        model.zero_grad(set_to_none=True)
        ops.bn_arena_reset(batch[0].device)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            loss, _ = model.forward_train(*batch)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss), [p.grad.detach().float().clone() if p.grad is not None else None for p in model.parameters()]
    finally:
        ConvBNAct.fused = True


def product_grads(trainer, model, batch):
    loss, _ = trainer._fwd_bwd(*batch)
    trainer.reducer.finish()
    torch.cuda.synchronize()
    out = []
    for p in model.parameters():
        g = trainer.grad_of(p)
        out.append(g.detach().float().clone() if g is not None else None)
    return float(loss), out


# ------------------------------------------------------------------------------------------ B: per layer, identical inputs
class Recorder:
    """Wraps layers.conv_bn_act for one product step: keeps every call's input and the gradient arriving at its output(s)."""

    def __init__(self):
        from heltondetection_amd import layers
        self.layers = layers
        self.orig = layers.conv_bn_act
        self.records = []
        self.depth = 0

    def __enter__(self):
        self.layers.conv_bn_act = self._call
        return self

    def __exit__(self, *exc):
        self.layers.conv_bn_act = self.orig
        return False

    def _call(self, conv, bn, act, x, residual=None, residual_after_act=False, out=None, fork=False):
        self.depth += 1
        try:
            r = self.orig(conv, bn, act, x, residual, residual_after_act, out, fork)
        finally:
            self.depth -= 1
        if self.depth:      # the fork fallback re-enters conv_bn_act: the outer call records
            return r
        ys = r[0] if fork else r
        outs = ys if isinstance(ys, tuple) else (ys,)
        e = dict(conv=conv, bn=bn, act=act, x=x.detach(), dys=[None] * len(outs),
                 res=residual.detach() if (residual is not None and not residual_after_act) else None)
        for i, o in enumerate(outs):
            if o.requires_grad:
                o.register_hook(lambda g, i=i, e=e: e["dys"].__setitem__(i, g.detach().clone()))
        self.records.append(e)
        return r


def _act(name, t):
    return F.silu(t) if name == "silu" else (torch.relu(t) if name == "relu" else t)


def layer_fidelity(rec: Recorder, names: dict, dev):
    """-> list of per-layer dicts (product vs float32 on the recorded x / dy)."""
    from heltondetection_amd import ops
    from heltondetection_amd.layers import LowpWeights, ShadowConv2d
    rows = []
    ops.bn_arena_reset(dev)
    for e in rec.records:
        conv, bn = e["conv"], e["bn"]
        if any(d is None for d in e["dys"]):
            continue
        dy = torch.cat([d.float() for d in e["dys"]], 1).contiguous(memory_format=torch.channels_last)
        x = e["x"]
        w_used = conv.weight_lp if (isinstance(conv, ShadowConv2d) and conv.weight_lp is not None) else conv.weight
        stem = not x.dtype == torch.bfloat16
        rm, rv = bn.running_mean.clone(), bn.running_var.clone()
        # ---- product: the layer alone on (x, dy)
        xp = x.clone().requires_grad_(not stem)
        LowpWeights.on = True
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                yp = rec.orig(conv, bn, e["act"], xp, e["res"], False)
            leaves = ([xp] if not stem else []) + [w_used, bn.weight, bn.bias]
            gp = torch.autograd.grad(yp, leaves, dy.to(yp.dtype))
            ops.flush_wgrad()
        finally:
            LowpWeights.on = False
        bn.running_mean.copy_(rm)
        bn.running_var.copy_(rv)
        gp = [g.float() for g in gp]
        if stem:
            gp = [None] + gp
        # ---- float32 restatement on the same tensors (the weight VALUES the product used)
        x32 = x.float().contiguous(memory_format=torch.channels_last).requires_grad_(not stem)
        w32 = w_used.detach().float().requires_grad_(True)
        ga, be = bn.weight.detach().float().requires_grad_(True), bn.bias.detach().float().requires_grad_(True)
        z = F.conv2d(x32, w32, None, conv.stride, conv.padding)
        zn = F.batch_norm(z, None, None, ga, be, True, 0.0, bn.eps)
        pre = zn if e["res"] is None else zn + e["res"].float()
        y32 = _act(e["act"], pre)
        leaves = ([x32] if not stem else []) + [w32, ga, be]
        gr = list(torch.autograd.grad(y32, leaves + [zn], dy))
        g_pre = gr.pop()      # gradient at the BatchNorm output = dy * act'
        if stem:
            gr = [None] + gr
        # ---- common-mode share of the pre-activation gradient (per channel, then energy-weighted over the layer)
        with torch.no_grad():
            g = g_pre * ga[None, :, None, None]          # d / d x_hat
            mean = z.mean((0, 2, 3), keepdim=True)
            xhat = (z - mean) * torch.rsqrt(z.var((0, 2, 3), unbiased=False, keepdim=True) + bn.eps)
            m1 = g.mean((0, 2, 3))
            m2 = (g * xhat).mean((0, 2, 3))
            e2 = (g * g).mean((0, 2, 3))
            removed = ((m1 * m1 + m2 * m2) / e2.clamp_min(1e-38)).clamp(max=1 - 1e-7)
            amp_c = torch.rsqrt(1 - removed)
            amp_layer = math.sqrt(float(e2.sum()) / max(float((e2 * (1 - removed)).sum()), 1e-38))
            cm_ratio = (m1.abs() / (e2 - m1 * m1).clamp_min(1e-38).sqrt())
        row = dict(name=names.get(id(conv), "?"), k=conv.kernel_size[0], s=conv.stride[0], cin=conv.in_channels, cout=conv.out_channels,
                   hw=tuple(z.shape[2:]), amp_layer=amp_layer, amp_med=float(amp_c.median()), amp_max=float(amp_c.max()),
                   cm_med=float(cm_ratio.median()), cm_max=float(cm_ratio.max()))
        for key, a, b in zip(("dx", "dw", "dgamma", "dbeta"), gp, gr):
            if a is None:
                row[key] = None
            else:
                row[key] = (rel_l2(a.reshape(b.shape), b), cosine(a, b))
        rows.append(row)
    return rows


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--runs", type=int, default=8)
    ap.add_argument("--pretrain", type=int, default=0, help="product train steps (lr 1e-3, 4 fixed batches) before measuring")
    ap.add_argument("--out", default="")
    ap.add_argument("--json", default="")
    ap.add_argument("--skip-layers", action="store_true")
    ap.add_argument("--miopen-find", action="store_true", help="MIOpen solver search + the shipped find-db, as bench.py runs")
    args = ap.parse_args(argv)
    from heltondetection_amd import synthetic as syn, tuning
    from heltondetection_amd.train import Trainer
    # process-wide settings are the CALLER's (the test suite runs this in-process: a solver search left switched on changed
    # what every later test's library convolutions did); the command line below sets them
    if args.miopen_find:
        tuning.use_shipped_find_db()
        torch.backends.cudnn.benchmark = True
    dev = torch.device("cuda", 0)
    g = syn.make_generator(1234)

    def make_batch():
        img = syn.make_images(args.batch, args.size, g).to(dev).contiguous(memory_format=torch.channels_last)
        gtb, gtc, mv = syn.make_targets(args.batch, args.size, g, "coco", args.classes, 40)
        return (img, gtb.to(dev), gtc.to(dev), mv.to(dev))

    batch = make_batch()
    model = build(args.size, args.classes, dev)
    trainer = Trainer(model, lr=1e-3, device=dev, total_steps=max(args.pretrain, 1) + 10, warmup_steps=max(args.pretrain // 10, 1),
                      use_graph=False)
    if args.pretrain:
        pool = [make_batch() for _ in range(4)]
        for s in range(args.pretrain):
            loss, _ = trainer.step(*pool[s % 4])
        print(f"pretrained {args.pretrain} steps, last loss {float(loss):.4f}", flush=True)
    trainer.shadows.refresh()
    names = {id(m): n for n, m in model.named_modules()}
    pnames = [n for n, _ in model.named_parameters()]

    # ------------------------------------------------------------------ A: whole step
    l32a, g32 = plain_torch_grads(model, batch, amp=False)
    l32b, g32b = plain_torch_grads(model, batch, amp=False)
    lamp0, gamp0 = plain_torch_grads(model, batch, amp=True)
    lamp1, gamp1 = plain_torch_grads(model, batch, amp=True)
    runs, losses = [], []
    for r in range(args.runs):
        l, gs = product_grads(trainer, model, batch)
        runs.append(gs)
        losses.append(l)
    idx = [i for i, t in enumerate(g32) if t is not None and all(r[i] is not None for r in runs) and gamp0[i] is not None]
    mean = [sum(r[i] for r in runs) / len(runs) for i in idx]
    ref = [g32[i] for i in idx]
    summary = dict(size=args.size, batch=args.batch, runs=args.runs, pretrain=args.pretrain,
                   loss_fp32=l32a, loss_amp=lamp0, loss_product=losses,
                   fp32_rerun=global_stats([g32b[i] for i in idx], ref),
                   product_single=global_stats([runs[0][i] for i in idx], ref),
                   product_mean=global_stats(mean, ref),
                   product_run1_vs_run0=global_stats([runs[1][i] for i in idx], [runs[0][i] for i in idx]) if args.runs > 1 else None,
                   amp_single=global_stats([gamp0[i] for i in idx], ref),
                   amp_run1_vs_run0=global_stats([gamp1[i] for i in idx], [gamp0[i] for i in idx]),
                   product_vs_amp=global_stats([runs[0][i] for i in idx], [gamp0[i] for i in idx]))
    per = []
    for j, i in enumerate(idx):
        per.append(dict(name=pnames[i], numel=ref[j].numel(), norm=float(ref[j].norm()),
                        single=(rel_l2(runs[0][i], ref[j]), cosine(runs[0][i], ref[j])),
                        mean=(rel_l2(mean[j], ref[j]), cosine(mean[j], ref[j])),
                        rerun=(rel_l2(runs[1][i], runs[0][i]), cosine(runs[1][i], runs[0][i])) if args.runs > 1 else None,
                        amp=(rel_l2(gamp0[i], ref[j]), cosine(gamp0[i], ref[j]))))
    print(json.dumps(summary), flush=True)

    # ------------------------------------------------------------------ B: per layer
    rows = []
    if not args.skip_layers:
        with Recorder() as rec:
            product_grads(trainer, model, batch)
        rows = layer_fidelity(rec, names, dev)
        worst = {k: (max((r[k][0] for r in rows if r[k] is not None), default=0.0), min((r[k][1] for r in rows if r[k] is not None), default=1.0))
                 for k in ("dx", "dw", "dgamma", "dbeta")}
        summary["layers"] = len(rows)
        summary["layer_worst"] = worst
        summary["amp_layer_max"] = max(r["amp_layer"] for r in rows)
        print(json.dumps({"layer_worst": worst, "amp_layer_max": summary["amp_layer_max"]}), flush=True)

    if args.json:
        with open(args.json, "w") as f:
            json.dump(dict(summary=summary, per_param=per, layers=rows), f)
    if args.out:
        write_markdown(args.out, summary, per, rows)
    return summary, per, rows


def write_markdown(path, summary, per, rows):
    def f2(t):
        return "—" if t is None else f"{t[0]:.3g} / {t[1]:.4f}"
    L = []
    L.append(f"## YOLOv5s {summary['size']}² bs{summary['batch']}, {summary['pretrain']} pre-training steps, {summary['runs']} bf16 product runs\n")
    L.append(f"loss: float32 {summary['loss_fp32']:.5f}, torch-AMP bf16 {summary['loss_amp']:.5f}, product bf16 "
             + " ".join(f"{v:.5f}" for v in summary["loss_product"]) + "\n")
    L.append("| all parameters as one vector | relative L2 | cosine |\n|---|---|---|")
    for k, label in (("fp32_rerun", "float32 run 2 vs float32 run 1"), ("product_single", "bf16 product, one run vs float32"),
                     ("product_mean", f"bf16 product, mean of {summary['runs']} runs vs float32"),
                     ("product_run1_vs_run0", "bf16 product, run 1 vs run 0"), ("amp_single", "bf16 torch-AMP (plain modules), one run vs float32"),
                     ("amp_run1_vs_run0", "bf16 torch-AMP, run 1 vs run 0"), ("product_vs_amp", "bf16 product vs bf16 torch-AMP")):
        v = summary.get(k)
        if v:
            L.append(f"| {label} | {v[0]:.4g} | {v[1]:.5f} |")
    L.append("\n### Per parameter tensor, sorted by the single-run cosine (relative L2 / cosine against float32)\n")
    L.append("| parameter | elements | product, one run | product, mean of runs | product run 1 vs run 0 | torch-AMP, one run |\n|---|---|---|---|---|---|")
    for r in sorted(per, key=lambda r: r["single"][1]):
        L.append(f"| {r['name']} | {r['numel']} | {f2(r['single'])} | {f2(r['mean'])} | {f2(r['rerun'])} | {f2(r['amp'])} |")
    if rows:
        L.append("\n### Per layer on IDENTICAL recorded inputs: product kernels vs float32 restatement (relative L2 / cosine), and the "
                 "share of the pre-activation gradient that BatchNorm backward projects out (amplification of dy's rounding error: "
                 "layer, median channel, worst channel; |mean| / std of g per channel: median, max)\n")
        L.append("| layer | k/s | Cin→Cout | HxW | dx | dW | dγ | dβ | amplification | common-mode |\n|---|---|---|---|---|---|---|---|---|---|")
        for r in rows:
            L.append(f"| {r['name']} | {r['k']}/{r['s']} | {r['cin']}→{r['cout']} | {r['hw'][0]}x{r['hw'][1]} | {f2(r['dx'])} | {f2(r['dw'])} | "
                     f"{f2(r['dgamma'])} | {f2(r['dbeta'])} | {r['amp_layer']:.2f} / {r['amp_med']:.2f} / {r['amp_max']:.1f} | "
                     f"{r['cm_med']:.3f} / {r['cm_max']:.2f} |")
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "a") as f:
        f.write("\n".join(L) + "\n\n")


if __name__ == "__main__":
    main()
