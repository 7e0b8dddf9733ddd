#!/usr/bin/env python3
"""bench.py — images/sec of one training step, YOLOv5s 640x640 bs=16/GPU (BASELINE.json metric).

  python bench.py --gpus 1 --steps 20 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W
  python bench.py --model frcnn            (BASELINE config 3: FasterRCNN-R50-PAFPN-DecoupledHead 832x832 bs16)

A step = forward (backbone + neck on MIOpen, bf16 autocast, channels-last, fused HIP BN+act) + HIP target
assignment + HIP loss / RoI ops + backward + bucketed RCCL all-reduce + fused AdamW + LR schedule, on synthetic
batches already resident in HBM.  Forward + loss + backward + gradient packing replay from one hipGraph
(--no-graph: eager, where the all-reduce is launched from autograd hooks and overlaps backward); with the graph
the collectives run after the replay.  Weak scaling: bs=16 per GPU at every N.  Rank 0 prints ONE JSON line.
Two guards keep a corrupted graph run from being reported (finite loss, replay-vs-eager loss): if one trips the
steps are re-timed eagerly and config.hip_graph_error says so.  A graph run that is merely SLOW (replays slower
than the eager steps they replace) is reported as measured and flagged (config.hip_graph_slow_replay), never hidden.

The stdout line is COMPACT (< 4 KB, compact_line): the contract's keys, the headline `roofline` (every kernel of the family as
[fraction of its bound, summed ms per step]), `cpu_baseline`, one short entry per secondary roofline and per companion leg.  It
is printed once as soon as the headline, its rooflines and the CPU baseline are in hand and again, completed, as the last thing
the process writes.  The full record described below goes to bench_detail.json (next to this file and under gpurun_out/).

Objects of the full record:
  roofline      the hand-written kernel FAMILY with the largest summed launch time in the step (the convolution family:
                1x1 forward / one-launch 1x1 backward / 3x3 / stem kernels), step-weighted: algorithmic bytes of all its
                launches / their summed HIP-event time, measured live on the model's real layer shapes; every kernel of the
                family is also priced against its own bound in `per_kernel` (HBM for the streaming kernels, the dense bf16
                MFMA peak for the 3x3 implicit GEMMs).  Same rule for --model frcnn.
  roofline_secondary   the other families / kernels on the path (fused BN+act, loss dense gradient, RoIAlign, pairwise IoU),
                timed the same way, one entry each.
  cpu_baseline  this repo's PyTorch-CPU restatement of the same step (oracle/torch_ref.py — the reference's CPU path is not
                in the snapshot) on the box's host cores, bounded sample; --model frcnn: the oracle's RoIAlign + RPN ops.
  kernels       (N=1 only) HIP-event rooflines of the other hot-path kernels at BASELINE.json's config shapes —
                parity-case workloads, not the headline metric.
  packed        (N=1 only) the same step with the gradients packed into the all-reduce buckets as every rank of an N > 1 run
                does (HD_DIRECT_GRADS=0): the per-rank program a scaling efficiency should be quoted against.
  fp32, size_832, frcnn_832, frcnn_1280_visdrone   companion legs under the same timing contract (north_star names 640^2 and
                832^2; BASELINE.json configs 3 and 5), each with its own dominant-family roofline; skipped, and marked so, once
                the run has used its wall-clock budget (--leg-budget seconds, default 240: the whole default run then ends within
                about five minutes even on a box that has to search every library solver anew).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E spec (6.3 TB/s is the measured copy ceiling)
HBM_COPY_GBS = 6290.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--model", choices=("yolov5", "frcnn"), default="yolov5",
                    help="yolov5: BASELINE.json's metric (configs 2 and 4); frcnn: config 3, FasterRCNN-R50-PAFPN-DecoupledHead")
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=0, help="image side; default 640 (yolov5) / 832 (frcnn)")
    ap.add_argument("--variant", default="s")
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-micro", action="store_true")
    ap.add_argument("--no-roofline", action="store_true",
                    help="skip the roofline legs (profiling runs: keeps the kernel trace to the train steps alone)")
    ap.add_argument("--fp32", action="store_true", help="disable bf16 autocast (debug)")
    ap.add_argument("--no-graph", action="store_true",
                    help="run forward+backward eagerly instead of replaying a hipGraph (the graph has its memset "
                         "nodes rewritten into kernel nodes, DESIGN.md section 9, and is guarded: a non-finite or "
                         "eager-mismatching loss makes bench.py re-time eagerly)")
    ap.add_argument("--graph-sync", choices=("auto", "on", "off"), default="auto",
                    help="drain the stream after each replayed graph segment before its collective (auto: only for the gloo "
                         "rehearsal backend, see train.Trainer)")
    ap.add_argument("--graph-split", choices=("auto", "on", "off"), default="auto",
                    help="multi-rank hipGraph mode: cut the captured step at the gradient-bucket boundaries (on / auto) or replay "
                         "one graph and launch the collectives behind it (off)")
    ap.add_argument("--layer-detail", default="", help="write the per-launch table of the hand-written convolution kernels "
                                                       "(name, shape, us, GB/s, TFLOP/s) of the roofline passes to this file")
    ap.add_argument("--no-legs", action="store_true", help="skip the fp32 and 832x832 companion legs of the default run")
    ap.add_argument("--leg-steps", type=int, default=10, help="timed steps of each companion leg (after 5 warm-up steps)")
    ap.add_argument("--leg-budget", type=float, default=240.0,
                    help="seconds since start after which no further companion leg is started (they are marked skipped)")
    ap.add_argument("--style", choices=("coco", "visdrone"), default="coco", help="synthetic ground truth (SURVEY.md section 8d)")
    ap.add_argument("--no-miopen-find", action="store_true",
                    help="skip MIOpen's per-shape solver search (faster first step, ~20%% slower steady state)")
    args = ap.parse_args()
    if args.size <= 0:
        args.size = 640 if args.model == "yolov5" else 832
    return args


def ensure_world(args) -> None:
    """`--gpus N` is a promise about the number of ranks.  Called before anything touches the GPU:
      * WORLD_SIZE == N (the driver's torch.distributed.run launch, or N == 1 run directly): nothing to do;
      * no launcher in the environment and N > 1: start the N ranks ourselves as fresh child processes
        (`python -m torch.distributed.run ... bench.py <same flags>`), pass rank 0's JSON line through and exit with the
        launcher's code — this process never initialises HIP, so nothing is exec'ed or forked behind a live GPU context;
      * a launcher is present but disagrees with --gpus: exit non-zero with the launch line (never a line that claims
        n_gpus = WORLD_SIZE for a run asked to use --gpus N)."""
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is not None:
        if int(env_world) != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={env_world}; launch as\n  python -m torch.distributed.run "
                             f"--nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port P bench.py --gpus {args.gpus} ...")
        return
    if args.gpus == 1:
        return
    import socket
    import subprocess
    if os.environ.get("HD_DIST_BACKEND") != "gloo":
        n_dev = torch.cuda.device_count()  # counting devices does not initialise HIP on this image
        if n_dev < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node shows {n_dev} GPU(s) (HD_DIST_BACKEND=gloo rehearses "
                             "several ranks on one device and labels the line as a rehearsal)")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print(f"[bench] --gpus {args.gpus} without a launcher: starting {' '.join(cmd)}", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


def model_cfg(args):
    if args.model == "frcnn":
        return dict(type="FasterRCNN", num_classes=args.classes)
    return dict(type="YOLOv5", variant=args.variant, num_classes=args.classes, img_size=args.size)


def model_name(args):
    return "FasterRCNN-R50-PAFPN-DecoupledHead" if args.model == "frcnn" else f"YOLOv5{args.variant}"


def make_batches(n, B, size, nc, seed, device, style="coco"):
    from heltondetection_amd import synthetic as syn
    g = syn.make_generator(seed)
    out = []
    for _ in range(n):
        img = syn.make_images(B, size, g).to(device).contiguous(memory_format=torch.channels_last)
        gtb, gtc, mv = syn.make_targets(B, size, g, style, nc, 40 if style == "coco" else 600)
        out.append((img, gtb.to(device), gtc.to(device), mv.to(device)))
    return out


def loss_bwd_bytes(B, size, no, na=3, strides=(8, 16, 32)):
    """Algorithmic bytes of one hd_yolo_loss_bwd_dense launch: write the dense gradient
    4*B*n*no and read the objectness residual the forward pass left behind, 4*B*n."""
    n = sum(na * (size // s) ** 2 for s in strides)
    return 4 * B * n * no + 4 * B * n


def pmc_traffic(kernel, shape: str = ""):
    """HBM bytes from the committed counter run of THIS round's build (profiles/r05_pmc_traffic.json, made by tools/prof_r05.sh +
    tools/pmc_traffic_r05.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled for 16-byte streams as
    MI355X_MICROARCH.md prescribes).  `kernel`: a family name of that file ("hand-written convolutions", "fused BN+act") -> bytes
    per train step (YOLOv5s 640^2 bs16); a single-kernel key of its "single_kernels" table -> bytes per launch; or a kernel-name
    fragment -> bytes per launch of the first step kernel whose name contains it.  None when the file has no such entry (other
    model / size).  PMC counters cannot be read from inside this process, so this is a look-up, and the line says so."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r05_pmc_traffic.json")))
        fam = rec.get("families_per_step", {}).get(kernel)
        if fam is not None:
            return int(fam["traffic_bytes_per_step"])
        one = rec.get("single_kernels", {}).get(kernel)
        if one is not None:
            return int(one["traffic_bytes_per_launch"])
        for name, v in rec.get("kernels", {}).items():
            if kernel in name:
                return int(v["traffic_bytes_per_launch"])
    except Exception:
        pass
    return None


TRAFFIC_SOURCE = "profiles/r05_pmc_traffic.json (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this round's build: eager " \
                 "YOLOv5s 640x640 bs16 train steps (tools/prof_r05.sh: families = bytes per train step, step kernels = bytes per launch) " \
                 "and the single-kernel probe at the micro leg's shapes (tools/pmc_traffic_r05.sh); null = no entry for this workload).  " \
                 "The counter run predates the round's last two changes: the 3x3 weight gradients' 512-thread pair form was removed after " \
                 "it (twice the partial slabs: round 4's counter A/B of exactly that switch measured +0.3 GB per step on the convolution " \
                 "family, so 6.97 rather than 6.67 GB is the figure to expect), and bn_bwd_reduce runs on twice the workgroups (same bytes)"


def usable_cores() -> int:
    """Host threads this process may really use: affinity mask, cgroup CPU quota, and the GPU box's
    per-GPU CPU share (16) — os.cpu_count() reports the whole host and would oversubscribe."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, 16))


def cpu_baseline(args):
    """Our PyTorch-CPU restatement of the same train step (fp32), bounded sample, host cores of this box."""
    from heltondetection_amd import synthetic as syn
    from heltondetection_amd.detectors import build_detector
    from oracle import torch_ref as tr
    cores = usable_cores()
    torch.set_num_threads(cores)
    model = build_detector(model_cfg(args))
    model = model.to(memory_format=torch.channels_last).train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    g = syn.make_generator(1234)
    B = args.batch
    img = syn.make_images(B, args.size, g).contiguous(memory_format=torch.channels_last)
    gtb, gtc, mv = syn.make_targets(B, args.size, g, "coco", args.classes, 40)
    head = model.head

    def step():
        opt.zero_grad(set_to_none=True)
        preds = head(model.extract(img))
        loss = tr.yolo_loss(preds, head.strides, head.anchors, gtb, gtc, mv, 4.0, head.box_gain, head.obj_gain,
                            head.cls_gain, head.balance)[0]
        loss.backward()
        opt.step()

    # SURVEY.md section 8d protocol, scaled to the bounded sample the bench contract allows (about 10-30 s of CPU work):
    # 2 untimed warm-up steps, then 20 individually timed steps (fewer only past 30 s), the MEDIAN step time is what is reported
    warm, want, budget_s = 2, 20, 30.0   # SURVEY.md section 8d asks for >= 20 timed iterations
    load0 = os.getloadavg()[0] if hasattr(os, "getloadavg") else None
    for i in range(warm):
        print(f"[bench cpu] warm-up step {i} ...", file=sys.stderr, flush=True)
        step()
    per, t_all = [], time.perf_counter()
    while len(per) < want and (time.perf_counter() - t_all < budget_s or len(per) < 3):
        t0 = time.perf_counter()
        step()
        per.append(time.perf_counter() - t0)
        if len(per) % 5 == 0:
            print(f"[bench cpu] {len(per)} steps timed, last {per[-1]:.2f} s", file=sys.stderr, flush=True)
    srt = sorted(per)
    med = srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2])
    return {"value": round(B / med, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model_string(), "threads": torch.get_num_threads(), "host_logical_cpus": os.cpu_count(),
            "loadavg_1min_before": load0, "step_s": {"median": round(med, 3), "min": round(srt[0], 3), "max": round(srt[-1], 3)},
            "sample": f"median of {len(per)} individually timed train steps of YOLOv5s {args.size}x{args.size} bs={B} fp32 after "
                      f"{warm} warm-up steps (our PyTorch-CPU restatement of the step, oracle/torch_ref.py loss; not the reference's "
                      f"CPU path, which is not in the snapshot)"}


def cpu_baseline_frcnn(args):
    """--model frcnn: no CPU restatement of the whole two-stage step exists, so the stated baseline is the detection-specific
    half of it on the CPU checker (oracle/hd_oracle.c through oracle/c_oracle.py, one thread): RPN proposals (top-k + decode +
    NMS) and RoIAlign forward over the FPN levels at this configuration's sizes, per image."""
    import math
    import numpy as np
    from heltondetection_amd import synthetic as syn
    from oracle import c_oracle as co
    g = syn.make_generator(1234)
    S, B = args.size, min(args.batch, 2)          # a bounded sample: two images
    strides = [4, 8, 16, 32, 64]
    hw = [(math.ceil(S / s), math.ceil(S / s)) for s in strides]
    base = np.stack([co.anchor_base([8.0 * s], [0.5, 1.0, 2.0]) for s in strides])
    anchors = co.anchor_grid([h for h, _ in hw], [w for _, w in hw], strides, base)
    ls = [h * w * 3 for h, w in hw]
    NA = anchors.shape[0]
    logits = torch.randn((B, NA), generator=g).numpy()
    deltas = (torch.randn((B, NA, 4), generator=g) * 0.2).numpy()
    feats = [torch.randn((B, 256, S // s, S // s), generator=g).numpy() for s in (4, 8, 16, 32)]
    per_img = 512
    rb = syn.random_boxes(B * per_img, float(S), g, 16.0, S / 2.0)
    rois = torch.cat([torch.arange(B).repeat_interleave(per_img)[:, None].float(), rb], 1).numpy()
    t_rpn, t_roi = [], []
    for it in range(4):
        t0 = time.perf_counter()
        co.rpn_proposals(logits, deltas, anchors, ls, 2000, 1000, 0.7, 0.0, float(S), float(S))
        t1 = time.perf_counter()
        co.roi_align_fwd(feats, rois, 7, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 2, True, 2, 224.0, 4)
        t2 = time.perf_counter()
        if it:   # the first pass warms the caches
            t_rpn.append(t1 - t0)
            t_roi.append(t2 - t1)
        print(f"[bench cpu] frcnn ops pass {it}: rpn {t1 - t0:.2f} s, roi_align {t2 - t1:.2f} s", file=sys.stderr, flush=True)
    med = lambda v: sorted(v)[len(v) // 2]
    per_image = (med(t_rpn) + med(t_roi)) / B
    return {"value": round(1.0 / per_image, 3), "unit": "images/s", "cores": 1, "kind": "port", "cpu_model": cpu_model_string(),
            "ops_s_per_image": {"rpn_proposals": round(med(t_rpn) / B, 4), "roi_align_fwd": round(med(t_roi) / B, 4)},
            "sample": f"median of 3 passes over {B} images at {S}x{S}: RPN proposals (NA={NA}, pre 2000 / post 1000) + RoIAlign forward "
                      f"({per_img} RoIs per image, C=256, P=7, 4 levels) on the serial C checker (oracle/hd_oracle.c) — ONLY the "
                      "detection-specific ops of the step, not the backbone: a stated baseline for those ops, not a step time"}


def _yolo_views(B, size, no, device, na=3, strides=(8, 16, 32)):
    """Channels-last head outputs viewed as [B,na,H,W,no] (what YOLOv5Head.forward hands to the kernels)."""
    from heltondetection_amd.heads import YOLOV5_ANCHORS
    views = []
    for s in strides:
        H = size // s
        conv = torch.randn((B, na * no, H, H), device=device).contiguous(memory_format=torch.channels_last)
        views.append(conv.view(B, na, no, H, H).permute(0, 1, 3, 4, 2))
    return views, strides, YOLOV5_ANCHORS


def micro_kernels(device):
    """HIP-event timings of the other hot-path kernels at BASELINE.json config shapes (parity-case workloads)."""
    import math
    from heltondetection_amd import ops, synthetic as syn
    res = []

    def timeit(fn, iters=10, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        evs = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in evs)
        return ts[len(ts) // 2]

    def ktime(name, fn, iters=10, warm=3):
        """Kernel-only HIP-event timing (the launch stamps its own event pair) of a single-kernel op."""
        for _ in range(warm):
            fn()
        ops.enable_timing(name)
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        ts = sorted(ops.collect_timing()[name])
        ops.disable_timing()
        return ts[len(ts) // 2]

    def add(name, ms, nbytes, shape, bound="hbm", tkey=None):
        print(f"[bench micro] {name[:40]} {shape[:48]}: {ms:.4f} ms", file=sys.stderr, flush=True)
        gbs = nbytes / ms / 1e6
        res.append({"kernel": name, "shape": shape, "ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 2),
                    "GB/s": round(gbs, 1), "frac_of_8TB/s": round(gbs / HBM_PEAK_GBS, 4), "bound": bound})
        if tkey is not None:   # HBM bytes per launch from this round's counter run on exactly this shape (pmc_traffic)
            res[-1]["traffic"] = pmc_traffic(tkey)

    g = syn.make_generator(7)
    # config 5: 1280^2 A=3 anchors x ~500 gt, materialised + fused (B=2 keeps the output at 1.6 GB)
    B, N, M = 2, 409200, 500
    a = torch.stack([syn.random_boxes(N, 1280.0, g, 8.0, 512.0) for _ in range(1)])[0].to(device)
    b = torch.stack([syn.random_boxes(M, 1280.0, g, 4.0, 64.0) for _ in range(B)]).to(device)
    out_holder = {}
    ms = ktime("pairwise_iou", lambda: out_holder.__setitem__("o", ops.pairwise_iou(a, b)))
    add("pairwise_iou_cols_kernel", ms, B * (16 * (N + M) + 4 * N * M), f"B={B} N={N} M={M}", tkey="pairwise_iou_cols_kernel")
    out_holder.clear()
    ms = timeit(lambda: ops.iou_assign(a, b))
    add("iou_assign_kernel(+finalize)", ms, B * (16 * (N + M) + 8 * N), f"B={B} N={N} M={M}", "valu")
    # config 2: COCO-like M=40 at 640^2 A=9 (fused assignment)
    B2, N2, M2 = 16, 34100 * 9, 40
    a2 = syn.random_boxes(N2, 640.0, g).to(device)
    b2 = torch.stack([syn.random_boxes(M2, 640.0, g) for _ in range(B2)]).to(device)
    ms = timeit(lambda: ops.iou_assign(a2, b2))
    add("iou_assign_kernel(+finalize)", ms, B2 * (16 * (N2 + M2) + 8 * N2), f"B={B2} N={N2} M={M2}", "valu")
    ms = ktime("pairwise_iou", lambda: out_holder.__setitem__("o", ops.pairwise_iou(a2, b2)))
    add("pairwise_iou_rows_kernel (M / 4 <= 32)", ms, B2 * (16 * (N2 + M2) + 4 * N2 * M2), f"B={B2} N={N2} M={M2}")
    out_holder.clear()
    # YOLOv5 decode bs16 640^2 C=80
    views, strides, anchors = _yolo_views(16, 640, 85, device)
    ms = timeit(lambda: ops.yolo_decode(views, strides, anchors))
    n = sum(v.shape[1] * v.shape[2] * v.shape[3] for v in views)
    add("yolo_decode_kernel", ms, 2 * 16 * n * 85 * 4, "B=16 640x640 C=80")
    dec = ops.yolo_decode(views, strides, anchors)
    ms = timeit(lambda: ops.yolo_nms(dec, 0.25, 0.45, 30000, 300))
    add("yolo_nms (cand+sort+mask+scan+emit)", ms, 16 * n * 85 * 4,
        "B=16 n=25200 conf=0.25 on random logits: ~75% of rows are candidates (stress case)", "latency")
    ms = timeit(lambda: ops.yolo_detect(views, strides, anchors, 0.25, 0.45, 30000, 300))
    add("yolo_detect (raw cand+sort+mask+scan+emit)", ms, 16 * n * 85 * 4,
        "B=16 n=25200 conf=0.25 on random logits, straight from the head outputs (no decoded tensor)", "latency")
    ms = timeit(lambda: ops.yolo_nms(dec, 0.9, 0.45, 30000, 300))
    add("yolo_nms (cand+sort+mask+scan+emit)", ms, 16 * n * 4,
        "B=16 n=25200 conf=0.9: ~15 candidates per image (detector-like pass rate; reads the objectness column)",
        "latency")
    # Mosaic-4 on the batch resident in HBM (the step before the hot path)
    from heltondetection_amd.augment import Mosaic
    mimg = syn.make_images(16, 640, g).to(device).contiguous(memory_format=torch.channels_last)
    mgb, mgc, mmv = [t.to(device) for t in syn.make_targets(16, 640, g, "coco", 80, 40)]
    msrc, mctr, mapl = Mosaic(p=1.0, generator=torch.Generator(device=device).manual_seed(0)).draw(16, 640, device)
    ms = timeit(lambda: ops.mosaic4(mimg, mgb, mgc, mmv, msrc, mctr, mapl))
    add("mosaic4 (image + boxes)", ms, 5 * 16 * 3 * 640 * 640 * 4, "B=16 640x640 p=1 (4 reads + 1 write per element)")
    del mimg
    # WBF at config-5 density: 3 TTA views x ~500 objects of 10 classes per image
    wb0 = torch.stack([syn.random_boxes(500, 1280.0, g, 4.0, 64.0) for _ in range(8)])
    wbx = torch.cat([wb0 + torch.randn(wb0.shape, generator=g) * 1.5 for _ in range(3)], 1).clamp(0, 1280.0).to(device)
    wsc = (torch.rand((8, 1500), generator=g) * 0.7 + 0.3).to(device)
    wlb = torch.randint(0, 10, (8, 500), generator=g, dtype=torch.int32).repeat(1, 3).to(device)
    wmw = torch.ones((8, 1500), device=device)
    ms = timeit(lambda: ops.wbf(wbx, wsc, wlb, wmw, None, 3.0, 0.55, 0.05))
    add("wbf (sort + per-label clustering + sort)", ms, 8 * 1500 * 24, "B=8 K=1500 (3 views x 500 objects, 10 classes)", "latency")
    # NMS at RPN-like sizes
    for Nn in (2000, 8000):
        bx = torch.stack([syn.random_boxes(Nn, 832.0, g, 16.0, 300.0) for _ in range(16)]).to(device)
        sc = torch.rand((16, Nn), generator=g).to(device)
        ms = timeit(lambda: ops.nms_batched(bx, sc, None, None, -1.0, 0.7, 0, 1000))
        add("nms_batched (sort+mask+scan)", ms, 16 * 28 * Nn, f"B=16 N={Nn} thr=0.7 max_out=1000", "latency")
    # RoIAlign config 3: 832^2 bs16, P2-P5 channels-last, 512 RoIs per image (forward, the inference-side count) and
    # 128 per image (the train step's sampled RoIs, forward + backward), float32 and bfloat16 maps.
    # Algorithmic bytes: the output (or incoming gradient) tensor + every feature byte under at least one RoI,
    # bounded above by the four maps (forward reads them, backward read-modify-writes the f32 gradient maps).
    for dt, dn in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        feats = [torch.randn((16, 256, 832 // s, 832 // s), device=device).to(dt).contiguous(memory_format=torch.channels_last)
                 for s in (4, 8, 16, 32)]
        map_bytes = sum(f.numel() * f.element_size() for f in feats)
        for per_img in (512, 128):
            R = 16 * per_img
            rb = syn.random_boxes(R, 832.0, g, 16.0, 400.0)
            rois = torch.cat([torch.arange(16).repeat_interleave(per_img)[:, None].float(), rb], 1).to(device)
            esz = 4 if dt == torch.float32 else 2
            touched = roi_touched_elements(rois, feats, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 2, 4)
            ms = ktime("roi_align_fwd", lambda: ops.roi_align(feats, rois, 7, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 2, True, 2, 224.0, 4, True))
            add(f"roi_align_fwd_cl_kernel<{dn}>", ms, R * 256 * 49 * esz + touched * esz,
                f"R={R} C=256 P=7 s=2 {dn} maps -> {dn} out (out bytes + the {touched * esz / 1e6:.0f} MB of feature bytes under the RoIs, once)",
                tkey=f"roi_align_fwd_cl_kernel<{dn}> R={R}")
            if per_img == 128:
                fr = [f.detach().requires_grad_(True) for f in feats]
                out = ops.roi_align(fr, rois, 7, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 2, True, 2, 224.0, 4, True)
                go = torch.randn_like(out)
                ops.enable_timing("roi_align_bwd")
                for _ in range(8):
                    out.backward(go, retain_graph=True)
                torch.cuda.synchronize()
                tb = sorted(ops.collect_timing()["roi_align_bwd"])
                ops.disable_timing()
                add("roi_align_bwd_gather_kernel", tb[len(tb) // 2], R * 256 * 49 * esz + map_bytes,
                    f"R={R} C=256 P=7 s=2 grad {dn} -> {dn} maps (kernel only; no atomics: bytes = grad in + every map cell written once)",
                    tkey=f"roi_align_bwd_gather_kernel<{dn}>")
                del fr, out, go
        del feats
    # RPN proposals, config 3: 832^2, 5 levels x 3 anchors, pre 2000 / post 1000
    strides_r = [4, 8, 16, 32, 64]
    hw = [(math.ceil(832 / s), math.ceil(832 / s)) for s in strides_r]
    base = torch.stack([ops.anchor_base([8.0 * s], [0.5, 1.0, 2.0]) for s in strides_r])
    anchors_r = ops.anchor_grid(hw, strides_r, base, 0.5, device)
    ls = [h * w * 3 for h, w in hw]
    NA = anchors_r.shape[0]
    lg = torch.randn((16, NA), generator=g).to(device)
    dl = (torch.randn((16, NA, 4), generator=g) * 0.2).to(device)
    ms = timeit(lambda: ops.rpn_proposals(lg, dl, anchors_r, ls, 2000, 1000, 0.7, 0.0, 832.0, 832.0))
    add("rpn_proposals (topk+decode+sort+mask+scan+emit)", ms, 16 * NA * 20, f"B=16 NA={NA} pre=2000 post=1000", "latency")
    return res


def _hbm_entry(kernel, ms, nbytes, launches, extra=None, traffic=None):
    gbs = nbytes / ms / 1e6 if ms > 0 else 0.0
    d = {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(gbs / HBM_PEAK_GBS, 4), "frac_of_measured_copy_6.29TB/s": round(gbs / HBM_COPY_GBS, 4),
         "avg_launch_ms": round(ms / max(launches, 1), 5), "algorithmic_bytes_per_launch": int(nbytes / max(launches, 1)),
         "launches_timed": launches, "traffic": traffic, "traffic_source": TRAFFIC_SOURCE}
    if extra:
        d.update(extra)
    return d


def _conv_kernel_entry(name, ms, nbytes, calls, flops):
    """One kernel of the convolution family against ITS bound: the 1x1 kernels stream their operands once (HBM), the 3x3
    kernels are implicit GEMMs whose 9x operand reuse puts them under the MFMA roof (dense bf16 peak, not the sparse one)."""
    if ms <= 0:
        return {"sum_ms": 0.0, "calls": calls}
    gbs, tfs = nbytes / ms / 1e6, flops / ms / 1e9
    mfma = name.startswith("conv3x3") and not name.startswith("conv3x3s2")   # the first stride-2 layer (32 / 64 channels) streams
    return {"sum_ms": round(ms, 4), "calls": calls, "bound": "mfma" if mfma else "hbm",
            "GB/s": round(gbs, 1), "frac_hbm": round(gbs / HBM_PEAK_GBS, 4),
            "TFLOP/s": round(tfs, 1), "frac_mfma": round(tfs / MFMA_PEAK_TFLOPS, 4),
            "frac": round(tfs / MFMA_PEAK_TFLOPS, 4) if mfma else round(gbs / HBM_PEAK_GBS, 4)}


def bn_family_roofline(model, batch, args, device, trainer=None):
    """Step-weighted roofline of the fused BN+act kernels, measured IN the train step: three eager forward + loss +
    backward passes of this workload with every fused-BN launch stamping its own HIP event pair (kernel begin / end,
    hd_profile_time_next_launch), so each kernel is timed on its real tensors in its real cache state (x fresh from the
    convolution for stats / apply, x cold and dy fresh for the backward pair) — what a profiler reports for the same
    launches inside the replayed graph.  achieved = algorithmic bytes of all launches of a step / their summed time;
    the ops note each launch's algorithmic bytes and shape themselves (ops._note_bytes), in launch order."""
    from heltondetection_amd import ops
    names = ("bn_stats_kernel", "bn_apply_kernel", "bn_bwd_reduce_kernel", "bn_bwd_dx_kernel")
    pw_names = ("pointwise_conv", "pointwise_dgrad", "pointwise_wgrad", "pointwise_bwd", "conv3x3_fwd", "conv3x3_dgrad", "conv3x3_wgrad",
                "conv3x3s2_fwd", "conv3x3s2_dgrad", "conv3x3s2_wgrad", "stem_conv_fwd", "stem_conv_wgrad", "head_conv", "head_conv_bwd", "wgrad_reduce")
    passes, per_pass, notes, tags = 3, [], None, None
    try:
        for it in range(passes + 1):  # the first pass warms up
            model.zero_grad(set_to_none=True)
            ops.bn_arena_reset(device)
            if it >= 1:
                ops.enable_timing(*names, *pw_names)
            if trainer is not None:
                # exactly the step's forward + loss + backward (bf16 weight shadows, slice sums riding in the next
                # weight-gradient launch), run eagerly so that every launch can stamp its event pair
                trainer._fwd_bwd(*batch)
            else:
                with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.fp32):
                    loss, _ = model.forward_train(*batch)
                loss.backward()
                ops.flush_wgrad()
            torch.cuda.synchronize()
            if it >= 1:
                per_pass.append({k: v for k, v in ops.collect_timing().items()})
                notes, tags = ops.collect_bytes(), ops.collect_tags()
                ops.disable_timing()
    finally:
        ops.disable_timing()
        model.zero_grad(set_to_none=True)
    esz = 4 if args.fp32 else 2
    # algorithmic tensor passes: stats 1 (only where the convolution's epilogue did not take them), apply 2 (+1 with a
    # fused residual read), bwd_reduce 2 (+1), bwd_dx 3 (+2: residual in, d(residual) out)
    bytes_k = {k: sum(notes[k]) for k in names}
    med = lambda v: sorted(v)[len(v) // 2]
    ms_k = {k: med([sum(p[k]) for p in per_pass]) for k in names}
    n_k = {k: len(per_pass[0][k]) for k in names}
    tot_ms, tot_bytes, n_launch = sum(ms_k.values()), sum(bytes_k.values()), sum(n_k.values())
    n_layers = n_k["bn_apply_kernel"]
    # per layer shape: the launch times of the middle pass, paired with the tags noted at the same launches
    mid = per_pass[len(per_pass) // 2]
    by_shape = {}
    for k in names:
        for t_ms, tag in zip(mid[k], tags[k]):
            e = by_shape.setdefault(tag, {"n": {q: 0 for q in names}, "us": {q: 0.0 for q in names}})
            e["n"][k] += 1
            e["us"][k] += t_ms * 1e3
    sizes = sorted(int(torch.tensor(tag[0]).prod().item()) * esz for tag in by_shape)
    fam = _hbm_entry("fused BN+act family (bn_stats / bn_apply / bn_bwd_reduce / bn_bwd_dx), step-weighted over the "
                     f"model's {n_layers} BatchNorm layers", tot_ms, tot_bytes, n_launch,
                     {"bn_layers": n_layers, "stats_in_conv_epilogue_layers": n_layers - n_k["bn_stats_kernel"],
                      "launches_per_step": n_launch, "sum_launch_ms_per_step": round(tot_ms, 4),
                      "algorithmic_bytes_per_step": int(tot_bytes), "distinct_layer_shapes": len(by_shape),
                      "layer_bytes_min_max": [sizes[0], sizes[-1]] if sizes else None,
                      "timed": "inside 3 eager forward+loss+backward passes of this workload right after the timed region: every "
                               "fused-BN launch stamps its own HIP event pair with the kernel's begin / end (hipExtLaunchKernel "
                               "through hd_profile_time_next_launch) on the launch stream; per kernel the summed time of a pass, "
                               "median over the passes",
                      "per_kernel": {k: {"sum_ms": round(ms_k[k], 4), "GB/s": round(bytes_k[k] / ms_k[k] / 1e6, 1),
                                         "frac": round(bytes_k[k] / ms_k[k] / 1e6 / HBM_PEAK_GBS, 4), "launches": n_k[k]}
                                     for k in names},
                      "per_shape": [{"shape": list(tag[0]), "act": {0: None, 1: "silu", 2: "relu"}.get(tag[1], tag[1]),
                                     "residual_mode": tag[2], "stats_in_conv_epilogue": bool(tag[3]),
                                     "layers": e["n"]["bn_apply_kernel"],
                                     "MB": round(int(torch.tensor(tag[0]).prod().item()) * esz / 1e6, 2),
                                     "us_per_launch": {k.replace("bn_", "").replace("_kernel", ""):
                                                       (round(e["us"][k] / e["n"][k], 2) if e["n"][k] else None) for k in names}}
                                    for tag, e in by_shape.items()]},
                     traffic=pmc_traffic("fused BN+act") if (args.model == "yolov5" and args.size == 640 and args.batch == 16
                                                             and args.variant == "s" and not args.fp32) else None)
    # the hand-written 1x1 convolution kernels of the same passes (csrc/pointwise_conv.hip), step-weighted likewise
    pw = None
    if any(len(per_pass[0].get(k, ())) for k in pw_names):
        pms = {k: med([sum(p[k]) for p in per_pass]) for k in pw_names}
        pby = {k: sum(notes[k]) for k in pw_names}
        pn = {k: len(per_pass[0][k]) for k in pw_names}
        pw = _hbm_entry("hand-written convolution family (1x1: pointwise_conv fwd + BN statistics / pointwise_bwd = data gradient + "
                        "weight-gradient partials in one launch / pointwise_dgrad / pointwise_wgrad where that does not apply; 3x3: conv3x3_fwd + BN statistics / conv3x3_dgrad / conv3x3_wgrad + slice sum; first stride-2 3x3: conv3x3s2_fwd + BN statistics / conv3x3s2_dgrad / conv3x3s2_wgrad + slab sum; 6x6 stem: "
                        "stem_conv_fwd + BN statistics / stem_conv_wgrad; Detect: head_conv (float32 padded rows) / head_conv_bwd), step-weighted over the model's layers that take them",
                        sum(pms.values()), sum(pby.values()),
                        sum(pn.values()),
                        {"sum_launch_ms_per_step": round(sum(pms.values()), 4), "algorithmic_bytes_per_step": int(sum(pby.values())),
                         "algorithmic_bytes": "2 (R (Cin + Cout) + taps Cin Cout) per call: each operand once, bf16 (the 3x3 "
                                              "kernels are MFMA / LDS bound, their HBM fraction is reported for completeness)",
                         "timed": "same passes and event mechanism as the BN family (kernel begin / end per launch); the weight-gradient "
                                  "launches carry the slice sum of the previous one as in the Trainer's step, `wgrad_reduce` = the "
                                  "few slice sums launched on their own",
                         "per_kernel": {k: _conv_kernel_entry(k, pms[k], pby[k], pn[k], sum((t or {}).get("flops", 0) for t in tags[k]))
                                        for k in pw_names}})
    if pw is not None and getattr(args, "layer_detail", ""):
        with open(args.layer_detail, "w") as f:
            for k in pw_names:
                for t_ms, nb, tag in zip(mid.get(k, ()), notes.get(k, ()), tags.get(k, ())):
                    t_ = tag or {}
                    f.write(json.dumps({"kernel": k, "shape": t_.get("shape"), "us": round(t_ms * 1e3, 2), "MB": round(nb / 1e6, 2),
                                        "GB/s": round(nb / t_ms / 1e6, 1), "TFLOP/s": round(t_.get("flops", 0) / t_ms / 1e9, 1)}) + "\n")
    if pw is not None and args.model == "yolov5" and args.size == 640 and args.batch == 16 and args.variant == "s" and not args.fp32:
        pw["traffic"] = pmc_traffic("hand-written convolutions")
    for f_ in (fam, pw):
        if f_ is not None:
            f_["traffic_unit"] = "bytes per train step (compare with algorithmic_bytes_per_step)"
    fam["_pointwise"] = pw
    return fam


MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (spec)


def family_rooflines(model, batch, args, device, trainer=None):
    """The hand-written kernel families of the step, largest summed launch time first: element 0 is the line's `roofline`
    (the DOMINANT family), the rest ride in `roofline_secondary`.  Every kernel of a family is also priced against its own
    bound in `per_kernel`: HBM bytes for the streaming kernels (1x1 convolutions, fused BN), MFMA FLOPs / 2.5 PFLOP/s for
    the 3x3 implicit-GEMM kernels."""
    fam = bn_family_roofline(model, batch, args, device, trainer)
    pw = fam.pop("_pointwise", None)
    fams = [f for f in (fam, pw) if f]
    fams.sort(key=lambda f: -float(f.get("sum_launch_ms_per_step", 0.0)))
    if fams:
        fams[0]["dominant"] = True
        fams[0]["family_sum_ms"] = {f["kernel"].split(" (")[0].split(",")[0][:48]: f.get("sum_launch_ms_per_step") for f in fams}
    return fams


def yolo_dense_grad_roofline(model, batch, args):
    """The dense pass of the loss backward in the form the measured step runs it: with bf16 weight shadows (the default) the
    gradient leaves as bf16 rows padded to the Detect convolution's pitch (hd_yolo_loss_bwd_dense_bf16: 2 * B * pixels * pitch
    bytes written + the 4-byte objectness residual per cell read); --fp32: float32 rows of the predictions' own layout."""
    from heltondetection_amd import ops
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.fp32):
        preds = model.head(model.extract(batch[0]))
    head = model.head
    shapes = [(head.na, p.shape[2], p.shape[3], head.no) for p in preds]
    slots = head.assigner(shapes, head.strides, head.anchors, batch[1], batch[2], batch[3])
    pitch = ops.head_conv_pitch(head.na * head.no)
    lowp = (not args.fp32) and ops.HeadConvSwitch.enabled and ops.HeadConvSwitch.bf16_grad and pitch > 0 and pitch % 8 == 0
    ms_list = ops.time_yolo_loss_bwd_dense(preds, head.strides, head.anchors, slots, batch[2].shape[1],
                                           (head.box_gain, head.obj_gain, head.cls_gain), head.balance, iters=20, bf16_rows=lowp)
    k_ms = sum(ms_list) / max(len(ms_list), 1)
    if lowp:
        cells = sum(head.na * p.shape[2] * p.shape[3] for p in preds) * args.batch
        nbytes = 2 * (cells // head.na) * pitch + 4 * cells
        name = "yolo_grad_dense_bf16_kernel (hd_yolo_loss_bwd_dense_bf16: bf16 rows padded to %d channels)" % pitch
    else:
        nbytes = loss_bwd_bytes(args.batch, args.size, args.classes + 5)
        name = "yolo_grad_dense_kernel (hd_yolo_loss_bwd_dense)"
    return _hbm_entry(name, k_ms * len(ms_list), nbytes * len(ms_list), len(ms_list),
                      {"timed": "every launch stamps its own HIP event pair (kernel begin / end) on this workload's head outputs"},
                      traffic=pmc_traffic("yolo_grad_dense_bf16_kernel" if lowp else "yolo_grad_dense_kernel")
                      if (args.size == 640 and args.batch == 16 and args.classes == 80) else None)


def roi_touched_elements(rois5, feats, scales, k_min, n_levels):
    """Feature pixels x channels under at least one RoI's sampling window, summed over levels: what a perfect gather
    would have to read once (forward) or read-modify-write once (backward).  2-D difference array per level, on the
    device: +1 / -1 at the window corners, two prefix sums, count of cells > 0."""
    from heltondetection_amd import ops
    lvl = ops.roi_level_map(rois5, k_min, n_levels, 224.0, 4).long()
    total = 0
    for l, f in enumerate(feats[:n_levels]):
        B, C, H, W = f.shape
        sel = lvl == l
        if not bool(sel.any()):
            continue
        r = rois5[sel]
        b = r[:, 0].long()
        x0 = (r[:, 1] * scales[l] - 0.5).floor().clamp(0, W - 1).long()
        y0 = (r[:, 2] * scales[l] - 0.5).floor().clamp(0, H - 1).long()
        x1 = ((r[:, 3] * scales[l] - 0.5).floor() + 1).clamp(0, W - 1).long()
        y1 = ((r[:, 4] * scales[l] - 0.5).floor() + 1).clamp(0, H - 1).long()
        d = torch.zeros((B, H + 1, W + 1), dtype=torch.int32, device=f.device)
        one = torch.ones_like(b, dtype=torch.int32)
        d.index_put_((b, y0, x0), one, accumulate=True)
        d.index_put_((b, y0, x1 + 1), -one, accumulate=True)
        d.index_put_((b, y1 + 1, x0), -one, accumulate=True)
        d.index_put_((b, y1 + 1, x1 + 1), one, accumulate=True)
        cover = d.cumsum(1).cumsum(2)[:, :H, :W] > 0
        total += int(cover.sum()) * C
    return total


def roi_align_roofline(model, batch, args, device):
    """FasterRCNN (config 3): hd_roi_align_fwd / _bwd on the step's own pyramid (bf16 under autocast) and sampled RoIs.
    Algorithmic bytes ("R*C*P^2*4 out + touched feature bytes"): fwd = RoI features out + every
    feature byte under at least one RoI window once; bwd = incoming gradient + the same cells of the float32 gradient
    maps read-modify-written once (what a perfect scatter would move; the kernel itself is float-atomic bound)."""
    from heltondetection_amd import ops
    images, gt_boxes, gt_cls, m_valid = batch
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.fp32):
        feats = model.extract(images)
        logits, deltas = model.rpn_head(feats)
        anchors, level_sizes = model.rpn_head.anchors_for(feats)
        rois, _, count = model.rpn_head.proposals(logits, deltas, anchors, level_sizes, images.shape[-2:], True)
        rois5 = model.roi_head.sample(rois, count, gt_boxes, gt_cls, m_valid)[0]
    rh = model.roi_head
    fs = [f.detach().requires_grad_(True) for f in feats[: len(rh.scales)]]
    ops.enable_timing("roi_align_fwd", "roi_align_bwd")
    outs = None
    for _ in range(12):
        outs = ops.roi_align(fs, rois5, rh.P, rh.scales[: len(fs)], rh.sr, True, rh.k_min, 224.0, 4, channels_last_out=rh.decoupled)
        outs.backward(torch.ones_like(outs))
    torch.cuda.synchronize()
    t = ops.collect_timing()
    ops.disable_timing()
    med = lambda v: sorted(v)[len(v) // 2]
    f_ms, b_ms = med(t["roi_align_fwd"]), med(t["roi_align_bwd"])
    R, Cc = rois5.shape[0], fs[0].shape[1]
    esz = fs[0].element_size()
    touched = roi_touched_elements(rois5, fs, rh.scales, rh.k_min, len(fs))
    f_bytes = R * Cc * rh.P * rh.P * esz + touched * esz
    # backward = the gather (csrc/roi_align.hip): the incoming gradient once + every cell of the gradient maps WRITTEN once in
    # the features' dtype (no float32 accumulators, no zero-fill, no cast)
    map_elems = int(sum(f.numel() for f in fs))
    b_bytes = R * Cc * rh.P * rh.P * esz + map_elems * esz
    shape = f"R={R} C={Cc} P={rh.P} s={rh.sr} {args.size} bs{args.batch} {str(fs[0].dtype).replace('torch.', '')}"
    main = _hbm_entry("roi_align_fwd_cl_kernel + roi_align_bwd_gather_kernel (hd_roi_align_fwd / hd_roi_align_bwd_gather), one launch each per step",
                      f_ms + b_ms, f_bytes + b_bytes, 2,
                      {"shape": shape, "touched_feature_elements": int(touched), "pyramid_elements": int(sum(f.numel() for f in fs)),
                       "timed": "every launch stamps its own HIP event pair (kernel begin / end) on the step's own pyramid and sampled RoIs",
                       "per_kernel": {"roi_align_fwd_cl_kernel": {"ms": round(f_ms, 4), "GB/s": round(f_bytes / f_ms / 1e6, 1),
                                                                  "frac": round(f_bytes / f_ms / 1e6 / HBM_PEAK_GBS, 4)},
                                      "roi_align_bwd_gather_kernel": {"ms": round(b_ms, 4), "GB/s": round(b_bytes / b_ms / 1e6, 1),
                                                                      "frac": round(b_bytes / b_ms / 1e6 / HBM_PEAK_GBS, 4),
                                                                      "bound": "hbm write (every map cell once) + per-RoI tile work",
                                                                      "algorithmic_bytes": int(b_bytes)}}},
                      traffic=None)   # the counter run prices the two kernels on the micro leg's random RoIs (`kernels`), not these
    return main


def cpu_model_string() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine()


def run_leg(args, device, rank, world, log, steps, warmup, tag=""):
    """One measured configuration under the bench contract: build the model + Trainer, `warmup` untimed steps (MIOpen's
    solver search and the hipGraph capture happen there), then EXACTLY `steps` steps between barrier + synchronise pairs,
    max over ranks.  The two hipGraph guards (finite loss; a replay reproduces the eager loss) re-time eagerly when they
    trip.  Returns the numbers plus the live objects the roofline legs measure on."""
    from heltondetection_amd import ops
    from heltondetection_amd.detectors import build_detector
    from heltondetection_amd.train import Trainer

    model = build_detector(model_cfg(args)).to(device).to(memory_format=torch.channels_last).train()
    # the hipGraph is captured after `graph_warmup` eager steps (the first one runs MIOpen's solver search, which
    # cannot happen inside a capture); keep that inside the untimed warm-up whatever --warmup is
    graph_warmup = min(3, max(1, warmup - 1))
    lr = 1e-3 if args.model == "yolov5" else 2e-4
    trainer = Trainer(model, lr=lr, amp_dtype=None if args.fp32 else torch.bfloat16,
                      total_steps=steps + warmup + 8, warmup_steps=3, device=device,
                      use_graph=not args.no_graph, graph_warmup=graph_warmup,
                      graph_sync={"auto": None, "on": True, "off": False}[args.graph_sync],
                      graph_split={"auto": True, "on": True, "off": False}[args.graph_split])
    batches = make_batches(4 if args.model == "yolov5" else 2, args.batch, args.size, args.classes, 1234 + rank, device,
                           getattr(args, "style", "coco"))

    log(f"{tag}{model_name(args)} {args.size}^2 bs{args.batch} {'fp32' if args.fp32 else 'bf16'} + data ready on "
        f"{torch.cuda.get_device_name(device)}; warm-up {warmup} steps (first steps include MIOpen kernel selection/compilation)")
    eager_ms = []  # steady eager steps seen before the graph takes over (step 0 is MIOpen's search)
    for i in range(warmup):
        tw = time.perf_counter()
        trainer.step(*batches[i % len(batches)])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - tw) * 1e3
        if i >= 1 and trainer._graph is None:
            eager_ms.append(dt)
        log(f"{tag}warm-up step {i}: {dt:.1f} ms")
    # --warmup too small for the capture to have happened: finish it as untimed set-up (reported in config), so the
    # K timed steps are K steady-state steps and not (K-1) steps plus a graph capture
    extra_setup = 0
    while trainer.use_graph and trainer._graph is None and extra_setup < graph_warmup + 2:
        trainer.step(*batches[extra_setup % len(batches)])
        extra_setup += 1
    if extra_setup:
        log(f"{tag}{extra_setup} extra untimed set-up step(s) to finish the hipGraph capture (--warmup {warmup})")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()

    def timed_region():
        graph_on_ = trainer._graph is not None
        torch.cuda.synchronize()
        t0_ = time.perf_counter()
        for i in range(steps):
            loss_, parts_ = trainer.step(*batches[i % len(batches)])
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0_, loss_, graph_on_

    elapsed, loss, graph_on = timed_region()
    ok = torch.isfinite(loss).all().to(torch.int32).reshape(1)
    if graph_on and bool(ok.item()) and trainer._segments is None:
        # second guard: one more replay (no optimizer step) must reproduce what an eager forward of the same
        # weights computes on the same batch (bf16 convs + atomics: agree to ~1e-2, corruption is off by >>)
        for dst, src in zip(trainer._static, batches[0]):
            dst.copy_(src)
        trainer._graph.replay()
        torch.cuda.synchronize()
        l_graph = float(trainer._static_out[0])
        if args.model == "yolov5":  # FasterRCNN's loss samples RoIs at random: two evaluations are not comparable
            # the eager evaluation runs the step's OWN forward numerics (bf16 weight shadows on, hence the HIP Detect convolution
            # with its unrounded float32 logits): with the shadows off the head takes the library path, whose logits are rounded
            # to bf16 — on a converged model (loss ~0.1 after 3000 steps on two batches) that alone moved the loss by 6 % and
            # tripped this guard in one of four soak runs, re-timing a healthy run eagerly
            from heltondetection_amd.layers import LowpWeights
            was_lowp = LowpWeights.on
            LowpWeights.on = getattr(trainer, "shadows", None) is not None
            try:
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.fp32):
                    l_eager = float(model.forward_train(*batches[0])[0])
            finally:
                LowpWeights.on = was_lowp
            if not (abs(l_graph - l_eager) <= 0.05 * abs(l_eager) + 1e-3):
                log(f"{tag}graph replay loss {l_graph} != eager loss {l_eager}")
                ok.zero_()
            else:
                log(f"{tag}graph replay loss {l_graph:.6g} == eager loss {l_eager:.6g} (within 5 %)")
        elif not (l_graph == l_graph and abs(l_graph) < 1e4):
            ok.zero_()
    slow_replay = bool(graph_on and eager_ms and elapsed / steps * 1e3 > 1.5 * min(eager_ms))
    if slow_replay:
        # NOT a fallback: a stall between the replay and the collectives would show up exactly like this
        # (DESIGN.md section 9); the slow number is reported as measured and flagged so a recurrence cannot hide
        log(f"{tag}ERROR: graph-mode step {elapsed / steps * 1e3:.1f} ms is slower than the eager steps it replaces "
            f"({min(eager_ms):.1f} ms); reporting it as measured with config.hip_graph_slow_replay = true")
    if world > 1:  # every rank must take the same branch
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if graph_on and not bool(ok.item()):
        # safety net for the hipGraph path: never report a number from a corrupted run
        log(f"{tag}hipGraph guard tripped (non-finite / mismatching loss): rebuilding the trainer and re-timing eagerly")
        ops.disable_timing()
        model = build_detector(model_cfg(args)).to(device).to(memory_format=torch.channels_last).train()
        trainer = Trainer(model, lr=lr, amp_dtype=None if args.fp32 else torch.bfloat16,
                          total_steps=steps + warmup + 1, warmup_steps=3, device=device, use_graph=False)
        trainer.graph_error = "hipGraph guard tripped (loss check); re-timed eagerly"
        for i in range(warmup):
            trainer.step(*batches[i % len(batches)])
        elapsed, loss, graph_on = timed_region()
        slow_replay = False
    ops.disable_timing()
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return {"elapsed": elapsed, "loss": float(loss.item()), "model": model, "trainer": trainer, "batches": batches,
            "slow_replay": slow_replay, "eager_ms": eager_ms, "extra_setup": extra_setup, "steps": steps, "warmup": warmup}


COMPACT_LIMIT = 4096   # bytes: the driver keeps only a tail of stdout + stderr, a 35 KB line lost the round-4 record


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 1] + "\u2026"


def _compact_roofline(r):
    """The keys of a roofline object the bench contract names (+ what locates the measurement), strings cut short."""
    if not isinstance(r, dict):
        return None
    keep = ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "launches_timed", "algorithmic_bytes_per_launch",
            "algorithmic_bytes_per_step", "sum_launch_ms_per_step", "traffic")
    d = {"kernel": _short(r.get("kernel", ""), 72)}
    d.update({k: r[k] for k in keep if k in r})
    pk = r.get("per_kernel")
    if isinstance(pk, dict):   # every kernel of the family against its own bound: [fraction, summed ms per step]
        d["per_kernel_frac_ms"] = {k: [v.get("frac"), v.get("sum_ms", v.get("ms"))] for k, v in pk.items()
                                   if isinstance(v, dict) and (v.get("sum_ms") or v.get("ms"))}
    return d


def compact_line(out, detail_path=None):
    """The ONE stdout line of the bench contract, at most COMPACT_LIMIT bytes: the contract's keys, the headline `roofline` and
    `cpu_baseline`, one short entry per secondary roofline and per companion leg.  Everything else (per-shape tables, the
    micro-kernel list, the legs' full rooflines) goes to the detail file named in `detail`."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
           "dtype", "data")
    c = {k: out[k] for k in top if k in out}
    cfg = dict(out.get("config", {}))
    cfg["workload"] = _short(cfg.get("workload", ""), 160)
    c["config"] = cfg
    for k in ("final_loss", "ms_per_step_packed", "eff_vs_packed_n1", "roofline_error", "kernels_error"):
        if k in out:
            c[k] = out[k] if not isinstance(out[k], str) else _short(out[k], 120)
    if "roofline" in out:
        c["roofline"] = _compact_roofline(out["roofline"])
    sec = out.get("roofline_secondary") or []
    if sec:
        c["roofline_secondary"] = [{"kernel": _short(r.get("kernel", ""), 40), "bound": r.get("bound"), "frac": r.get("frac"),
                                    "achieved": r.get("achieved"), "unit": r.get("unit"),
                                    "ms": r.get("sum_launch_ms_per_step", r.get("avg_launch_ms"))} for r in sec if isinstance(r, dict)]
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c["cpu_baseline"] = {k: (_short(v, 150) if isinstance(v, str) else v) for k, v in cb.items()
                             if k in ("value", "unit", "cores", "kind", "cpu_model", "threads", "sample", "loadavg_1min_before")}
    legs = {}
    for name in LEG_NAMES:
        l = out.get(name)
        if not isinstance(l, dict):
            continue
        e = {k: l[k] for k in ("ms_per_step", "value", "dtype", "hip_graph") if k in l}
        if "skipped" in l or "error" in l:
            e["note"] = _short(l.get("skipped", l.get("error")), 80)
        if isinstance(l.get("roofline"), dict):
            e["roofline_frac"] = l["roofline"].get("frac")
            e["roofline_kernel"] = _short(l["roofline"].get("kernel", ""), 32)
        legs[name] = e
    if legs:
        c["legs"] = legs
    if detail_path:
        c["detail"] = detail_path
    line = json.dumps(c, separators=(",", ":"))
    # never exceed the limit: drop the optional parts, least important first
    for victim in ("roofline_secondary", "legs"):
        if len(line) <= COMPACT_LIMIT:
            break
        c.pop(victim, None)
        line = json.dumps(c, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT and isinstance(c.get("roofline"), dict):
        c["roofline"].pop("per_kernel_frac_ms", None)
        line = json.dumps(c, separators=(",", ":"))
    return line


LEG_NAMES = ("packed", "fp32", "size_832", "frcnn_832", "frcnn_1280_visdrone")


def write_detail(out):
    """Full record (per-shape tables, micro kernels, the legs' rooflines) next to bench.py and, on a gpurun box, under gpurun_out/
    so that it travels back; returns the path written into the compact line."""
    name = "bench_detail.json"
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, name), "w") as f:
                    json.dump(out, f, indent=1)
        except OSError:
            pass
    return name


T_START = time.perf_counter()


def main():
    args = parse()
    ensure_world(args)  # before the first GPU call
    from heltondetection_amd import ops
    from heltondetection_amd.dist import init_distributed

    rank, local, world = init_distributed()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    rehearsal = os.environ.get("HD_DIST_BACKEND") == "gloo"
    if rehearsal:  # every rank on GPU 0
        local = 0
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    torch.manual_seed(0)  # identical init on every rank (also broadcast by GradReducer)
    torch.backends.cudnn.benchmark = not args.no_miopen_find  # MIOpen find: +20% steady state, ~80 s first step
    backend = dist.get_backend() if world > 1 else None
    n_devices = 1 if (rehearsal or world == 1) else world  # one device per rank unless this is the one-GPU rehearsal

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    shipped_find_db = False
    if not args.no_miopen_find:
        # the library convolutions' solver search of one MI355X run is shipped (heltondetection_amd/miopen_db, tuning.py): the
        # same library kernels every time instead of whatever a fresh search's first-call timings favour, and no search
        from heltondetection_amd import tuning
        shipped_find_db = tuning.use_shipped_find_db()
        log(f"MIOpen user find-db: {'shipped (heltondetection_amd/miopen_db)' if shipped_find_db else os.environ.get('MIOPEN_USER_DB_PATH', 'default')}")

    if (args.model == "frcnn" or not args.no_legs) and not args.no_miopen_find and rank == 0:
        # MIOpen's solver search for ResNet50 at 832^2 is one long call inside the first step (minutes on a fresh
        # box): a heartbeat keeps the run visibly alive
        import threading

        def _beat():
            t0 = time.time()
            while True:
                time.sleep(90)
                log(f"heartbeat {int(time.time() - t0)} s")
        threading.Thread(target=_beat, daemon=True).start()

    leg = run_leg(args, device, rank, world, log, args.steps, args.warmup)
    elapsed, model, trainer, batches = leg["elapsed"], leg["model"], leg["trainer"], leg["batches"]

    if rank == 0:
        exchange = "no exchange (one rank)" if world == 1 else \
            (f"{backend} all-reduce" + (" = RCCL over xGMI" if backend == "nccl" else " (REHEARSAL: ranks time-slice one GPU)"))
        out = {
            "metric": f"images/sec train step, {model_name(args)} {args.size}² bs={args.batch}/GPU",
            "value": round(world * args.batch * args.steps / elapsed, 2),
            "unit": "images/s",
            "n_gpus": n_devices,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "fp32" if args.fp32 else "bf16",
            "data": "synthetic",
            "config": {"workload": f"{model_name(args)} {args.size}x{args.size} bs={args.batch}/GPU train step "
                                   f"(fwd + HIP assign/loss/RoI ops + bwd + {exchange} + AdamW), random-init weights",
                       "global_batch": world * args.batch, "img_size": args.size, "num_classes": args.classes,
                       "parallelism": f"dp{world}", "ranks": world, "backend": backend,
                       "miopen_find_db": ("off" if args.no_miopen_find else
                                          ("shipped" if shipped_find_db else os.environ.get("MIOPEN_USER_DB_PATH", "default")))},
            "final_loss": round(leg["loss"], 4),
        }
        if rehearsal and world > 1:
            out["config"]["rehearsal"] = True  # `value` is NOT a multi-GPU figure: `ranks` processes shared one device
        out["config"]["hip_graph"] = trainer._graph is not None
        if getattr(trainer, "_segments", None) is not None:
            out["config"]["hip_graph_segments"] = len(trainer._segments)   # split replay: all-reduce k overlaps segment k + 1
        if world > 1:
            out["config"]["graph_sync"] = bool(trainer.graph_sync)
            # overlapped split replay checks itself once against the drained form (train.Trainer._verify_overlap)
            out["config"]["graph_overlap_check_rel_l2"] = trainer.graph_overlap_check
            if trainer.graph_overlap_error:
                out["config"]["graph_overlap_error"] = trainer.graph_overlap_error
        if leg["slow_replay"]:
            out["config"]["hip_graph_slow_replay"] = True
            out["config"]["eager_ms_per_step_seen_in_warmup"] = round(min(leg["eager_ms"]), 3)
        if leg["extra_setup"]:
            out["config"]["extra_untimed_setup_steps"] = leg["extra_setup"]
        if trainer.graph_error:
            out["config"]["hip_graph_error"] = trainer.graph_error[:200]
        log(f"timed region done: {out['value']} images/s (hipGraph={'on' if trainer._graph is not None else 'off'})")
        # Rooflines are measured live, with HIP events on the launch stream, on this workload's own shapes, right
        # after the timed region (inside it the launches sit in a replayed hipGraph, where no event can be recorded).
        # world > 1: the roofline passes run on rank 0 ALONE, so no collective may be launched from them (an all-reduce the other
        # ranks never join would sit in front of the final barrier for ever: seen as a hang of the 2-rank rehearsal in round 5)
        hold_was = trainer.reducer._hold
        trainer.reducer._hold = trainer.reducer._hold or world > 1
        try:
            if args.no_roofline:
                out["roofline"] = None
            elif args.model == "yolov5":
                fams = family_rooflines(model, batches[0], args, device, trainer)
                out["roofline"], rest = fams[0], fams[1:]
                out["roofline_secondary"] = rest + [yolo_dense_grad_roofline(model, batches[0], args)]
            else:
                fams = family_rooflines(model, batches[0], args, device, trainer)
                out["roofline"] = fams[0]
                out["roofline_secondary"] = fams[1:] + [roi_align_roofline(model, batches[0], args, device)]
        except Exception as e:  # never lose the headline number to a secondary measurement
            out["roofline_error"] = repr(e)[:300]
        finally:
            trainer.reducer._hold = hold_was
        # The CPU baseline comes BEFORE the companion legs: everything the bench contract requires is then in hand and is printed
        # once right away (a later time-limit kill inside a leg cannot lose the headline); the completed line follows at the end.
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args) if args.model == "yolov5" else cpu_baseline_frcnn(args)
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)[:200]}
        if world > 1:
            # scaling efficiency is the driver's to compute; what this line adds is the N = 1 figure of the SAME per-rank program
            # (gradients packed into buckets), if an N = 1 run on this checkout left it behind (bench_n1_packed.json)
            try:
                n1 = json.load(open(os.path.join(ROOT, "bench_n1_packed.json")))
                if n1.get("workload") == out["config"]["workload"].split(" train step")[0]:
                    out["ms_per_step_packed_n1"] = n1["ms_per_step_packed"]
                    out["eff_vs_packed_n1"] = round(n1["ms_per_step_packed"] / out["ms_per_step"], 4)
            except Exception:
                pass
        has_more = world == 1 and (not args.no_micro or not args.no_legs)
        if has_more:
            write_detail(out)
            print(compact_line(out, "bench_detail.json (partial: companion legs still running)"), flush=True)
        if world == 1 and not args.no_micro:
            try:
                out["kernels"] = micro_kernels(device)
                for k in out["kernels"]:  # the IoU kernel north_star singles out rides with the secondary rooflines too
                    if k["kernel"].startswith("pairwise_iou_cols_kernel"):
                        out.setdefault("roofline_secondary", []).append(
                            {"kernel": k["kernel"], "bound": "hbm", "achieved": k["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": k["frac_of_8TB/s"], "shape": k["shape"], "avg_launch_ms": k["ms"],
                             "algorithmic_bytes_per_launch": int(k["algorithmic_MB"] * 1e6),
                             "traffic": pmc_traffic("pairwise_iou_cols_kernel"), "traffic_source": TRAFFIC_SOURCE})
            except Exception as e:  # the headline number must survive a micro-leg problem
                out["kernels_error"] = repr(e)
        # Companion legs (N = 1, default workload only): short runs under the same timing contract, reported inside this line,
        # each with the roofline of ITS dominant hand-written family.  north_star names 640^2 and 832^2; BASELINE.json configs
        # 3 (FasterRCNN 832^2 bs16) and 5 (FasterRCNN 1280^2, VisDrone-style ground truth).  `packed`: the per-rank program of an
        # N > 1 run (gradients packed into the all-reduce buckets), for the scaling efficiency to be quoted against.
        if world == 1 and not args.no_legs and args.model == "yolov5" and not args.fp32 and args.size == 640:
            import copy
            import gc
            del leg, model, trainer, batches
            legs = (("packed", {}, {"HD_DIRECT_GRADS": "0"}, False), ("fp32", {"fp32": True}, {}, False),
                    ("size_832", {"size": 832}, {}, True),
                    ("frcnn_832", {"model": "frcnn", "size": 832, "batch": 16}, {}, True),
                    ("frcnn_1280_visdrone", {"model": "frcnn", "size": 1280, "batch": 8, "style": "visdrone"}, {}, True))
            for name, over, env, want_roof in legs:
                gc.collect()
                torch.cuda.empty_cache()
                spent = time.perf_counter() - T_START
                if spent > args.leg_budget:
                    out[name] = {"skipped": f"wall-clock budget: {spent:.0f} s of --leg-budget {args.leg_budget:.0f} s used"}
                    log(f"[{name}] skipped ({spent:.0f} s used)")
                    continue
                a2 = copy.copy(args)
                for k, v in over.items():
                    setattr(a2, k, v)
                saved = {k: os.environ.get(k) for k in env}
                os.environ.update(env)
                try:
                    frc = a2.model == "frcnn"
                    l2 = run_leg(a2, device, rank, world, log, args.leg_steps, 4 if frc else 5, tag=f"[{name}] ")
                    out[name] = {"ms_per_step": round(l2["elapsed"] / l2["steps"] * 1e3, 3),
                                 "value": round(a2.batch * l2["steps"] / l2["elapsed"], 2), "unit": "images/s",
                                 "steps": l2["steps"], "warmup": l2["warmup"], "dtype": "fp32" if a2.fp32 else "bf16",
                                 "workload": f"{model_name(a2)} {a2.size}x{a2.size} bs={a2.batch}"
                                             + (" VisDrone-style ground truth (400-600 boxes per image)" if a2.style == "visdrone" else "")
                                             + (" gradients packed into the all-reduce buckets (HD_DIRECT_GRADS=0)" if env else ""),
                                 "hip_graph": l2["trainer"]._graph is not None, "final_loss": round(l2["loss"], 4)}
                    if l2["trainer"].graph_error:
                        out[name]["hip_graph_error"] = l2["trainer"].graph_error[:200]
                    if want_roof and not args.no_roofline:
                        try:
                            fams = family_rooflines(l2["model"], l2["batches"][0], a2, device, l2["trainer"])
                            for f_ in fams:
                                f_.pop("per_shape", None)   # the headline leg carries the per-shape detail
                            out[name]["roofline"] = fams[0]
                            out[name]["roofline_secondary"] = fams[1:] + ([roi_align_roofline(l2["model"], l2["batches"][0], a2, device)]
                                                                          if frc else [])
                        except Exception as e:
                            out[name]["roofline_error"] = repr(e)[:300]
                    log(f"[{name}] {out[name]['ms_per_step']} ms/step, {out[name]['value']} images/s")
                    del l2
                except Exception as e:
                    out[name] = {"error": repr(e)[:300]}
                finally:
                    for k, v in saved.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
            if "packed" in out and "ms_per_step" in out["packed"]:
                out["ms_per_step_packed"] = out["packed"]["ms_per_step"]
                out["eff_vs_packed_n1"] = round(out["packed"]["ms_per_step"] / out["ms_per_step"], 4)   # N = 1: direct vs packed
        detail = write_detail(out)
        if world == 1 and "ms_per_step_packed" in out:
            try:
                with open(os.path.join(ROOT, "bench_n1_packed.json"), "w") as f:
                    json.dump({"workload": out["config"]["workload"].split(" train step")[0],
                               "ms_per_step_packed": out["ms_per_step_packed"], "ms_per_step": out["ms_per_step"]}, f)
            except OSError:
                pass
        print(compact_line(out, detail), flush=True)   # the LAST thing this process writes
    if world > 1:
        # a rank that hangs in the teardown must not hang the launcher (and the driver's clock): dump where, then leave
        import faulthandler
        faulthandler.dump_traceback_later(90, exit=True)
        dist.barrier()
        dist.destroy_process_group()
        faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
