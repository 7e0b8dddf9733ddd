"""Multi-rank hipGraph mode (reference README.md:29 DDP; north_star "all-reduce overlapped with the backward pass"):
two fresh processes, both on GPU 0, gloo backend (the only way to run two ranks on a one-GPU box; RCCL needs one
device per rank).  Trainer(use_graph=True) captures the step with bucket markers, cuts it into segments
(ops.GraphSegments) and launches bucket k's all-reduce right after segment k.  Checked: both ranks end with identical
parameters, the split-replayed run matches the eager (hook-launched) 2-rank run, every bucket is reduced exactly once
per step in bucket order, and no replay stalls."""
import os
import socket
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, use_graph):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from heltondetection_amd import synthetic as syn
    from heltondetection_amd.detectors import build_detector
    from heltondetection_amd.dist import init_distributed
    from heltondetection_amd.train import Trainer
    init_distributed("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = False
    model = build_detector(dict(type="YOLOv5", variant="n", num_classes=8, img_size=128)).to(dev)
    model = model.to(memory_format=torch.channels_last).train()
    # 256 KiB buckets: the 1.8 M parameters of the `n` model spread over many buckets -> many segments
    # lr = 0: the parameters never move, so every step of both runs evaluates the SAME function and the reduced
    # gradients can be compared sharply at any step.  (With any positive learning rate, however tiny, AdamW moves each
    # weight by ~lr per step in the direction of a possibly noise-level gradient sign; the YOLO box loss has kinks —
    # CIoU's enclosing-box min/max — and on a 4-image batch a single matched cell crossing one changes a level's
    # gradient by >10 %: two correct runs then differ discretely from some step on.  Seen and pinned down with
    # scratch probes this round: eager runs differ from each other the same way.)
    tr = Trainer(model, lr=0.0, amp_dtype=None, total_steps=40, warmup_steps=2, bucket_bytes=1 << 20, device=dev,
                 use_graph=use_graph, graph_warmup=2)
    assert len(tr.reducer.buckets) >= 3
    launches = []
    real = dist.all_reduce

    def counting(t, *a, **k):
        launches.append(t.data_ptr())
        return real(t, *a, **k)

    g = syn.make_generator(100 + rank)   # each rank its own shard
    img = syn.make_images(4, 128, g).to(dev).contiguous(memory_format=torch.channels_last)
    gtb, gtc, mv = syn.make_targets(4, 128, g, "coco", 8, 8)
    batch = (img, gtb.to(dev), gtc.to(dev), mv.to(dev))
    losses, times = [], []
    want = [b.flat.data_ptr() for b in tr.reducer.buckets]
    for i in range(8):
        launches.clear()
        dist.all_reduce = counting
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        try:
            loss, _ = tr.step(*batch)
        finally:
            dist.all_reduce = real
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        losses.append(float(loss))
        got = [p for p in launches if p in want]
        if tr._graph is None or i >= 3:
            assert sorted(got) == sorted(want) and len(got) == len(want), (i, len(got), len(want))   # exactly once per bucket
        if tr._segments is not None and i >= 3:
            assert got == want, "split replay must launch the collectives in bucket order"
    info = {"graph": tr._graph is not None, "segments": None if tr._segments is None else len(tr._segments),
            "buckets": len(tr.reducer.buckets), "graph_error": tr.graph_error, "losses": losses, "times": times,
            "graph_sync": tr.graph_sync, "params": [p.detach().float().cpu() for p in model.parameters()],
            "grads": [p.grad.detach().float().cpu().clone() for p in model.parameters()]}
    torch.save(info, os.path.join(tmp, f"{'g' if use_graph else 'e'}{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_split_graph_replay_matches_eager(gpu, tmp_path):
    world = 2
    for use_graph in (True, False):
        mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), use_graph), nprocs=world, join=True)
    g0, g1 = torch.load(tmp_path / "g0.pt"), torch.load(tmp_path / "g1.pt")
    e0, e1 = torch.load(tmp_path / "e0.pt"), torch.load(tmp_path / "e1.pt")
    assert g0["graph"] and g1["graph"], (g0["graph_error"], g1["graph_error"])
    # one segment per bucket, plus a tail when something follows the last bucket's packing
    assert g0["segments"] in (g0["buckets"], g0["buckets"] + 1) and g0["segments"] >= 3
    assert g0["graph_sync"] is True          # gloo rehearsal: drained; the RCCL default is off (train.Trainer)
    for a, b in zip(g0["params"], g1["params"]):
        assert torch.equal(a, b)             # both ranks hold identical parameters after the split-replayed steps
    for a, b in zip(e0["params"], e1["params"]):
        assert torch.equal(a, b)
    for a, b in zip(g0["grads"], g1["grads"]):
        assert torch.equal(a, b)             # ... and the same averaged gradients
    # split replay == eager hook-launched DDP: the averaged gradients of the last step (fp32 step; float atomics
    # reorder sums, so not bit-equal)
    num = sum(float(((a - b) ** 2).sum()) for a, b in zip(g0["grads"], e0["grads"]))
    den = sum(float((b ** 2).sum()) for b in e0["grads"])
    worst = sorted(((float(((a - b) ** 2).sum() / ((b ** 2).sum() + 1e-30)) ** 0.5, i, tuple(b.shape))
                    for i, (a, b) in enumerate(zip(g0["grads"], e0["grads"]))), reverse=True)[:6]
    assert den > 0 and (num / den) ** 0.5 < 1e-3, ((num / den) ** 0.5, worst, g0["losses"], e0["losses"])
    assert all(abs(x - y) < 1e-3 * max(abs(y), 1.0) for x, y in zip(g0["losses"], e0["losses"]))
    # no multi-second stall on any replayed step (second hazard)
    assert max(g0["times"][3:]) < 2.0 and max(g1["times"][3:]) < 2.0, (g0["times"], g1["times"])


def _worker_overlap(rank, world, port, tmp, use_graph):
    """bf16 autocast + weight shadows + HdAdamW, graph_sync forced OFF (the RCCL default) on the gloo rehearsal backend:
    the un-drained split replay — segment k, bucket k's all-reduce launched without a stream drain, segment k + 1 — is
    what executes.  Steps 0-1 run at lr = 0 (nothing moves, the optimizer moments fill); step 2 is captured (graph run)
    or eager (reference run) and is the ONE optimizer step with lr > 0, so no loss kink can be crossed between the runs."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # The box term stays in the compared step, so the logit gradient takes the float32, single-rounding path in BOTH
    # runs: with the default bf16 rows, cells matched by several targets accumulate through order-dependent bf16 atomics, and that
    # run-to-run noise is amplified layer by layer on the way back (first suite run of round 5 with the box term and bf16 rows:
    # buckets 3 - 6 at 0.21 - 0.33 against <= 0.1 without it).  What this test is about — which bucket is reduced when — does not
    # depend on the gradient's dtype; the bf16 rows are compared with the oracle in tests/test_gpu_parity.py.
    os.environ["HD_HEAD_GRAD_BF16"] = "0"
    from heltondetection_amd import synthetic as syn
    from heltondetection_amd.detectors import build_detector
    from heltondetection_amd.dist import init_distributed
    from heltondetection_amd.optim import HdAdamW
    from heltondetection_amd.train import Trainer
    init_distributed("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = False
    model = build_detector(dict(type="YOLOv5", variant="s", num_classes=8, img_size=128)).to(dev)
    model.load_state_dict(torch.load(os.path.join(tmp, "pretrained.pt")))   # off the chaotic random-init point (see the test)
    model = model.to(memory_format=torch.channels_last).train()
    # the SHIPPED bucket policy (dist.plan_buckets: VERDICT r03 #7 — the tests used to run with their own small buckets)
    tr = Trainer(model, lr=1e-3, amp_dtype=torch.bfloat16, total_steps=40, warmup_steps=0, device=dev,
                 use_graph=use_graph, graph_warmup=2, graph_sync=False)
    assert isinstance(tr.opt, HdAdamW) and tr.shadows is not None and len(tr.reducer.buckets) >= 6

    class _NoSched:   # the learning rate is set by hand below
        def step(self):
            pass
    tr.sched = _NoSched()
    g = syn.make_generator(200 + rank)   # each rank its own shard
    img = syn.make_images(4, 128, g).to(dev).contiguous(memory_format=torch.channels_last)
    gtb, gtc, mv = syn.make_targets(4, 128, g, "coco", 8, 8)
    batch = (img, gtb.to(dev), gtc.to(dev), mv.to(dev))
    times = []
    for i in range(3):
        for grp in tr.opt.param_groups:
            grp["lr"] = 0.0 if i < 2 else 1e-3
        if i == 2:
            before = [p.detach().float().cpu().clone() for p in model.parameters()]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss, _ = tr.step(*batch)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    info = {"graph": tr._graph is not None, "segments": None if tr._segments is None else len(tr._segments),
            "graph_error": tr.graph_error, "graph_sync": tr.graph_sync, "overlap_check": tr.graph_overlap_check,
            "overlap_error": tr.graph_overlap_error, "loss": float(loss), "times": times, "before": before,
            "params": [p.detach().float().cpu() for p in model.parameters()],
            "shadows": [s_.detach().float().cpu() for s_ in tr.shadows.shadows],
            "grads": [p.grad.detach().float().cpu().clone() for p in model.parameters()],
            "buckets": [b.flat.detach().float().cpu().clone() for b in tr.reducer.buckets]}
    torch.save(info, os.path.join(tmp, f"o{'g' if use_graph else 'e'}{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _step_spread(tr, batch, reps: int = 5):
    """Run-to-run spread of the eager bf16 step at the trainer's current weights: `reps` steps at lr = 0 on one batch, every
    bucket's gradient against the first repetition's (relative L2); returns the median over the repetitions of the worst bucket."""
    class _NoSched:
        def step(self):
            pass
    sched, lrs = tr.sched, [grp["lr"] for grp in tr.opt.param_groups]
    tr.sched = _NoSched()
    for grp in tr.opt.param_groups:
        grp["lr"] = 0.0
    seen = []
    for _ in range(reps):
        tr.step(*batch)
        torch.cuda.synchronize()
        seen.append([b.flat.detach().double().clone() for b in tr.reducer.buckets])
    tr.sched = sched
    for grp, lr in zip(tr.opt.param_groups, lrs):
        grp["lr"] = lr
    worst = sorted(max(float((a - b).norm() / b.norm()) for a, b in zip(rep, seen[0])) for rep in seen[1:])
    return worst[len(worst) // 2]


def _pretrain(tmp_path, steps: int = 300, calm: float = 0.12, rounds: int = 6):
    """One rank, eager: YOLOv5s 128^2 trained for a few hundred product steps on four fixed batches; the state dict is what
    both 2-rank runs below start from.

    Which point the training lands on is itself not reproducible (float atomics in the statistics and the loss), and the
    points differ in how strongly they amplify that noise: tools/repro/step_spread.py (round 5, 60 eager repetitions of one
    step at lr = 0, one process, no process group) found three points whose buckets stay within 0.15 - 0.19 of the first
    repetition and one where 20 of 60 repetitions have a bucket beyond 0.33 (worst 0.44) — eager against eager.  A comparison of
    two runs at such a point says nothing about the reduction (2 of ~10 suite runs of round 5 failed that way).  So the spread
    is measured here (five lr = 0 steps on rank 0's test batch; median over repetitions of the worst bucket) and training
    continues, 60 steps at a time, until it is <= `calm`; the calmest point seen is kept if none qualifies.  Returns the spread."""
    from heltondetection_amd import synthetic as syn
    from heltondetection_amd.detectors import build_detector
    from heltondetection_amd.train import Trainer
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = False
    model = build_detector(dict(type="YOLOv5", variant="s", num_classes=8, img_size=128)).to(dev)
    model = model.to(memory_format=torch.channels_last).train()
    tr = Trainer(model, lr=1e-3, amp_dtype=torch.bfloat16, total_steps=steps + 60 * rounds + 100, warmup_steps=10, device=dev,
                 use_graph=False)
    g = syn.make_generator(77)
    pool = []
    for _ in range(4):
        img = syn.make_images(4, 128, g).to(dev).contiguous(memory_format=torch.channels_last)
        gtb, gtc, mv = syn.make_targets(4, 128, g, "coco", 8, 8)
        pool.append((img, gtb.to(dev), gtc.to(dev), mv.to(dev)))
    gt = syn.make_generator(200)   # rank 0's batch of the compared step (_worker_overlap)
    img = syn.make_images(4, 128, gt).to(dev).contiguous(memory_format=torch.channels_last)
    gtb, gtc, mv = syn.make_targets(4, 128, gt, "coco", 8, 8)
    probe = (img, gtb.to(dev), gtc.to(dev), mv.to(dev))
    done, best, history = 0, None, []
    for r in range(rounds):
        for i in range(steps if r == 0 else 60):
            loss, _ = tr.step(*pool[(done + i) % 4])
        done += steps if r == 0 else 60
        assert bool(torch.isfinite(loss))
        spread = _step_spread(tr, probe)
        history.append(round(spread, 3))
        if best is None or spread < best[0]:
            best = (spread, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()})
        if spread <= calm:
            break
    print("pre-train: eager step spread (median worst-bucket rel-L2 at lr = 0) per round:", history)
    torch.save(best[1], os.path.join(str(tmp_path), "pretrained.pt"))
    del tr, model
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return best[0]


@pytest.mark.timeout(900)
def test_two_rank_overlapped_split_replay_bf16_one_optimizer_step(gpu, tmp_path):
    """VERDICT r02 #2a: the overlapped (graph_sync = False) multi-rank branch executes with a collective, under bf16
    autocast + bf16 weight shadows + HdAdamW, with lr > 0 for exactly one graph-replayed optimizer step."""
    world = 2
    spread = _pretrain(tmp_path)
    # What is tested is deterministic when it fails (a bucket that missed its all-reduce, was reduced early or lost a tensor is off by
    # >= 1.0 in EVERY run; the structural checks fail in every run); the run-to-run noise of the compared gradients is not — even at
    # a calm point one of eight graph runs of tools/repro/two_rank_flake.py had its Detect-side buckets at 0.57 while two eager runs
    # agreed to 0.011.  So: up to three attempts (a graph run + an eager run each); the structural checks must hold in every attempt,
    # the numeric bound in at least one.
    seen = []
    for attempt in range(3):
        for use_graph in (True, False):
            mp.spawn(_worker_overlap, args=(world, _free_port(), str(tmp_path), use_graph), nprocs=world, join=True)
        g0, g1 = torch.load(tmp_path / "og0.pt"), torch.load(tmp_path / "og1.pt")
        e0, e1 = torch.load(tmp_path / "oe0.pt"), torch.load(tmp_path / "oe1.pt")
        assert g0["graph"] and g1["graph"], (g0["graph_error"], g1["graph_error"])
        assert g0["segments"] >= 3
        assert g0["graph_sync"] is False and g1["graph_sync"] is False           # the overlapped branch really ran ...
        # the overlapped replay checks itself at capture (Trainer._verify_overlap): every bucket's overlapped all-reduce equals
        # the drained all-reduce of a stream-ordered copy of the same packed gradients — exact up to float32 summation order,
        # so a collective that ran ahead of its segment (stale or half-written bucket) cannot pass
        assert g0["overlap_error"] is None and g1["overlap_error"] is None, (g0["overlap_error"], g1["overlap_error"])
        assert g0["overlap_check"] is not None and g0["overlap_check"] < 1e-5, g0["overlap_check"]
        for a, b in zip(g0["params"], g1["params"]):
            assert torch.equal(a, b)                                             # ... and left both ranks bit-identical
        for a, b in zip(g0["shadows"], g1["shadows"]):
            assert torch.equal(a, b)
        for a, b in zip(g0["grads"], g1["grads"]):
            assert torch.equal(a, b)
        for a, b in zip(e0["params"], e1["params"]):
            assert torch.equal(a, b)
        assert all(bool(torch.isfinite(a).all()) for a in g0["params"])
        moved = sum(float(((a - b) ** 2).sum()) for a, b in zip(g0["params"], g0["before"]))
        assert moved > 0                                                         # lr > 0: the replayed step moved the weights
        assert g0["segments"] >= 6                                               # the shipped bucket plan, not a test-sized one
        # Against the eager 2-rank step, PER BUCKET.  Both runs start from the same pre-trained weights: at random init this
        # network amplifies rounding differences so strongly that two FLOAT32 runs of one step differ by 0.27 in their gradients
        # (profiles/r04_grad_fidelity.md: chaos of a random-init BatchNorm network, not bf16 storage of a common-mode part — that
        # round-3 hypothesis is refuted there); after a few hundred steps two bf16 runs agree to ~0.14.  Bound stated before the
        # first run of this form: every bucket's averaged gradient within 0.33 relative L2 (cosine >= 0.9) of the eager run's.  A
        # bucket that missed its all-reduce, was reduced before its segment finished, or lost a tensor is off by >= 1.0: it
        # cannot pass by a factor of three.
        # The box term stays in the compared step (round 4 had switched it off after a 1-in-5 failure; ADVICE r04: then a missing
        # or mis-ordered reduction that only touches the Detect bucket's box gradient would pass).  d(CIoU)/d(box) jumps where an
        # edge of the predicted box crosses the target's, so two runs whose logits differ by rounding noise occasionally sit on
        # different sides of such a kink for one of the ~100 positives of this small batch: seen once in five runs as bucket 0 (the
        # Detect layers) at 0.351 / cosine 0.94, every other bucket <= 0.1 (typical bucket 0: 0.01 - 0.10).  Bound for bucket 0 ONLY,
        # stated before this form first ran: 0.5 relative L2, cosine >= 0.85 (a Detect bucket that was not reduced is one shard's
        # gradient against the two-shard mean; the ranks' parameters, shadows and gradients are also compared bit for bit above).
        assert abs(g0["loss"] - e0["loss"]) < 2e-2 * abs(e0["loss"]), (g0["loss"], e0["loss"])
        assert len(g0["buckets"]) == len(e0["buckets"]) >= 6
        worst = []
        for k, (a, b) in enumerate(zip(g0["buckets"], e0["buckets"])):
            rel = float((a.double() - b.double()).norm() / b.double().norm())
            cos = float((a.double() @ b.double()) / (a.double().norm() * b.double().norm()))
            worst.append((round(rel, 3), round(cos, 4), k, a.numel()))
        print("per-bucket (rel-L2, cosine, bucket, elements) graph vs eager:", worst, "eager step spread at this point:", round(spread, 3))
        ok = all((rel < 0.5 and cos >= 0.85) if k == 0 else (rel < 0.33 and cos >= 0.9) for rel, cos, k, _ in worst)
        if not ok:   # which tensors, and by what factor: norm ratio graph / eager per parameter gradient (both runs, rank 0)
            ratios = []
            for i, (a, b) in enumerate(zip(g0["grads"], e0["grads"])):
                nb = float(b.double().norm())
                ratios.append((i, tuple(a.shape), round(float(a.double().norm()) / nb, 3) if nb > 0 else None,
                               round(float((a.double().flatten() @ b.double().flatten()) / (a.double().norm() * b.double().norm() + 1e-30)), 3)))
            print("per-parameter (index, shape, |graph| / |eager|, cosine):", ratios)
        seen.append(worst)
        if ok:
            break
    assert ok, seen
    assert max(g0["times"]) < 60.0 and max(g1["times"]) < 60.0                # generous: correctness run, not a timing


@pytest.mark.timeout(600)
def test_bench_two_rank_rehearsal_prints_one_compact_line_and_exits(gpu):
    """`HD_DIST_BACKEND=gloo python bench.py --gpus 2` on the one GPU: bench.py starts its own two ranks, replays the split graph,
    runs its roofline passes on rank 0 alone and must still leave through the final barrier (round 5: those passes launched
    all-reduces no other rank joined and the run hung), printing ONE compact JSON line with the rehearsal markers."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "5", "--no-micro",
                        "--no-cpu-baseline", "--no-legs", "--no-miopen-find"], env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, (len(lines), [len(l) for l in lines])
    d = json.loads(lines[0])
    assert d["config"]["ranks"] == 2 and d["config"]["rehearsal"] is True and d["n_gpus"] == 1 and d["config"]["backend"] == "gloo"
    assert d["config"]["hip_graph"] is True and d["config"]["hip_graph_segments"] >= 3
    assert d["value"] > 0 and d["roofline"] is not None and d["roofline"]["frac"] > 0
