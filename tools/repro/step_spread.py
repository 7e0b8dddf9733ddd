"""Run-to-run spread of ONE rank's train step at the 2-rank test's pre-trained point (tests/test_gpu_dist_graph.py::_pretrain),
lr = 0, same batch every step: N eager steps, twice (two trainers), in one process, no process group.  Per bucket of the
shipped bucket plan: relative L2 of every repetition's gradient against the first repetition's.  Shows how strongly a given
pre-trained point amplifies the float-atomic summation order — eager against eager, nothing replayed, nothing exchanged
(profiles/r05_two_rank_flake.md).  One-rank graph mode keeps its gradients where autograd left them (GradReducer `direct`),
so only eager trainers are read here.

    python tools/repro/step_spread.py [points] [N]
"""
import os
import sys
import tempfile

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def run(tmp, use_graph, n, ref=None):
    from heltondetection_amd import synthetic as syn
    from heltondetection_amd.detectors import build_detector
    from heltondetection_amd.dist import plan_buckets
    from heltondetection_amd.train import Trainer
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = False
    model = build_detector(dict(type="YOLOv5", variant="s", num_classes=8, img_size=128)).to(dev)
    model.load_state_dict(torch.load(os.path.join(tmp, "pretrained.pt")))
    model = model.to(memory_format=torch.channels_last).train()
    tr = Trainer(model, lr=1e-3, amp_dtype=torch.bfloat16, total_steps=10 ** 6, warmup_steps=0, device=dev, use_graph=use_graph,
                 graph_warmup=2)

    class _NoSched:
        def step(self):
            pass
    tr.sched = _NoSched()
    for grp in tr.opt.param_groups:
        grp["lr"] = 0.0
    g = syn.make_generator(200)
    img = syn.make_images(4, 128, g).to(dev).contiguous(memory_format=torch.channels_last)
    gtb, gtc, mv = syn.make_targets(4, 128, g, "coco", 8, 8)
    batch = (img, gtb.to(dev), gtc.to(dev), mv.to(dev))
    red = tr.reducer
    out, losses = [], []
    for i in range(n + 3):
        loss, _ = tr.step(*batch)
        torch.cuda.synchronize()
        if i < 3:
            continue
        losses.append(float(loss))
        out.append([torch.cat([(red._grad_of(p) if red._grad_of(p) is not None else torch.zeros_like(p)).detach().float().flatten()
                               for p in b.params]).double() for b in red.buckets])
    assert (tr._graph is not None) == use_graph, tr.graph_error
    if ref is None:
        ref = out[0]
    rel = torch.tensor([[float((a - b).norm() / b.norm()) for a, b in zip(o, ref)] for o in out])
    tag = "graph" if use_graph else "eager"
    print(f"  {tag}: loss min/max {min(losses):.5f} {max(losses):.5f}")
    print(f"  {tag}: per-bucket median rel", [round(float(x), 3) for x in rel.median(0).values])
    print(f"  {tag}: per-bucket max rel   ", [round(float(x), 3) for x in rel.max(0).values])
    print(f"  {tag}: repetitions with any bucket > 0.33:", int((rel.max(1).values > 0.33).sum()), "of", len(out), flush=True)
    return ref


def main():
    from tests import test_gpu_dist_graph as t
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    for p in range(P):
        tmp = tempfile.mkdtemp(prefix="spread")
        t._pretrain(tmp)
        print(f"== pre-trained point {p}", flush=True)
        ref = run(tmp, False, N)
        run(tmp, False, N, ref)


if __name__ == "__main__":
    main()
