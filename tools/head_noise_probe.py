"""Run-to-run spread of the Detect layers' gradients at FIXED weights and batch (eager product step, bf16 shadows): how far two
evaluations of the same step differ per tensor, with the HIP Detect convolution (default) and with HD_HEAD_CONV=0 (library).
Background for the per-bucket bound of tests/test_gpu_dist_graph.py: bucket 0 = the Detect layers + the end of the neck.
usage: PYTHONPATH=. python tools/head_noise_probe.py [pretrain_steps]"""
import sys

import torch

from heltondetection_amd import synthetic as syn
from heltondetection_amd.detectors import build_detector
from heltondetection_amd.train import Trainer


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = False
    model = build_detector(dict(type="YOLOv5", variant="s", num_classes=8, img_size=128)).to(dev)
    model = model.to(memory_format=torch.channels_last).train()
    tr = Trainer(model, lr=1e-3, amp_dtype=torch.bfloat16, total_steps=steps + 10, warmup_steps=10, device=dev, use_graph=False)
    g = syn.make_generator(77)
    pool = []
    for _ in range(4):
        img = syn.make_images(4, 128, g).to(dev).contiguous(memory_format=torch.channels_last)
        gtb, gtc, mv = syn.make_targets(4, 128, g, "coco", 8, 8)
        pool.append((img, gtb.to(dev), gtc.to(dev), mv.to(dev)))
    for i in range(steps):
        tr.step(*pool[i % 4])
    names = dict((id(p), n) for n, p in model.named_parameters())
    params = list(model.parameters())
    runs = []
    for r in range(4):
        tr._fwd_bwd(*pool[0])
        tr.reducer.finish()
        runs.append([tr.grad_of(p).detach().float().clone() for p in params])
    ref = runs[0]
    head = [i for i, p in enumerate(params) if names[id(p)].startswith("head.")]
    rest = [i for i in range(len(params)) if i not in head]
    for r in range(1, 4):
        def rel(idx):
            num = sum(float(((runs[r][i] - ref[i]) ** 2).sum()) for i in idx)
            den = sum(float((ref[i] ** 2).sum()) for i in idx)
            return (num / den) ** 0.5
        per = {names[id(params[i])]: round(float((runs[r][i] - ref[i]).norm() / ref[i].norm()), 4) for i in head}
        print(f"run {r} vs 0: head tensors rel-L2 {rel(head):.4f}  all other tensors {rel(rest):.4f}  per head tensor {per}")


if __name__ == "__main__":
    main()
