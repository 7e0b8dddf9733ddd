"""Run-to-run spread of ONE bf16 train step at fixed weights (lr = 0) and fixed data: eager forward+backward repeated, the
gradient compared with the first repetition (relative L2) — per switch setting, so a nondeterministic kernel family can be
told from chaotic amplification of float-atomic summation order.  python tools/determinism_probe.py [size] [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    from heltondetection_amd import ops, synthetic as syn
    from heltondetection_amd.detectors import build_detector
    from heltondetection_amd.layers import ConvBNAct, FusedBNAct
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = False
    g = syn.make_generator(200)
    img = syn.make_images(B, size, g).to(dev).contiguous(memory_format=torch.channels_last)
    gtb, gtc, mv = syn.make_targets(B, size, g, "coco", 8, 8)
    batch = (img, gtb.to(dev), gtc.to(dev), mv.to(dev))
    for name, sw in (("default", {}), ("conv3x3_forward off", {"conv3x3_forward": False}), ("pointwise off (MIOpen convs)", {"pointwise": False}),
                     ("fused BN off + pointwise off", {"pointwise": False, "bn": False}), ("fp32 default", {"fp32": True})):
        ConvBNAct.pointwise = sw.get("pointwise", True)
        ConvBNAct.conv3x3_forward = sw.get("conv3x3_forward", True)
        FusedBNAct.enabled = sw.get("bn", True)
        torch.manual_seed(0)
        model = build_detector(dict(type="YOLOv5", variant="s", num_classes=8, img_size=size)).to(dev).to(memory_format=torch.channels_last).train()
        grads, losses = [], []
        for it in range(4):
            model.zero_grad(set_to_none=True)
            ops.bn_arena_reset(dev)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not sw.get("fp32", False)):
                loss, _ = model.forward_train(*batch)
            loss.backward()
            torch.cuda.synchronize()
            grads.append([p.grad.detach().float().clone() for p in model.parameters() if p.grad is not None])
            losses.append(float(loss))
        ds = []
        for k in range(1, 4):
            num = sum(float(((a - b).double() ** 2).sum()) for a, b in zip(grads[k], grads[0]))
            den = sum(float((b.double() ** 2).sum()) for b in grads[0])
            ds.append((num / den) ** 0.5)
        print(f"{size}^2 bs{B} {name:32s} loss spread {max(losses) - min(losses):.3e} (loss {losses[0]:.5f})  grad rel-L2 vs run 0: "
              + " ".join(f"{d:.2e}" for d in ds), flush=True)
    ConvBNAct.pointwise = ConvBNAct.conv3x3_forward = FusedBNAct.enabled = True


if __name__ == "__main__":
    main()
