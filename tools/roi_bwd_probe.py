"""RoIAlign backward at config 3's shapes (832^2 bs16 pyramid, C = 256, P = 7, 128 sampled RoIs per image): the gather kernel
(hd_roi_align_bwd_gather) against the float-atomic scatter + its zero-fill and cast, wall time per backward (HIP events around
the autograd backward, so the scatter's zero-fill of the float32 maps and its cast to bf16 are included) and kernel only."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import ops, synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
g = syn.make_generator(7)
for dt, dn in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
    feats = [torch.randn((16, 256, 832 // s, 832 // s), device=dev).to(dt).contiguous(memory_format=torch.channels_last) for s in (4, 8, 16, 32)]
    for per_img in (128, 512):
        R = 16 * per_img
        rb = syn.random_boxes(R, 832.0, g, 16.0, 400.0)
        rois = torch.cat([torch.arange(16).repeat_interleave(per_img)[:, None].float(), rb], 1).to(dev)
        for mode in ("gather", "scatter"):
            ops.RoiBwdSwitch.gather = mode == "gather"
            fr = [f.detach().requires_grad_(True) for f in feats]
            out = ops.roi_align(fr, rois, 7, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 2, True, 2, 224.0, 4, True)
            go = torch.randn_like(out)
            for _ in range(3):
                out.backward(go, retain_graph=True)
            torch.cuda.synchronize()
            ops.enable_timing("roi_align_bwd")
            evs = []
            for _ in range(8):
                for f in fr:
                    f.grad = None
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                out.backward(go, retain_graph=True)
                b.record()
                evs.append((a, b))
            torch.cuda.synchronize()
            kt = sorted(ops.collect_timing()["roi_align_bwd"])
            ops.disable_timing()
            wall = sorted(x.elapsed_time(y) for x, y in evs)
            print(f"{dn} R={R} {mode:8s}: kernel {kt[len(kt) // 2] * 1e3:8.1f} us   whole backward (zero-fill + kernel + cast) {wall[len(wall) // 2] * 1e3:8.1f} us", flush=True)
        ops.RoiBwdSwitch.gather = True
    del feats
