"""3x3 / stride 1 forward (+ statistics) and data gradient per YOLOv5s / 832^2 layer shape: kernel-only HIP-event times
(hd_profile_time_next_launch), median of 30 back-to-back launches, TFLOP/s and fraction of the 2.5 PF dense bf16 peak.
(The round-4 kernel it replaced is gone; its times on the same shapes: profiles/r05_conv3.md.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(16, 64, 80, 80, 64), (16, 128, 40, 40, 128), (16, 256, 20, 20, 256), (16, 64, 104, 104, 64), (16, 128, 52, 52, 128),
          (16, 256, 26, 26, 256)]
rep = _lib.load().hd_bn_act_replicas()
print("kernel: csrc/conv3x3.hip (round 5)")
for B, K, H, W, N in SHAPES:
    x = torch.randn(B, K, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(N, K, 3, 3, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    stats = torch.zeros(rep * 2 * N, dtype=torch.float32, device=dev)
    for name, st in (("fwd+stats", stats), ("dgrad", None)):
        for _ in range(5):
            ops.conv3x3_fwd_rows(x, w, st)
        ops.enable_timing("conv3x3_fwd")
        for _ in range(30):
            ops.conv3x3_fwd_rows(x, w, st)
        torch.cuda.synchronize()
        ts = sorted(ops.collect_timing()["conv3x3_fwd"])
        ops.disable_timing()
        ms = ts[len(ts) // 2]
        fl = 18.0 * B * H * W * K * N
        print(f"  {B}x{K}x{H}x{W}->{N} {name:10s} {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  {fl / ms / 1e9 / 2500:.3f} of peak", flush=True)
