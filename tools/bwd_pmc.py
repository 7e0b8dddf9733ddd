"""Launches hd_pointwise_bwd a few times on two layer shapes of YOLOv5s 640^2 bs16 (for tools/bwd_pmc.sh: rocprofv3 --pmc)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
for R, K, N in ((409600, 64, 64), (102400, 128, 128), (25600, 256, 256)):
    B, H = 16, int((R // 16) ** 0.5)
    g = torch.Generator().manual_seed(R)
    dy = torch.randn((B, N, H, H), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = torch.randn((B, K, H, H), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn((N, K), generator=g) * K ** -0.5).to(dev).to(torch.bfloat16)
    for _ in range(4):
        ops.pointwise_bwd_rows(dy, w, x)
        ops.pointwise_dgrad_rows(dy, w)
        ops.pointwise_wgrad_rows(dy, x)
    torch.cuda.synchronize()
print("done")
