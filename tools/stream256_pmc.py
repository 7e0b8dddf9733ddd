"""One shape of the K = 256 streaming 1x1 kernel for a PMC pass: rows 692224 (832^2 / 4, bs 16), 256 -> 256, plain data gradient,
forward with statistics, and data gradient with the addend."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import ops, _lib
dev = torch.device("cuda:0")
B, H, W, N, K = 16, 208, 208, 256, 256
dy = torch.randn((B, N, H, W), device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn((N, K), device=dev) * N ** -0.5).bfloat16()
r = torch.randn((B, K, H, W), device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
for _ in range(3):
    ops.pointwise_dgrad_rows(dy, w)
    ops.pointwise_dgrad_rows(dy, w, add=r)
    ops.pointwise_conv_rows(dy, w, None)
torch.cuda.synchronize()
