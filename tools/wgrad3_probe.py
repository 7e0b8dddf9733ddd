"""3x3 weight gradient (stride 1 and 2) per YOLOv5s layer shape: kernel-only HIP-event time of the partial-product launch (with the
previous launch's slice sum riding in it, as in the step), median of 20.  WGRAD3_SELECT=1 | 2: the row kernels only | the flat kernel wherever it runs (hd_conv3x3_wgrad_select)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
# (B, Cin, Ho, Wo, Cout, stride)
SHAPES = [(16, 64, 80, 80, 64, 1), (16, 128, 40, 40, 128, 1), (16, 256, 20, 20, 256, 1), (16, 64, 80, 80, 128, 2), (16, 128, 40, 40, 256, 2),
          (16, 256, 20, 20, 512, 2), (16, 128, 40, 40, 128, 2), (16, 256, 20, 20, 256, 2)]
if os.environ.get("WGRAD3_SHAPES") == "frcnn":  # BASELINE config 3 (ResNet50 + PAFPN, 832^2 bs16): 26 / 52 / 104-pixel rows
    SHAPES = [(16, 128, 104, 104, 128, 1), (16, 256, 52, 52, 256, 1), (16, 512, 26, 26, 512, 1), (16, 256, 104, 104, 256, 1),
              (16, 256, 26, 26, 256, 1), (16, 256, 52, 52, 256, 2), (16, 256, 26, 26, 256, 2), (16, 128, 104, 104, 128, 2)]
from heltondetection_amd import _lib  # noqa: E402
sel = int(os.environ.get("WGRAD3_SELECT", "0"))
_lib.load().hd_conv3x3_wgrad_select(sel)
if sel != 1:
    SHAPES += [(16, 32, 160, 160, 32, 1)]
print("kernels:", {0: "automatic choice", 1: "row kernels (round 4)", 2: "flat-position kernel (round 5)"}[sel])
for B, K, H, W, N, S in SHAPES:
    if not ops.conv3x3_wgrad_supported(torch.empty((B, K, S * H, S * W), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last), N, S):
        print(f"  {B}x{K}x{H}x{W}->{N} s{S}  not taken (library)")
        continue
    x = torch.randn(B, K, S * H, S * W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, N, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    leaf = torch.zeros((N, K, 3, 3), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    ops._WgradChain.enabled = True
    try:
        for _ in range(4):
            ops.conv3x3_wgrad_rows(dy, x, S, leaf=leaf)
        ops.enable_timing("conv3x3_wgrad")
        for _ in range(20):
            ops.conv3x3_wgrad_rows(dy, x, S, leaf=leaf)
        ops.flush_wgrad()
        torch.cuda.synchronize()
        ts = sorted(ops.collect_timing()["conv3x3_wgrad"])
        ops.disable_timing()
    finally:
        ops._WgradChain.enabled = False
        ops._WgradChain.reset()
    ms = ts[len(ts) // 2]
    fl = 18.0 * B * H * W * K * N
    print(f"  {B}x{K}x{H}x{W}->{N} s{S}  {ms * 1e3:7.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s  {fl / ms / 1e9 / 2500:.3f} of peak", flush=True)
