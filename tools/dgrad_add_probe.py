"""1x1 data gradient with the shortcut gradient added in its epilogue (hd_pointwise_conv_dgrad_add) against the plain data
gradient followed by a bf16 add launch, at the FasterRCNN 832^2 bs16 fork shapes (rows, Cout -> Cin)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import ops
dev = torch.device("cuda:0")
shapes = [(692224, 256, 256), (692224, 64, 64), (692224, 64, 256), (173056, 256, 512), (173056, 128, 256), (173056, 128, 512),
          (43264, 256, 1024), (43264, 256, 512), (10816, 256, 2048) , (102400, 64, 128), (25600, 128, 256)]
def timed(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
for R, N, K in shapes:
    H = 104; B = R // (H * H) if R % (H * H) == 0 else None
    if B is None: H, B = 1, R
    W = R // (B * H)
    dy = torch.randn((B, N, H, W), device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn((N, K), device=dev) * N ** -0.5).bfloat16()
    r = torch.randn((B, K, H, W), device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    ok = bool(__import__("heltondetection_amd")._lib.load().hd_pointwise_conv_dgrad_supported(R, K, N))
    if not ok:
        print(f"R={R} {N}->{K}: not supported"); continue
    t_plain = timed(lambda: ops.pointwise_dgrad_rows(dy, w))
    t_acc = timed(lambda: ops.pointwise_dgrad_rows(dy, w, add=r))
    t_add = timed(lambda: torch.add(r, r))
    mb = R * (N + K) * 2 / 1e6
    print(f"R={R:7d} {N:4d}->{K:4d}  plain {t_plain:7.1f} us ({mb / t_plain:6.2f} TB/s... MB/us)  with add {t_acc:7.1f} us   torch add {t_add:7.1f} us   "
          f"fused saves {t_plain + t_add - t_acc:7.1f} us", flush=True)
