import os, sys, torch
sys.path.insert(0, os.getcwd())
from heltondetection_amd import ops
dev = torch.device("cuda:0")
print("HD_TMP_CAP", os.environ.get("HD_TMP_CAP"))
for R, K, N in [(6400, 256, 256), (6400, 512, 256), (25600, 256, 256), (25600, 512, 256), (102400, 256, 128)]:
    B, H, W = 16, int((R // 16) ** 0.5), int((R // 16) ** 0.5)
    dy = torch.randn(B, N, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x = torch.randn(B, K, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
    leaf = torch.zeros(N, K, dtype=torch.bfloat16, device=dev, requires_grad=True)
    ops._WgradChain.enabled = True
    try:
        for _ in range(4):
            ops.pointwise_bwd_rows(dy, w, x, leaf=leaf)
        ops.enable_timing("pointwise_bwd")
        for _ in range(20):
            ops.pointwise_bwd_rows(dy, w, x, leaf=leaf)
        ops.flush_wgrad(); torch.cuda.synchronize()
        ts = sorted(ops.collect_timing()["pointwise_bwd"]); ops.disable_timing()
    finally:
        ops._WgradChain.enabled = False; ops._WgradChain.reset()
    print(f"  {R}x{K}x{N}: {ts[len(ts)//2]*1e3:6.1f} us")
