"""Phase stamps (s_memrealtime, 100 MHz) of the 3x3 kernel's workgroups from the diagnostic build tools/micro/bin/libconv3_stamps.so
(tools/micro/build_conv3_stamps.sh).  Per shape: when each workgroup started, when its first chunk had landed, when its MFMA
phases ended, when its epilogue ended — medians and the spread over the grid.  Stamps: 0 start, 1 first chunk stashed, 2 first
barrier passed, 3 last chunk multiplied, 4 tiles stored, 5 end."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tools", "micro", "bin", "libconv3_stamps.so"))
P, I, L = C.c_void_p, C.c_int, C.c_int64
lib.hd_conv3x3_fwd.argtypes = [P, L, P, I, I, I, I, I, P, L, P, P]
lib.hd_conv3x3_fwd.restype = I
lib.hd_conv3_stamps_read.argtypes = [P, I]
dev = torch.device("cuda:0")
for B, K, H, W, N, with_stats in [(16, 64, 80, 80, 64, 0), (16, 128, 40, 40, 128, 0), (16, 256, 20, 20, 256, 0), (16, 128, 40, 40, 128, 1)]:
    x = torch.randn(B, K, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(N, K, 3, 3, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y = torch.empty((B, N, H, W), dtype=torch.bfloat16, device=dev).contiguous(memory_format=torch.channels_last)
    st = torch.zeros(16 * 2 * N, dtype=torch.float32, device=dev)
    for _ in range(5):
        rc = lib.hd_conv3x3_fwd(x.data_ptr(), K, w.data_ptr(), B, H, W, K, N, y.data_ptr(), N, st.data_ptr() if with_stats else None, None)
        assert rc == 0, rc
    torch.cuda.synchronize()
    n = 4096 * 16
    host = np.zeros(n, dtype=np.uint64)
    lib.hd_conv3_stamps_read(host.ctypes.data_as(P), n)
    ROWS = {80: 3, 40: 3, 20: 5}[W]
    wgs = (N // 64) * B * ((H + ROWS - 1) // ROWS)
    s = host.reshape(4096, 16)[: min(wgs, 4096), :6].astype(np.int64)
    t0 = s[:, 0].min()
    us = (s - t0) / 100.0
    print(f"{B}x{K}x{H}x{W}->{N} stats={with_stats}: {wgs} workgroups; kernel span {us[:, 5].max():.2f} us")
    names = ["start", "chunk0 stashed", "first barrier", "MFMA done", "tiles stored", "end"]
    for k, nm in enumerate(names):
        print(f"   {nm:16s} median {np.median(us[:, k]):6.2f}  p10 {np.percentile(us[:, k], 10):6.2f}  p90 {np.percentile(us[:, k], 90):6.2f}  max {us[:, k].max():6.2f}")
    d = np.diff(us, axis=1)
    print("   phase durations (median us):", "  ".join(f"{names[k]}->{names[k + 1]}: {np.median(d[:, k]):.2f}" for k in range(5)))
    if K >= 96:   # chunk 2 in detail: 6 step entered, 7 first barrier passed, 8 stashed, 9 second barrier passed, 10 multiplied
        t = (host.reshape(4096, 16)[: min(wgs, 4096), 6:11].astype(np.int64) - t0) / 100.0
        dd = np.diff(t, axis=1)
        print("   chunk 2 (median us): wait barrier %.2f  stash (incl. wait for the loads) %.2f  issue + barrier %.2f  fragments + MFMA %.2f"
              % tuple(np.median(dd[:, k]) for k in range(4)))
