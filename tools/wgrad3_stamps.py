"""Phase stamps of the flat 3x3 weight-gradient kernel (diagnostic build tools/micro/bin/libconv3_stamps.so).  Stamps: 0 start,
1 zero fill done (first barrier), 2 first block copied (stores issued), 3 second barrier, 4 first block multiplied, 5 all blocks
done, 6 partial stored."""
import ctypes as C
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(ROOT, "tools", "micro", "bin", "libconv3_stamps.so"))
P, I, L, Z = C.c_void_p, C.c_int, C.c_int64, C.c_size_t


class Job(C.Structure):
    _fields_ = [("slab", P), ("slices", I), ("count", L), ("dw", P)]


lib.hd_conv3x3_wgrad_workspace_bytes.restype = Z
lib.hd_conv3x3_wgrad_workspace_bytes.argtypes = [I] * 6
lib.hd_conv3x3_wgrad_partial.argtypes = [P, L, P, L, I, I, I, I, I, I, P, P, Z, P, P, P]
lib.hd_wgrad3_stamps_read.argtypes = [P, I]
dev = torch.device("cuda:0")
for B, K, H, W, N, S in [(16, 128, 40, 40, 128, 1), (16, 64, 80, 80, 64, 1), (16, 128, 40, 40, 256, 2)]:
    x = torch.randn(B, K, S * H, S * W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, N, H, W, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dw = torch.empty((N, K, 3, 3), dtype=torch.bfloat16, device=dev)
    nb = lib.hd_conv3x3_wgrad_workspace_bytes(B, H, W, K, N, S)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    job = Job()
    for _ in range(4):
        rc = lib.hd_conv3x3_wgrad_partial(dy.data_ptr(), N, x.data_ptr(), K, B, H, W, K, N, S, dw.data_ptr(), ws.data_ptr(), nb, None, C.byref(job), None)
        assert rc == 0, rc
    torch.cuda.synchronize()
    n = 4096 * 16
    host = np.zeros(n, dtype=np.uint64)
    lib.hd_wgrad3_stamps_read(host.ctypes.data_as(P), n)
    wgs = min(4096, (N // 32) * (K // 32) * job.slices)
    s = host.reshape(4096, 16)[:wgs, :7].astype(np.int64)
    t0 = s[:, 0].min()
    us = (s - t0) / 100.0
    print(f"{B}x{K}x{H}x{W}->{N} s{S}: {wgs} workgroups ({job.slices} slices); span {us[:, 6].max():.2f} us")
    names = ["start", "zero fill", "block copied", "barrier", "block multiplied", "all blocks", "partial stored"]
    for k, nm in enumerate(names):
        print(f"   {nm:16s} median {np.median(us[:, k]):6.2f}  p10 {np.percentile(us[:, k], 10):6.2f}  p90 {np.percentile(us[:, k], 90):6.2f}  max {us[:, k].max():6.2f}")
    d = np.diff(us, axis=1)
    print("   durations (median us):", "  ".join(f"{names[k + 1]}: {np.median(d[:, k]):.2f}" for k in range(6)))
