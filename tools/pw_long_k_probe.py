"""1x1 forwards with Cin > 256 (the eleven YOLOv5s layers still on the library in round 5): hd_pointwise_conv's tiled form with
the BatchNorm statistics in its epilogue against the library's convolution + the separate bn_stats launch.
python tools/pw_long_k_probe.py   (one GPU; kernel time from HIP events over 200 launches, L2-cold operands rotated)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, n=25, inner=16):
    """us per call from a replayed hipGraph of `inner` calls (no host time in the figure)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(3):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(inner):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (n * inner) * 1e3


def main():
    from heltondetection_amd import _lib, ops
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = True
    lib = _lib.load()
    reps = lib.hd_bn_act_replicas()
    for (B, H, K, N) in ((16, 20, 512, 256), (16, 20, 512, 512), (16, 20, 1024, 512), (16, 40, 512, 128), (16, 40, 512, 256),
                         (16, 26, 512, 512), (16, 26, 1024, 512), (16, 40, 1024, 256), (16, 52, 512, 256)):
        ROT = 8
        xs = [torch.randn(B, K, H, H, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(ROT)]
        w = (torch.randn(N, K, 1, 1, device=dev) * K ** -0.5).to(torch.bfloat16)
        stats = torch.zeros(reps * 2 * N, device=dev)
        out = torch.empty(B, N, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ours = timed(lambda i: ops.pointwise_conv_rows(xs[i % ROT], w.view(N, K), stats, out=out))
        y = torch.nn.functional.conv2d(xs[0], w)
        libt = timed(lambda i: torch.nn.functional.conv2d(xs[i % ROT], w))
        from heltondetection_amd.ops._core import _ptr, _stream, check
        st = timed(lambda i: check(lib.hd_bn_act_stats(_ptr(y), 1, B * H * H, N, _ptr(stats), _stream(y)), "stats"))
        ref = torch.nn.functional.conv2d(xs[0].float(), w.float())
        got = ops.pointwise_conv_rows(xs[0], w.view(N, K), stats.zero_(), out=out).float()
        err = float((got - ref).norm() / ref.norm())
        fl = 2 * B * H * H * K * N
        print(f"R={B*H*H} Cin={K} Cout={N}: ours+stats {ours:.1f} us ({fl/ours/1e6:.0f} TF/s)  library {libt:.1f} us (+ bn_stats {st:.1f})  rel err {err:.1e}", flush=True)


if __name__ == "__main__":
    main()
