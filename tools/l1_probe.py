"""The first stride-2 3x3 layer of YOLOv5s at 640^2 bs16 (32 -> 64 channels, 320^2 -> 160^2): hd_conv3x3s2_fwd (+ statistics) and
hd_conv3x3s2_dgrad against the library's kernels (+ bn_stats, + the zero-fill its data gradient needs), in replayed hipGraphs.
python tools/l1_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    from pw_long_k_probe import timed
    from heltondetection_amd import _lib, ops
    from heltondetection_amd.ops._core import _ptr, _stream, check
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = True
    lib = _lib.load()
    reps = lib.hd_bn_act_replicas()
    B, H = 16, 160
    ROT = 4
    xs = [torch.randn(B, 32, 2 * H, 320, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(ROT)]
    gs = [torch.randn(B, 64, H, 160, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(ROT)]
    w = (torch.randn(64, 32, 3, 3, device=dev) * 0.06).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    stats = torch.zeros(reps * 2 * 64, device=dev)
    y = torch.nn.functional.conv2d(xs[0], w, None, 2, 1)
    t_fwd = timed(lambda i: ops.conv3x3s2_fwd_rows(xs[i % ROT], w, stats), inner=8)
    t_fwd0 = timed(lambda i: ops.conv3x3s2_fwd_rows(xs[i % ROT], w, None), inner=8)
    t_lib = timed(lambda i: torch.nn.functional.conv2d(xs[i % ROT], w, None, 2, 1), inner=8)
    t_st = timed(lambda i: check(lib.hd_bn_act_stats(_ptr(y), 1, B * H * 160, 64, _ptr(stats), _stream(y)), "stats"), inner=8)
    t_dg = timed(lambda i: ops.conv3x3s2_dgrad_rows(gs[i % ROT], w), inner=8)
    t_ldg = timed(lambda i: torch.ops.aten.convolution_backward(gs[i % ROT], xs[0], w, None, (2, 2), (1, 1), (1, 1), False, (0, 0), 1,
                                                                (True, False, False)), inner=8)
    t_wg = timed(lambda i: ops.conv3x3s2_wgrad_rows(gs[i % ROT], xs[i % ROT]), inner=8)
    t_lwg = timed(lambda i: torch.ops.aten.convolution_backward(gs[i % ROT], xs[i % ROT], w, None, (2, 2), (1, 1), (1, 1), False, (0, 0), 1,
                                                                (False, True, False)), inner=8)
    mb_f = 2 * (B * H * 160 * (4 * 32 + 64)) / 1e6
    print(f"forward : ours + statistics {t_fwd:.1f} us ({mb_f / t_fwd:.2f} TB/s of {mb_f:.0f} MB), without {t_fwd0:.1f}; library {t_lib:.1f} + bn_stats {t_st:.1f}")
    # the 32 -> 32 3x3 / stride 1 layer at 160^2 (csrc/conv3x3_c32.hip)
    x32 = [torch.randn(B, 32, 160, 160, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last) for _ in range(ROT)]
    w32 = (torch.randn(32, 32, 3, 3, device=dev) * 0.06).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    st32 = torch.zeros(reps * 2 * 32, device=dev)
    t_c = timed(lambda i: ops.conv3x3_fwd_rows(x32[i % ROT], w32, st32), inner=8)
    t_c0 = timed(lambda i: ops.conv3x3_fwd_rows(x32[i % ROT], w32, None), inner=8)
    t_cl = timed(lambda i: torch.nn.functional.conv2d(x32[i % ROT], w32, None, 1, 1), inner=8)
    t_cd = timed(lambda i: torch.ops.aten.convolution_backward(x32[i % ROT], x32[0], w32, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                               (True, False, False)), inner=8)
    print(f"32->32 3x3 at 160^2: ours + statistics {t_c:.1f} us, without (= data gradient) {t_c0:.1f}; library forward {t_cl:.1f}, data gradient incl. fill {t_cd:.1f}")
    print(f"wgrad   : ours (partials + sum) {t_wg:.1f} us; library incl. its fill and cast {t_lwg:.1f}")
    print(f"dgrad   : ours {t_dg:.1f} us ({mb_f / t_dg:.2f} TB/s); library incl. its fill {t_ldg:.1f}")


if __name__ == "__main__":
    main()
