"""Per layer shape of YOLOv5s 640^2 bs16: the one-launch 1x1 backward (hd_pointwise_bwd) against the pair
hd_pointwise_conv_dgrad + hd_pointwise_wgrad_partial, kernel-only times (the launches stamp their own events), slice sums
launched on their own so that they can be attributed.  python tools/pw_bwd_probe.py [grid overrides ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import _lib, ops  # noqa: E402

# (rows, Cin, Cout, layers per step)
SHAPES = [(409600, 64, 64, 2), (409600, 32, 32, 1), (102400, 128, 128, 3), (102400, 64, 64, 3), (102400, 256, 128, 1),
          (25600, 256, 256, 5), (25600, 128, 128, 5), (25600, 256, 128, 1), (25600, 512, 256, 1), (6400, 256, 256, 2),
          (6400, 512, 256, 2)]


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def run(fn, names, iters=12, warm=3):
    for _ in range(warm):
        fn()
    ops.enable_timing(*names)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    t = ops.collect_timing()
    ops.disable_timing()
    return {k: med(t[k]) * 1e3 if t.get(k) else 0.0 for k in names}


def main():
    dev = torch.device("cuda", 0)
    lib = _lib.load()
    grids = [int(a) for a in sys.argv[1:]] or [0]
    flush = torch.empty(256 << 20, dtype=torch.uint8, device=dev)   # evict L2 / MALL between calls
    tot_old = tot_new = {g: 0.0 for g in grids}
    tot_old = 0.0
    tot_new = {g: 0.0 for g in grids}
    print("| rows, Cin, Cout | layers | dgrad | wgrad | sum(wgrad slab) | pair total | " + " | ".join(f"fused g={g} (+sum)" for g in grids) + " |")
    for R, K, N, cnt in SHAPES:
        B, H = 16, int((R // 16) ** 0.5)
        assert B * H * H == R
        g = torch.Generator().manual_seed(R + K + N)
        dy = torch.randn((B, N, H, H), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        x = torch.randn((B, K, H, H), generator=g).to(dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = (torch.randn((N, K), generator=g) * K ** -0.5).to(dev).to(torch.bfloat16)

        def old():
            flush.zero_()
            ops.pointwise_dgrad_rows(dy, w)
            ops.pointwise_wgrad_rows(dy, x)
        o = run(old, ("pointwise_dgrad", "pointwise_wgrad", "wgrad_reduce"))
        pair = o["pointwise_dgrad"] + o["pointwise_wgrad"] + o["wgrad_reduce"]
        cells = []
        for gr in grids:
            lib.hd_pointwise_bwd_tune(gr)

            def new():
                flush.zero_()
                ops.pointwise_bwd_rows(dy, w, x)
            n = run(new, ("pointwise_bwd", "wgrad_reduce"))
            cells.append(f"{n['pointwise_bwd']:.1f} (+{n['wgrad_reduce']:.1f}) = {n['pointwise_bwd'] + n['wgrad_reduce']:.1f}")
            tot_new[gr] += cnt * (n["pointwise_bwd"] + n["wgrad_reduce"])
        lib.hd_pointwise_bwd_tune(0)
        tot_old += cnt * pair
        print(f"| {R}, {K}, {N} | {cnt} | {o['pointwise_dgrad']:.1f} | {o['pointwise_wgrad']:.1f} | {o['wgrad_reduce']:.1f} | {pair:.1f} | "
              + " | ".join(cells) + " |", flush=True)
    print(f"per step (us): pair {tot_old:.0f}; fused " + ", ".join(f"g={g}: {v:.0f}" for g, v in tot_new.items()))


if __name__ == "__main__":
    main()
