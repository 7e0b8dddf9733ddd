"""pairwise IoU cols kernel, store-path variants (HD_IOU_COLS_VARIANT = 0 row-by-row / 1 staged / 2 staged + nt): kernel-only
time at the bench shape and a few others, plus a checksum so the variants can be compared across processes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heltondetection_amd import ops, synthetic as syn  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    g = syn.make_generator(7)
    shapes = ((2, 409200, 500), (16, 25200, 512)) if os.environ.get("HD_IOU_SWEEP") else \
        ((2, 409200, 500), (16, 25200, 512), (4, 100000, 300), (2, 100003, 700), (3, 50001, 200))
    for B, N, M in shapes:
        a = syn.random_boxes(N, 1280.0, g, 8.0, 512.0).to(dev)
        b = torch.stack([syn.random_boxes(M, 1280.0, g, 4.0, 64.0) for _ in range(B)]).to(dev)
        mv = torch.tensor([M] + [max(M - 37 * (i + 1), 1) for i in range(B - 1)], dtype=torch.int32, device=dev) if N % 2 else None
        out = ops.pairwise_iou(a, b, mv)
        chk = float(out.double().sum())
        for _ in range(3):
            ops.pairwise_iou(a, b, mv)
        ops.enable_timing("pairwise_iou")
        for _ in range(10):
            ops.pairwise_iou(a, b, mv)
        torch.cuda.synchronize()
        ts = sorted(ops.collect_timing()["pairwise_iou"])
        ops.disable_timing()
        ms = ts[len(ts) // 2]
        nbytes = B * (16 * (N + M) + 4 * N * M)
        print(f"variant {os.environ.get('HD_IOU_COLS_VARIANT', '1')} rpw {os.environ.get('HD_IOU_RPW', '-')} grid {os.environ.get('HD_IOU_GRID', '-')} B={B} N={N} M={M} masked={mv is not None}: {ms:.4f} ms "
              f"{nbytes / ms / 1e9:.3f} TB/s  checksum {chk:.6f}", flush=True)
        del out


if __name__ == "__main__":
    main()
