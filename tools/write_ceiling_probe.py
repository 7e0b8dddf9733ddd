"""Write-only ceiling on this part for outputs larger than the 256 MiB Infinity Cache: torch fill / zero / copy on the
pairwise-IoU bench shape's 1.64 GB output and on a 139 MB buffer (the size of yolo_grad_dense's output), HIP events around
10 back-to-back launches."""
import torch

dev = torch.device("cuda", 0)


def t(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, numel in (("1.64 GB (2 x 409200 x 500 f32)", 2 * 409200 * 500), ("139 MB", 139 * 1000 * 1000 // 4), ("413 MB", 16 * 25200 * 256)):
    x = torch.empty(numel, dtype=torch.float32, device=dev)
    y = torch.empty(numel, dtype=torch.float32, device=dev)
    nb = numel * 4
    ms = t(lambda: x.zero_())
    print(f"{name}: zero_  {ms:.4f} ms {nb / ms / 1e9:.3f} TB/s written")
    ms = t(lambda: x.fill_(1.5))
    print(f"{name}: fill_  {ms:.4f} ms {nb / ms / 1e9:.3f} TB/s written")
    ms = t(lambda: y.copy_(x))
    print(f"{name}: copy_  {ms:.4f} ms {2 * nb / ms / 1e9:.3f} TB/s read+written")
    ms = t(lambda: torch.add(x, 1.0, out=y))
    print(f"{name}: add    {ms:.4f} ms {2 * nb / ms / 1e9:.3f} TB/s read+written")
    del x, y
