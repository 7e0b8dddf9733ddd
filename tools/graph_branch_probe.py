"""Do parallel branches of a captured hipGraph run concurrently on this stack?  N small launch-floor-bound kernels captured
(a) as one chain, (b) as two chains on two streams forked from / joined to the capture stream.  Prints replay times."""
import torch

dev = torch.device("cuda", 0)
xs = [torch.randn(16, 128, 40, 40, device=dev, dtype=torch.bfloat16) for _ in range(4)]   # 6.5 MB tensors (launch-floor-bound layers)
big = [torch.randn(16, 64, 160, 160, device=dev, dtype=torch.bfloat16) for _ in range(4)]  # 52 MB tensors (HBM-bound layers)


def chain(t, n):
    for _ in range(n):
        t = torch.nn.functional.silu(t)
    return t


def bench(ts, n, two_streams):
    side = torch.cuda.Stream(dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        if two_streams:
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                b = chain(ts[1], n)
            a = chain(ts[0], n)
            main.wait_stream(side)
        else:
            a = chain(ts[0], n)
            b = chain(ts[1], n)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


for name, ts in (("6.5 MB tensors", xs), ("52 MB tensors", big)):
    for n in (20, 60):
        s = bench(ts, n, False)
        p = bench(ts, n, True)
        print(f"{name}: 2 x {n} silu kernels  one chain {s * 1e3:.1f} us  two branches {p * 1e3:.1f} us  ratio {p / s:.2f}", flush=True)
