"""Separates the two suspects behind the alternate-replay stall of the 2-rank rehearsal (second
hazard): (a) a cross-stream wait on an event recorded right after hipGraphLaunch, (b) two processes time-slicing one
GPU.  No torch.distributed here: each process replays a ~5 ms graph, records an event, lets a side stream wait on it and
run one kernel, and then either the HOST waits for the side stream (what gloo's worker thread does before its D2H
staging copy) or the main stream waits on it (what RCCL's stream hand-back does).

    python tools/graph_event_probe.py single host|device|d2h [iters]
    python tools/graph_event_probe.py pair   host|device|d2h [iters]     (two concurrent processes on GPU 0)

d2h = gloo's CUDA all-reduce in miniature: the side stream waits on the event, copies 8 MB device -> PINNED host memory,
a WORKER THREAD synchronises the side stream, "reduces" on the CPU, copies back host -> device on the side stream, and
the main thread joins the worker before it lets the main stream continue (Work.wait()).
"""
import os, subprocess, sys, time
import torch


def worker(mode, iters, tag):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    x = torch.randn((1 << 21,), device=dev)
    w = torch.randn((256, 256), device=dev)
    a = torch.randn((2048, 256), device=dev)

    def body():
        y = x
        for _ in range(120):
            y = y * 1.0001 + 0.5
        b = a
        for _ in range(40):
            b = torch.tanh(b @ w)
        return y, b

    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body()
    side = torch.cuda.Stream(dev)
    z = torch.zeros((1 << 20,), device=dev)
    big = torch.randn((1 << 21,), device=dev)          # 8 MB "gradient bucket"
    pinned = torch.empty((1 << 21,), pin_memory=True)
    import threading
    torch.cuda.synchronize()
    t_replay = time.perf_counter()
    g.replay(); torch.cuda.synchronize()
    t_replay = (time.perf_counter() - t_replay) * 1e3
    times = []
    for i in range(iters):
        t0 = time.perf_counter()
        g.replay()
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            z.add_(1.0)
        if mode == "host":
            side.synchronize()               # host thread blocks on the side stream
        elif mode == "d2h":
            with torch.cuda.stream(side):
                pinned.copy_(big, non_blocking=True)

            def work():
                torch.cuda.set_device(dev)
                side.synchronize()
                pinned.mul_(0.5)
                with torch.cuda.stream(side):
                    big.copy_(pinned, non_blocking=True)
                side.synchronize()
            th = threading.Thread(target=work)
            th.start()
            th.join()
        else:
            torch.cuda.current_stream(dev).wait_stream(side)
        x.mul_(1.0)                           # the "optimizer" behind it on the main stream
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    times_s = sorted(times)
    print(f"[{tag}] mode={mode} one replay {t_replay:.2f} ms; per iteration: median {times_s[len(times_s)//2]:.2f} ms, "
          f"max {times_s[-1]:.2f} ms, >5x median: {sum(t > 5 * times_s[len(times_s)//2] for t in times)} of {iters}", flush=True)


if __name__ == "__main__":
    kind, mode = sys.argv[1], sys.argv[2]
    iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    if kind == "single":
        worker(mode, iters, "single")
    elif kind == "child":
        worker(mode, iters, sys.argv[4])
    else:
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", mode, str(iters), f"pair{k}"]) for k in range(2)]
        rc = [p.wait() for p in ps]
        sys.exit(max(rc))
