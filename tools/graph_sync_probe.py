"""Does hipStreamSynchronize (or an event wait) right after hipGraphLaunch really cover the whole replay?
A ~3 ms graph whose LAST kernel copies a per-iteration value into `out`; after the replay the host synchronises the
launch stream (mode stream), or an event recorded after the launch (mode event), or the device (mode device), and then
reads `out` through a SIDE stream that has no dependency on the launch stream (gloo's staging stream in miniature).
A stale read means the synchronise returned before the replay had finished.

    python tools/graph_sync_probe.py stream|event|device [iters] [pair]
"""
import os, subprocess, sys, time
import torch


def worker(mode, iters, tag):
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    x = torch.randn((1 << 22,), device=dev)
    val = torch.zeros(1, device=dev)
    out = torch.zeros(64, device=dev)

    def body():
        y = x
        for _ in range(150):
            y = y * 1.0001 + 0.5
        out.copy_(val.expand(64) + y[:64] * 0.0)   # the last kernels of the graph: out <- val

    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    side = torch.cuda.Stream(dev)
    pinned = torch.zeros(64).pin_memory()
    torch.cuda.synchronize()
    stale = 0
    main = torch.cuda.current_stream(dev)
    for i in range(1, iters + 1):
        val.fill_(float(i))
        g.replay()
        if mode == "stream":
            main.synchronize()
        elif mode == "event":
            ev = torch.cuda.Event()
            ev.record(main)
            ev.synchronize()
        else:
            torch.cuda.synchronize(dev)
        with torch.cuda.stream(side):
            pinned.copy_(out, non_blocking=True)
        side.synchronize()
        if float(pinned[0]) != float(i):
            stale += 1
    print(f"[{tag}] mode={mode}: {stale} stale reads of {iters}", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    if len(sys.argv) > 3 and sys.argv[3] == "pair":
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), mode, str(iters), f"child{k}"]) for k in range(2)]
        sys.exit(max(p.wait() for p in ps))
    worker(mode, iters, sys.argv[3] if len(sys.argv) > 3 else "single")
