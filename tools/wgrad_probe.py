"""1x1 weight gradient at the YOLOv5s shapes: the partial-product launch alone, the slice sum alone, and the partial launch
carrying the slice sum of a same-shaped previous job (hd_wgrad_job piggyback).  Rotates over NB buffer sets so that operands
do not sit in L2.  usage: python tools/wgrad_probe.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
shapes = [(6400, 256, 256), (6400, 512, 512), (6400, 512, 256), (6400, 1024, 512), (25600, 128, 128), (25600, 256, 256),
          (25600, 512, 256), (102400, 64, 64), (102400, 128, 128), (102400, 256, 128), (409600, 64, 64), (409600, 32, 32)]
if "--frcnn" in sys.argv:  # ResNet50 + PAFPN at 832^2 bs16
    shapes = [(692224, 64, 64), (692224, 64, 256), (692224, 256, 64), (173056, 256, 128), (173056, 128, 512), (173056, 512, 128),
              (43264, 512, 256), (43264, 256, 1024), (43264, 1024, 256), (10816, 1024, 512), (173056, 512, 256), (692224, 256, 256)]
NB, IT = (3, 10) if "--frcnn" in sys.argv else (6, 30)
def timed(fn):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(IT): fn(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / IT * 1e3
for R, K, N in shapes:
    nb = lib.hd_pointwise_wgrad_workspace_bytes(R, K, N)
    xs = [torch.randn(R, K, device=dev).bfloat16() for _ in range(NB)]
    dys = [torch.randn(R, N, device=dev).bfloat16() for _ in range(NB)]
    wss = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(NB)]
    dws = [torch.empty(N, K, dtype=torch.bfloat16, device=dev) for _ in range(NB)]
    jobs = [_lib.WgradJob() for _ in range(NB)]
    def partial(i, pig=None):
        j = i % NB
        _lib.check(lib.hd_pointwise_wgrad_partial(P(dys[j]), K * 0 + N, None, None, P(xs[j]), K, R, K, N, P(dws[j]), P(wss[j]), nb,
                                                  C.byref(pig) if pig is not None else None, C.byref(jobs[j]), st), "partial")
    for i in range(NB): partial(i)
    torch.cuda.synchronize()
    t_part = timed(lambda i: partial(i))
    t_red = timed(lambda i: _lib.check(lib.hd_wgrad_reduce(C.byref(jobs[i % NB]), st), "reduce"))
    # piggyback: launch i carries the sum of job (i - 1); the jobs of distinct buffer sets stay valid
    snap = [_lib.WgradJob() for _ in range(NB)]
    for i in range(NB):
        C.memmove(C.byref(snap[i]), C.byref(jobs[i]), C.sizeof(_lib.WgradJob))
    t_pig = timed(lambda i: partial(i, snap[(i - 1) % NB]))
    slab_mb = jobs[0].slices * jobs[0].count * 4 / 1e6
    print(f"R={R:7d} K={K:5d} N={N:5d} slices={jobs[0].slices:4d} slab={slab_mb:6.1f} MB operands={(R*(K+N)*2)/1e6:6.1f} MB | "
          f"partial {t_part:6.1f} us  reduce {t_red:5.1f} us  partial+piggyback {t_pig:6.1f} us", flush=True)
