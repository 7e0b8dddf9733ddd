"""Two long shapes of the 1x1 weight-gradient kernel for a PMC pass (config 3's 208 x 208 level): rows 692224, 256 x 256 and 64 x 256."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
for R, K, N in [(692224, 256, 256), (692224, 64, 256)]:
    nb = lib.hd_pointwise_wgrad_workspace_bytes(R, K, N)
    x = torch.randn(R, K, device=dev).bfloat16(); dy = torch.randn(R, N, device=dev).bfloat16()
    ws = torch.empty(nb, dtype=torch.uint8, device=dev); dw = torch.empty(N, K, dtype=torch.bfloat16, device=dev)
    job = _lib.WgradJob()
    for _ in range(3):
        _lib.check(lib.hd_pointwise_wgrad_partial(P(dy), N, None, None, P(x), K, R, K, N, P(dw), P(ws), nb, None, C.byref(job), st), "partial")
    torch.cuda.synchronize()
