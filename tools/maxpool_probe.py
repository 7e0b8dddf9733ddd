import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import ops
dev=torch.device('cuda:0')
x=torch.randn((16,64,416,416),device=dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_(True)
y=ops.maxpool3s2(x); gy=torch.randn_like(y)
def t(fn,it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/it*1e3
print("fwd us", t(lambda: ops.maxpool3s2(x)))
print("bwd us", t(lambda: torch.autograd.grad(y, x, gy, retain_graph=True)))
