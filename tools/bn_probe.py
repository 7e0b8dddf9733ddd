import sys, ctypes as C, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from heltondetection_amd import ops, _lib
dev = torch.device('cuda:0'); lib = _lib.load()
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); evs=[]
    for _ in range(iters):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); evs.append((a,b))
    torch.cuda.synchronize(); ts=sorted(x.elapsed_time(y) for x,y in evs); return ts[len(ts)//2]*1e3
for (B,Cc,H) in [(16,32,320),(16,64,160),(16,32,160),(16,128,80),(16,64,80),(16,256,40),(16,128,40),(16,512,20),(16,256,20)]:
    x = torch.randn((B,Cc,H,H), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(x)
    R = B*H*H
    sums = torch.zeros(64*Cc, device=dev); w = torch.ones(Cc, device=dev); b = torch.zeros(Cc, device=dev)
    save = torch.zeros((2,Cc), device=dev); y = torch.empty_like(x); dg = torch.empty((2,Cc), device=dev)
    st = ops._stream(x)
    t1 = timeit(lambda: lib.hd_bn_act_stats(ops._ptr(x),1,R,Cc,ops._ptr(sums),st))
    sums.zero_(); lib.hd_bn_act_stats(ops._ptr(x),1,R,Cc,ops._ptr(sums),st)
    t2 = timeit(lambda: lib.hd_bn_act_apply(ops._ptr(x),1,R,0,Cc,ops._ptr(sums),ops._ptr(w),ops._ptr(b),1e-3,0.03,None,None,ops._ptr(save[0]),ops._ptr(save[1]),1,None,0,ops._ptr(y),st))
    ds = sums[32*Cc:]
    t3 = timeit(lambda: lib.hd_bn_act_bwd_reduce(ops._ptr(x),ops._ptr(dy),0,1,R,Cc,ops._ptr(save[0]),ops._ptr(save[1]),ops._ptr(w),ops._ptr(b),1,None,ops._ptr(ds),st))
    t4 = timeit(lambda: lib.hd_bn_act_bwd_dx(ops._ptr(x),ops._ptr(dy),0,1,R,0,Cc,ops._ptr(save[0]),ops._ptr(save[1]),ops._ptr(w),ops._ptr(b),1,None,ops._ptr(ds),ops._ptr(y),None,ops._ptr(dg[0]),ops._ptr(dg[1]),st))
    mb = x.numel()*2/1e6
    print(f"B{B} C{Cc} {H}x{H} ({mb:6.1f} MB): stats {t1:6.1f}us ({mb/t1*1e6/1e6:5.2f} TB/s)  apply {t2:6.1f}us ({2*mb/t2:5.2f})  bwd_reduce {t3:6.1f}us ({2*mb/t3:5.2f})  bwd_dx {t4:6.1f}us ({3*mb/t4:5.2f})")
