import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
from heltondetection_amd.dist import init_distributed
from heltondetection_amd.detectors import build_detector
from heltondetection_amd.train import Trainer
from heltondetection_amd import synthetic as syn
rank, local, world = init_distributed()
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
torch.manual_seed(0); torch.backends.cudnn.benchmark = False
model = build_detector(dict(type="YOLOv5", variant="s", num_classes=80, img_size=320)).to(dev).to(memory_format=torch.channels_last).train()
tr = Trainer(model, lr=1e-3, total_steps=100, warmup_steps=3, device=dev, use_graph=True)
g = syn.make_generator(1234 + rank)
img = syn.make_images(8, 320, g).to(dev).contiguous(memory_format=torch.channels_last)
gtb, gtc, mv = syn.make_targets(8, 320, g, "coco", 80, 40)
batch = (img, gtb.to(dev), gtc.to(dev), mv.to(dev))
sync_before = "--sync-before-collective" in sys.argv
tick = torch.zeros(64, device=dev)
if "--trailing-kernel" in sys.argv:
    orig_replay = None
    def patch():
        g0 = tr._graph
        orig = g0.replay
        def replay():
            orig(); tick.add_(1.0)
        tr._graph = type("G", (), {"replay": staticmethod(replay)})()
    tr._patch = patch
if "--stream-sync" in sys.argv:
    tr.graph_sync = True
red = tr.reducer
def timed_finish():
    t0 = time.perf_counter()
    if sync_before: torch.cuda.synchronize()
    t1 = time.perf_counter()
    op = dist.ReduceOp.SUM
    works = [dist.all_reduce(b.flat, op=op, async_op=True) for b in red.buckets]
    t2 = time.perf_counter()
    for w, b in zip(works, red.buckets):
        w.wait(); b.flat.div_(red.world)
    t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    tr._t = (t1 - t0, t2 - t1, t3 - t2, t4 - t3)
for i in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if tr._graph is not None:
        red.finish = timed_finish
        if hasattr(tr, "_patch"):
            tr._patch(); del tr._patch
    loss, parts = tr.step(*batch)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    extra = "" if not hasattr(tr, "_t") else " sync %.1f launch %.1f wait %.1f drain %.1f ms" % tuple(x * 1e3 for x in tr._t)
    if rank == 0: print(f"[r{rank}] step {i}: {(t1 - t0) * 1e3:9.1f} ms graph={tr._graph is not None}{extra}", flush=True)
dist.barrier(); dist.destroy_process_group()
