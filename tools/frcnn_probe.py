import sys, time, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from heltondetection_amd import synthetic as syn, ops
from heltondetection_amd.detectors import build_detector
from heltondetection_amd.train import Trainer
dev = torch.device('cuda:0')
if '--find' in sys.argv:
    # MIOpen's solver search for ResNet50 at 832^2 is one long call inside the first step (> 7 min on a fresh box):
    # a heartbeat keeps the run visibly alive
    import threading, time as _t
    def _beat():
        t0 = _t.time()
        while True:
            _t.sleep(30)
            print(f"[heartbeat] {int(_t.time() - t0)} s", flush=True)
    threading.Thread(target=_beat, daemon=True).start()
torch.backends.cudnn.benchmark = '--find' in sys.argv
torch.manual_seed(0)
B, S = 16, 832
model = build_detector(dict(type="FasterRCNN", num_classes=80)).to(dev).to(memory_format=torch.channels_last).train()
tr = Trainer(model, lr=2e-4, total_steps=1000, warmup_steps=3, device=dev, use_graph='--graph' in sys.argv)
g = syn.make_generator(1234)
batches = []
for _ in range(2):
    img = syn.make_images(B, S, g).to(dev).contiguous(memory_format=torch.channels_last)
    gtb, gtc, mv = syn.make_targets(B, S, g, "coco", 80, 40)
    batches.append((img, gtb.to(dev), gtc.to(dev), mv.to(dev)))
t0 = time.perf_counter()
for i in range(6):
    loss, parts = tr.step(*batches[i % 2]); torch.cuda.synchronize()
    print("warm", i, round((time.perf_counter() - t0), 1), "s", parts.tolist(), flush=True)
torch.cuda.synchronize(); t = time.perf_counter()
N = 10
for i in range(N):
    loss, parts = tr.step(*batches[i % 2])
torch.cuda.synchronize(); el = (time.perf_counter() - t) / N
print(f"FasterRCNN-R50-PAFPN-Decoupled {S}x{S} bs{B}: {el*1e3:.1f} ms/step, {B/el:.1f} img/s, loss {float(loss):.3f}")
print("graph", tr._graph is not None, tr.graph_error, getattr(tr, "graph_nodes", None)); print("max mem GB", torch.cuda.max_memory_allocated() / 1e9)
