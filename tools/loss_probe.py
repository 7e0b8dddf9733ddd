# full loss fwd + bwd at bench shapes, 30 iterations: meant to run under rocprofv3 --kernel-trace
import sys, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from heltondetection_amd import ops, synthetic as syn
import bench
dev = torch.device('cuda:0')
views, strides, anchors = bench._yolo_views(16, 640, 85, dev)
# realistic logits: objectness/class biases as the head initialises them
for v in views:
    v[..., 4] -= 5.0; v[..., 5:] -= 4.0
views = [v.requires_grad_(True) for v in views]
g = syn.make_generator(0)
gtb, gtc, mv = syn.make_targets(16, 640, g, 'coco', 80, 40)
shapes = [(3, v.shape[2], v.shape[3], 85) for v in views]
slots = ops.yolo_assign(shapes, strides, anchors, gtb.to(dev), gtc.to(dev), mv.to(dev))
biases = [torch.zeros(255, device=dev, requires_grad=True) for _ in views]
for i in range(30):
    total, parts = ops.yolo_loss(views, strides, anchors, slots, 40, biases=biases)
    total.backward()
torch.cuda.synchronize()
print("loss", parts.tolist(), "valid slots", int((slots[0] >= 0).sum()), "of", slots[0].numel())
