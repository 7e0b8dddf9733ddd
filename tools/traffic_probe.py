"""Launches each HBM-bound kernel a few times at bench shapes; run under `rocprofv3 --pmc FETCH_SIZE` and again
under `--pmc WRITE_SIZE` (tools/pmc_traffic.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import ops, synthetic as syn
from heltondetection_amd.augment import Mosaic
import bench
dev = torch.device('cuda:0'); g = syn.make_generator(3)
N_IT = 4
# fused BN+SiLU on the 105 MB layer
x = (torch.randn((16, 32, 320, 320), generator=g)).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
w = torch.ones(32, device=dev, requires_grad=True); b = torch.zeros(32, device=dev, requires_grad=True)
go = torch.randn((16, 32, 320, 320), generator=g).to(torch.bfloat16).to(dev).contiguous(memory_format=torch.channels_last)
for _ in range(N_IT):
    ops.bn_arena_reset(dev)
    y = ops.bn_act(x, w, b, None, None, 1e-3, 0.03, "silu", True)
    y.backward(go)
    x.grad = None
# decode
views, strides, anchors = bench._yolo_views(16, 640, 85, dev)
for _ in range(N_IT): dec = ops.yolo_decode(views, strides, anchors)
# mosaic
img = syn.make_images(16, 640, g).to(dev).contiguous(memory_format=torch.channels_last)
gtb, gtc, mv = [t.to(dev) for t in syn.make_targets(16, 640, g, "coco", 80, 40)]
src, ctr, apl = Mosaic(p=1.0, generator=torch.Generator(device=dev).manual_seed(0)).draw(16, 640, dev)
for _ in range(N_IT): ops.mosaic4(img, gtb, gtc, mv, src, ctr, apl)
# small-M pairwise IoU
a = syn.random_boxes(306900, 640.0, g).to(dev); bb = torch.stack([syn.random_boxes(40, 640.0, g) for _ in range(16)]).to(dev)
for _ in range(N_IT): o = ops.pairwise_iou(a, bb)
torch.cuda.synchronize(); print("done")
