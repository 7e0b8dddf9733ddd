"""Round-2 HBM-traffic probe: launches the kernels bench.py reports rooflines for, at exactly the shapes it reports
them on; run under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (tools/pmc_traffic_r02.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import ops, synthetic as syn
from heltondetection_amd.detectors import build_detector
from heltondetection_amd.layers import ConvBNAct
import bench
dev = torch.device('cuda:0'); g = syn.make_generator(3)
N_IT = int(os.environ.get("HD_PROBE_ITERS", "3"))
# 1. the fused BN+act family over every BatchNorm layer of YOLOv5s 640^2 bs16 (one forward + backward of each layer)
model = build_detector(dict(type="YOLOv5", variant="s", num_classes=80, img_size=640)).to(dev).to(memory_format=torch.channels_last).train()
recs = []
def _hook(mod, inp, out):
    shape = (out[0].shape[0], sum(o.shape[1] for o in out)) + tuple(out[0].shape[2:]) if isinstance(out, tuple) else tuple(out.shape)
    recs.append((shape, mod.act_name, len(inp) > 1 and inp[1] is not None))
hooks = [m.register_forward_hook(_hook) for m in model.modules() if isinstance(m, ConvBNAct)]
img = syn.make_images(16, 640, g).to(dev).contiguous(memory_format=torch.channels_last)
ConvBNAct.fused = False   # shape collection only: keep this pass out of the counted bn_* launches
with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
    model.extract(img)
ConvBNAct.fused = True
for h in hooks: h.remove()
del model
print("bn layers", len(recs))
tens = {}
for shape, act, res in recs:
    if shape not in tens:
        x = torch.randn(shape, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        tens[shape] = (x, torch.randn_like(x), torch.randn_like(x), torch.ones(shape[1], device=dev, requires_grad=True),
                       torch.zeros(shape[1], device=dev, requires_grad=True))
for _ in range(N_IT):
    ops.bn_arena_reset(dev)
    for shape, act, res in recs:
        x, go, r, w, b = tens[shape]
        xr = x.detach().requires_grad_(True)
        y = ops.bn_act(xr, w, b, None, None, 1e-3, 0.03, act, True, None, None, r if res else None, True)
        y.backward(go)
torch.cuda.synchronize()
del tens
# 2. dense loss gradient (bs16 640^2 C=80)
views, strides, anchors = bench._yolo_views(16, 640, 85, dev)
gtb, gtc, mv = [t.to(dev) for t in syn.make_targets(16, 640, g, "coco", 80, 40)]
shapes = [(3, v.shape[2], v.shape[3], 85) for v in views]
slots = ops.yolo_assign(shapes, strides, anchors, gtb, gtc, mv, 4.0)
ops.time_yolo_loss_bwd_dense(views, strides, anchors, slots, 40, (0.05, 1.0, 0.5), (4.0, 1.0, 0.4), iters=N_IT)
# 3. pairwise IoU, both kernels
a = syn.random_boxes(409200, 1280.0, g, 8.0, 512.0).to(dev)
bb = torch.stack([syn.random_boxes(500, 1280.0, g, 4.0, 64.0) for _ in range(2)]).to(dev)
for _ in range(N_IT): o = ops.pairwise_iou(a, bb)
del o
a2 = syn.random_boxes(306900, 640.0, g).to(dev); b2 = torch.stack([syn.random_boxes(40, 640.0, g) for _ in range(16)]).to(dev)
for _ in range(N_IT): o = ops.pairwise_iou(a2, b2)
del o
# 4. RoIAlign: config 3 pyramid; R = 8192 forward (f32 and bf16 maps), R = 2560 forward + backward (the train step's
#    160 sampled RoIs per image: 32 foreground + 128 background)
for dt in (torch.float32, torch.bfloat16):
    feats = [torch.randn((16, 256, 832 // s, 832 // s), device=dev).to(dt).contiguous(memory_format=torch.channels_last) for s in (4, 8, 16, 32)]
    for per_img in (512, 160):
        R = 16 * per_img
        rb = syn.random_boxes(R, 832.0, g, 16.0, 400.0)
        rois = torch.cat([torch.arange(16).repeat_interleave(per_img)[:, None].float(), rb], 1).to(dev)
        fr = [f.detach().requires_grad_(per_img == 160) for f in feats]
        for _ in range(N_IT):
            out = ops.roi_align(fr, rois, 7, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 2, True, 2, 224.0, 4, True)
            if per_img == 160:
                out.backward(torch.ones_like(out))
    del feats
torch.cuda.synchronize(); print("done")
