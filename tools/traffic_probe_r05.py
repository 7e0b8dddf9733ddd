"""Round-4 HBM-traffic probe of the kernels bench.py reports single-kernel rooflines for (pairwise IoU, RoIAlign forward and
the gather backward), at exactly the shapes bench.py's micro leg uses; run under `rocprofv3 --pmc FETCH_SIZE` and again under
`--pmc WRITE_SIZE` (tools/pmc_traffic_r05.sh)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import ops, synthetic as syn
dev = torch.device('cuda:0'); g = syn.make_generator(3)
N_IT = int(os.environ.get("HD_PROBE_ITERS", "3"))
a = syn.random_boxes(409200, 1280.0, g, 8.0, 512.0).to(dev)
bb = torch.stack([syn.random_boxes(500, 1280.0, g, 4.0, 64.0) for _ in range(2)]).to(dev)
for _ in range(N_IT): o = ops.pairwise_iou(a, bb)
del o
a2 = syn.random_boxes(306900, 640.0, g).to(dev); b2 = torch.stack([syn.random_boxes(40, 640.0, g) for _ in range(16)]).to(dev)
for _ in range(N_IT): o = ops.pairwise_iou(a2, b2)
del o
for dt in (torch.float32, torch.bfloat16):
    feats = [torch.randn((16, 256, 832 // s, 832 // s), device=dev).to(dt).contiguous(memory_format=torch.channels_last) for s in (4, 8, 16, 32)]
    for per_img in (512, 128):
        R = 16 * per_img
        rb = syn.random_boxes(R, 832.0, g, 16.0, 400.0)
        rois = torch.cat([torch.arange(16).repeat_interleave(per_img)[:, None].float(), rb], 1).to(dev)
        fr = [f.detach().requires_grad_(per_img == 128) for f in feats]
        for _ in range(N_IT):
            out = ops.roi_align(fr, rois, 7, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 2, True, 2, 224.0, 4, True)
            if per_img == 128:
                out.backward(torch.ones_like(out))
    del feats
torch.cuda.synchronize(); print("done")
