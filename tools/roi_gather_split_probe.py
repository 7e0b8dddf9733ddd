"""Where the time of hd_roi_align_bwd_gather goes: the same launch with RoIs that match no image (scan + zero tiles only)."""
import os, sys, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from heltondetection_amd import ops, _lib, synthetic as syn
dev = torch.device("cuda", 0)
g = syn.make_generator(7)
lib = _lib.load()
maps = [torch.empty((16, 256, 832 // s, 832 // s), device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last) for s in (4, 8, 16, 32)]
st = ops._roi_levels(maps, [1 / 4, 1 / 8, 1 / 16, 1 / 32], 2, 224.0, 4, allow_bf16=True)
for R in (2048, 8192):
    rb = syn.random_boxes(R, 832.0, g, 16.0, 400.0)
    go = torch.randn((R, 256, 7, 7), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    for name, img in (("real", torch.arange(16).repeat_interleave(R // 16)), ("no image matches", torch.full((R,), 99))):
        rois = torch.cat([img[:, None].float(), rb], 1).to(dev)
        ws = torch.empty(lib.hd_roi_align_bwd_gather_workspace_bytes(R, 16), dtype=torch.uint8, device=dev)
        def run():
            ops.check(lib.hd_roi_align_bwd_gather(C.byref(st), 16, ops._ptr(rois), R, 256, 7, 2, 1, ops._ptr(go), 1, go.stride(0), go.stride(1),
                                                  go.stride(2), go.stride(3), ops._ptr(ws), ws.numel(), ops._stream(rois)), "gather")
        for _ in range(2): run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): run()
        b.record(); torch.cuda.synchronize()
        print(f"R={R} {name}: {a.elapsed_time(b) / 5 * 1e3:.0f} us", flush=True)
