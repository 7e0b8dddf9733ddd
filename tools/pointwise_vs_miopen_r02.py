"""Run under rocprofv3 --kernel-trace: per shape 10 x ours (partial + reduce), then 10 x MIOpen wgrad; a marker kernel
(torch.zeros of a distinctive size) separates groups."""
import sys, torch
sys.path.insert(0, '.')
from heltondetection_amd import ops
torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')
shapes = [(160, 64, 64), (80, 128, 128), (80, 64, 64), (40, 256, 256), (40, 128, 128), (20, 512, 512),
          (20, 256, 256), (20, 512, 256), (20, 1024, 512), (40, 512, 256), (40, 256, 128), (80, 256, 128)]
data = []
for hw, K, N in shapes:
    B = 16
    x = torch.randn((B, K, hw, hw), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn((B, N, hw, hw), device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn((N, K, 1, 1), device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = w.view(N, K).t().contiguous()
    for _ in range(3):  # warm (MIOpen find)
        ops.pointwise_wgrad_rows(dy, x)
        torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False))
        torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (True, False, False))
        torch.nn.functional.conv2d(x, w)
    data.append((x, dy, w, wt))
torch.cuda.synchronize()
mark = torch.empty(12345, device=dev)
def sep():
    torch.cuda.synchronize(); mark.fill_(1.0); torch.cuda.synchronize()
sep()
for (hw, K, N), (x, dy, w, wt) in zip(shapes, data):
    for _ in range(10): ops.pointwise_wgrad_rows(dy, x)
    sep()
    for _ in range(10): torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (False, True, False))
    sep()
    for _ in range(10): torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (0, 0), (1, 1), False, (0, 0), 1, (True, False, False))
    sep()
    for _ in range(10): torch.nn.functional.conv2d(x, w)
    sep()
    if K <= 256:
        for _ in range(10): ops.pointwise_conv_rows(x, w.view(N, K))
    sep()
    if N <= 256:
        for _ in range(10): ops.pointwise_conv_rows(dy, wt)
    sep()
print("done")
