"""Two ranks on GPU 0 (gloo rehearsal): the captured split-replay step run DRAINED twice and OVERLAPPED twice on the same
static batch, no optimizer step; prints the relative L2 distance of the reduced gradient buckets between the runs, so the
noise floor (drained vs drained) can be told from an ordering error (overlapped vs drained).  python tools/overlap_probe.py [fp32]"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, fp32):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", HD_GRAPH_VERIFY_OVERLAP="0")
    from heltondetection_amd import synthetic as syn
    from heltondetection_amd.detectors import build_detector
    from heltondetection_amd.dist import init_distributed
    from heltondetection_amd.train import Trainer
    init_distributed("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = False
    model = build_detector(dict(type="YOLOv5", variant="s", num_classes=8, img_size=128)).to(dev).to(memory_format=torch.channels_last).train()
    tr = Trainer(model, lr=0.0, amp_dtype=None if fp32 else torch.bfloat16, total_steps=40, warmup_steps=0, bucket_bytes=2 << 20,
                 device=dev, use_graph=True, graph_warmup=2, graph_sync=False)
    g = syn.make_generator(200 + rank)
    img = syn.make_images(4, 128, g).to(dev).contiguous(memory_format=torch.channels_last)
    gtb, gtc, mv = syn.make_targets(4, 128, g, "coco", 8, 8)
    batch = (img, gtb.to(dev), gtc.to(dev), mv.to(dev))
    for _ in range(3):
        tr.step(*batch)
    assert tr._segments is not None, tr.graph_error

    def run(sync):
        tr.graph_sync = sync
        tr._replay_split()
        tr.reducer.finish()
        torch.cuda.synchronize()
        return [b.flat.clone() for b in tr.reducer.buckets]

    def dist_(a, b):
        num = sum(float(((x - y).double() ** 2).sum()) for x, y in zip(a, b))
        den = sum(float((y.double() ** 2).sum()) for y in b)
        return (num / den) ** 0.5, [round((float(((x - y).double() ** 2).sum()) / (float((y.double() ** 2).sum()) + 1e-30)) ** 0.5, 5) for x, y in zip(a, b)]

    s1, s2, o1, o2 = run(True), run(True), run(False), run(False)
    if rank == 0:
        print("segments", len(tr._segments), "buckets", len(tr.reducer.buckets), "fp32" if fp32 else "bf16")
        print("drained vs drained      ", dist_(s2, s1))
        print("overlapped vs drained   ", dist_(o1, s1))
        print("overlapped vs overlapped", dist_(o2, o1))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, "fp32" in sys.argv), nprocs=2, join=True)
