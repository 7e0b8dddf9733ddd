"""Can two ranks of the RCCL (`nccl`) backend share ONE device?  (A one-GPU box is all this round has; if RCCL accepts
it, the overlapped split-replay path can be executed on the real backend instead of the gloo rehearsal.)
Prints the outcome; never raises.  Run: python tools/rccl_same_device_probe.py"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        t = torch.full((1024,), float(rank + 1), device="cuda:0")
        dist.all_reduce(t)
        torch.cuda.synchronize()
        print(f"[rank {rank}] all_reduce on a shared device -> {float(t[0])} (expected 3.0)", flush=True)
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        print(f"[rank {rank}] RCCL refused two ranks on one device: {type(e).__name__}: {str(e)[:300]}", flush=True)


if __name__ == "__main__":
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    try:
        mp.spawn(_worker, args=(2, port), nprocs=2, join=True)
    except Exception as e:  # noqa: BLE001
        print(f"probe ended with {type(e).__name__}: {str(e)[:300]}")
    sys.exit(0)
