"""N>1 path on CPU: two gloo ranks drive GradReducer (bucketed all-reduce launched from autograd
hooks) over the YOLOv5 backbone+neck; the reduced gradients must equal the mean of the per-rank
gradients computed serially, and both ranks must end a step with identical parameters.
The detection ops are per-image and never exchange data, so the collective under
test is exactly the one the multi-GPU hot path has."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    from heltondetection_amd.backbones import CSPDarknet
    from heltondetection_amd.necks import YOLOv5PAFPN
    torch.manual_seed(0)
    bb = CSPDarknet("n")
    neck = YOLOv5PAFPN(bb.out_channels, depth=bb.depth)
    return torch.nn.Sequential(bb, neck)


def _loss(model, x):
    return sum((f.float() ** 2).mean() for f in model(x))


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from heltondetection_amd.dist import GradReducer, init_distributed, shard_batch
    r, _, w = init_distributed("gloo")
    assert (r, w) == (rank, world)
    model = _model()
    if rank == 1:  # deliberately different init: the reducer must broadcast rank 0's parameters
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    red = GradReducer(model, bucket_bytes=1 << 20, first_bucket_bytes=1 << 18)
    assert len(red.buckets) >= 2
    g = torch.Generator().manual_seed(42)
    x_all = torch.rand((4, 3, 64, 64), generator=g)
    x = x_all[shard_batch(4, rank, world)]
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    red.prepare()
    _loss(model, x).backward()
    red.finish()
    grads = [p.grad.clone() for p in model.parameters()]
    opt.step()
    torch.save({"grads": grads, "params": [p.detach().clone() for p in model.parameters()]}, os.path.join(tmp, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_grad_reducer_two_ranks_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    for a, b in zip(r0["grads"], r1["grads"]):
        assert torch.equal(a, b)  # both ranks hold the same reduced gradients
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    # serial reference: mean of the two shards' gradients (BatchNorm statistics are per shard)
    g = torch.Generator().manual_seed(42)
    x_all = torch.rand((4, 3, 64, 64), generator=g)
    ref = None
    for rk in range(world):
        m = _model()
        _loss(m, x_all[rk * 2:(rk + 1) * 2]).backward()
        gs = [p.grad for p in m.parameters()]
        ref = gs if ref is None else [a + b for a, b in zip(ref, gs)]
    ref = [t / world for t in ref]
    worst = max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(r0["grads"], ref))
    assert worst < 1e-5, worst


def test_shard_batch_partitions():
    from heltondetection_amd.dist import shard_batch
    seen = []
    for r in range(4):
        s = shard_batch(64, r, 4)
        seen += list(range(64))[s]
    assert seen == list(range(64))
