"""SyncBN (SURVEY.md §8e "optional second exchange"): two ranks, each holding half of a batch, with the fused
BatchNorm's statistics all-reduced over the group, must reproduce what one process computes on the full batch —
outputs, input gradients, running statistics; the weight/bias gradients of the two ranks must add up to the
full-batch ones (the data-parallel gradient all-reduce averages them afterwards like any other parameter)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    g = torch.Generator().manual_seed(151)
    B, C, H, W = 8, 64, 6, 10
    x = (torch.randn((B, C, H, W), generator=g) * 1.3 + 0.2)
    w = torch.rand(C, generator=g) + 0.5
    b = torch.randn(C, generator=g) * 0.2
    go = torch.randn((B, C, H, W), generator=g)
    return x, w, b, go


def _run(x, w, b, go, dev, sync_group):
    from heltondetection_amd import ops
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    rm, rv = torch.zeros(x.shape[1], device=dev), torch.ones(x.shape[1], device=dev)
    ops.bn_arena_reset(dev)
    y = ops.bn_act(xd, wd, bd, rm, rv, 1e-3, 0.03, "silu", True, sync_group=sync_group)
    y.backward(go.to(dev).contiguous(memory_format=torch.channels_last))
    return [t.detach().float().cpu() for t in (y, xd.grad, wd.grad, bd.grad, rm, rv)]


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), HD_DIST_BACKEND="gloo")
    import torch.distributed as dist
    from heltondetection_amd.dist import init_distributed
    init_distributed("gloo")
    dev = torch.device("cuda", 0)          # both ranks share the one GPU of the test box
    torch.cuda.set_device(dev)
    x, w, b, go = _case()
    h = x.shape[0] // world
    sl = slice(rank * h, (rank + 1) * h)
    out = _run(x[sl], w, b, go[sl], dev, dist.group.WORLD)
    torch.save(out, os.path.join(tmp, f"sync{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sync_bn_two_ranks_equal_full_batch(gpu, tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    x, w, b, go = _case()
    full = _run(x, w, b, go, gpu, None)
    r = [torch.load(tmp_path / f"sync{k}.pt") for k in range(world)]
    h = x.shape[0] // world

    def close(a, c, tol=2e-5):
        return float((a - c).abs().max()) <= tol * float(c.abs().max()) + 1e-7

    for k in range(world):
        assert close(r[k][0], full[0][k * h:(k + 1) * h])          # y
        assert close(r[k][1], full[1][k * h:(k + 1) * h])          # dx
        assert close(r[k][4], full[4]) and close(r[k][5], full[5])  # running mean / var from the global statistics
    assert close(r[0][2] + r[1][2], full[2]) and close(r[0][3] + r[1][3], full[3])   # dgamma, dbeta add up
    # and without synchronisation the halves would NOT match the full batch (the test has teeth)
    half = _run(x[:h], w, b, go[:h], gpu, None)
    assert not close(half[0], full[0][:h], 1e-3)
