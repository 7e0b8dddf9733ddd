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


def _shadow_worker(rank, world, port, tmp):
    """Same collective, with the convolution weight gradients arriving on bf16 shadow leaves
    (train.LowpShadows' arrangement, emulated on CPU): the reducer must convert them while packing."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from heltondetection_amd.dist import GradReducer, init_distributed
    from heltondetection_amd.layers import LowpWeights, ShadowConv2d
    init_distributed("gloo")
    torch.manual_seed(0)
    model = torch.nn.Sequential(ShadowConv2d(3, 8, 3, padding=1), torch.nn.Tanh(), ShadowConv2d(8, 4, 1, bias=False))
    convs = [m for m in model if isinstance(m, ShadowConv2d)]
    sources = {}
    for m in convs:
        m.weight_lp = m.weight.detach().to(torch.bfloat16).requires_grad_(True)
        sources[m.weight] = m.weight_lp
    red = GradReducer(model, bucket_bytes=1 << 20, first_bucket_bytes=64, grad_sources=sources)
    x = torch.rand((2, 3, 16, 16), generator=torch.Generator().manual_seed(100 + rank))
    for step in range(2):  # twice: prepare() must drop the shadows' gradients too
        red.prepare()
        LowpWeights.on = True
        try:
            y = model[2](model[1](torch.nn.functional.conv2d(x.to(torch.bfloat16), convs[0].lp_weight(),
                                                             convs[0].bias.to(torch.bfloat16), padding=1)))
            loss = (y.float() ** 2).mean()
            # this rank's own bf16 gradients, taken before the reducer's hooks consume them in backward()
            local = [g.float().clone() for g in torch.autograd.grad(loss, [m.weight_lp for m in convs], retain_graph=True)]
            loss.backward()
        finally:
            LowpWeights.on = False
        red.finish()
    assert all(m.weight_lp.grad is None for m in convs)
    assert all(m.weight.grad.dtype == torch.float32 for m in convs)
    torch.save({"reduced": [m.weight.grad.clone() for m in convs] + [convs[0].bias.grad.clone()], "local": local},
               os.path.join(tmp, f"s{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_grad_reducer_with_bf16_grad_sources_two_ranks_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_shadow_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    s0, s1 = torch.load(tmp_path / "s0.pt"), torch.load(tmp_path / "s1.pt")
    for a, b in zip(s0["reduced"], s1["reduced"]):
        assert torch.equal(a, b)
    for red, l0, l1 in zip(s0["reduced"][:2], s0["local"], s1["local"]):
        assert torch.allclose(red, (l0 + l1) / 2, rtol=1e-6, atol=1e-9)  # fp32 mean of the bf16 shard gradients


class _PartlyUsed(torch.nn.Module):
    """`b` never takes part in the forward pass: its bucket stays incomplete after backward."""

    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = torch.nn.Linear(4, 4)
        self.b = torch.nn.Linear(4, 4)
        self.c = torch.nn.Linear(4, 4)

    def forward(self, x):
        return self.c(torch.tanh(self.a(x)))


def _unused_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from heltondetection_amd.dist import GradReducer, init_distributed
    init_distributed("gloo")
    out = {}
    for defer in (False, True):
        model = _PartlyUsed()
        # tiny buckets: a/b/c land in different buckets, and b's bucket is never completed by autograd
        red = GradReducer(model, bucket_bytes=32, first_bucket_bytes=32, defer_collectives=defer)
        assert len(red.buckets) >= 3
        x = torch.full((2, 4), float(rank + 1))
        launches = []
        real = dist.all_reduce

        def counting(t, *a, **k):
            launches.append(t.data_ptr())
            return real(t, *a, **k)

        dist.all_reduce = counting
        try:
            for step in range(2):
                red.prepare()
                model(x).sum().backward()
                red.pack_incomplete()
                red.finish()
                # every bucket reduced exactly once per step, in bucket order
                assert launches == [b.flat.data_ptr() for b in red.buckets], (defer, step)
                launches.clear()
            # "replay": the hooks do not run and prepare() is not called (the hipGraph replays the packing copies);
            # finish() must again reduce every bucket exactly once and leave .grad on the flat views
            if defer:
                for b in red.buckets:
                    b.flat.fill_(float(rank + 1))
                red.finish()
                assert launches == [b.flat.data_ptr() for b in red.buckets]
                launches.clear()
                assert all(torch.allclose(b.flat, torch.full_like(b.flat, 1.5)) for b in red.buckets)
                assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr()
                           for b in red.buckets for p, v in zip(b.params, b.views))
                # restore real gradients for the comparison below
                red.prepare()
                model(x).sum().backward()
                red.finish()
                launches.clear()
        finally:
            dist.all_reduce = real
        out[defer] = {n: p.grad.clone() for n, p in model.named_parameters()}
    torch.save(out, os.path.join(tmp, f"u{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_grad_reducer_unused_parameter_reduced_once_gloo(tmp_path):
    """a bucket with a gradient-less parameter was all-reduced twice in deferred mode (mean 3.0 instead
    of 2.0 for constant per-rank gradients) — every bucket must be reduced exactly once, eager and deferred alike."""
    world, port = 2, _free_port()
    mp.spawn(_unused_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    u0, u1 = torch.load(tmp_path / "u0.pt"), torch.load(tmp_path / "u1.pt")
    # serial reference
    ref = None
    for rk in range(world):
        m = _PartlyUsed()
        m(torch.full((2, 4), float(rk + 1))).sum().backward()
        gs = {n: (p.grad if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()}
        ref = gs if ref is None else {n: ref[n] + gs[n] for n in gs}
    ref = {n: v / world for n, v in ref.items()}
    for defer in (False, True):
        for n in ref:
            assert torch.equal(u0[defer][n], u1[defer][n]), (defer, n)
            assert torch.allclose(u0[defer][n], ref[n], rtol=1e-6, atol=1e-8), (defer, n)
        assert float(u0[defer]["b.weight"].abs().max()) == 0.0  # the unused layer's mean gradient is zero
