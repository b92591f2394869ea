"""CPU, world_size 2, gloo: the bucketed gradient reducer averages gradients across ranks exactly like DDP would,
launches one collective per bucket, and batch sharding reproduces the single-process gradients."""
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


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from otter_amd.dp import GradReducer

        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 8))
        net[0].bias.requires_grad_(False)  # a frozen parameter must simply be ignored
        X = torch.randn(8, 16, generator=torch.Generator().manual_seed(1))
        Y = torch.randn(8, 8, generator=torch.Generator().manual_seed(2))
        # single-process reference on the full batch (mean loss over 8 samples)
        ref = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 8))
        ref.load_state_dict(net.state_dict())
        ((ref(X) - Y) ** 2).mean().backward()
        red = GradReducer(net.parameters(), bucket_bytes=2048)  # tiny buckets -> several collectives
        assert len(red.buckets) >= 2
        for step in range(2):  # twice: buffers are reused across steps
            red.zero_grad()
            xs, ys = X[rank::world], Y[rank::world]  # strided shard, like DistributedProxySampler
            ((net(xs) - ys) ** 2).mean().backward()
            red.wait()
            for (n, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
                if p.requires_grad:
                    assert torch.allclose(p.grad, pr.grad, atol=1e-6), n
                    assert p.grad.data_ptr() == red._view[p].data_ptr()
        # no_sync: local accumulation only
        red.zero_grad()
        with red.no_sync():
            ((net(X[rank::world]) - Y[rank::world]) ** 2).mean().backward()
            red.wait()
        g_local = net[2].weight.grad.clone()
        allg = [torch.zeros_like(g_local) for _ in range(world)]
        dist.all_gather(allg, g_local)
        assert not torch.allclose(allg[0], allg[1])
        # gradient accumulation: one no_sync micro-batch + one synchronising micro-batch == the full-batch mean gradient x 2/2,
        # and the synchronising backward launches every bucket FROM ITS HOOKS (overlap), not from wait()
        red.zero_grad()
        halves = [(X[rank::world][:2], Y[rank::world][:2]), (X[rank::world][2:], Y[rank::world][2:])]
        with red.no_sync():
            (((net(halves[0][0]) - halves[0][1]) ** 2).mean() * 0.5).backward()
        assert all(b.work is None for b in red.buckets)
        (((net(halves[1][0]) - halves[1][1]) ** 2).mean() * 0.5).backward()
        assert all(b.work is not None for b in red.buckets), "buckets must launch from the hooks of the synchronising backward"
        red.wait()
        for (n, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
            if p.requires_grad:
                assert torch.allclose(p.grad, pr.grad, atol=1e-6), n
        # a second reducer over the same parameters detaches the first one (no double counting, no rebinding to old buckets)
        from otter_amd import functional as OF

        red2 = GradReducer(net.parameters(), bucket_bytes=1 << 20)
        assert OF.grad_sink is red2 and red._hooks == []
        red2.zero_grad()
        ((net(X[rank::world]) - Y[rank::world]) ** 2).mean().backward()
        red2.wait()
        for (n, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
            if p.requires_grad:
                assert torch.allclose(p.grad, pr.grad, atol=1e-6), n
                assert p.grad.data_ptr() == red2._view[p].data_ptr()
        red2.close()
        assert OF.grad_sink is None
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_grad_reducer_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
