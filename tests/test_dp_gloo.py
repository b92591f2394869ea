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
        # overlap=False (bench.py --dp-overlap off): nothing is launched from the hooks, every bucket is reduced in wait(); same gradients
        red3 = GradReducer(net.parameters(), bucket_bytes=2048, overlap=False)
        red3.zero_grad()
        ((net(X[rank::world]) - Y[rank::world]) ** 2).mean().backward()
        assert all(b.work is None for b in red3.buckets), "overlap=False must not launch from the gradient hooks"
        red3.wait()
        for (n, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
            if p.requires_grad:
                assert torch.allclose(p.grad, pr.grad, atol=1e-6), n
        red3.close()
        # collective="rs_ag" (bench.py --dp-collective rs_ag): every bucket as reduce-scatter + all-gather; same averages (another summation
        # order), views still alias the flat buffers, buckets padded to a multiple of the world size
        red4 = GradReducer(net.parameters(), bucket_bytes=2048, collective="rs_ag")
        assert all(b.flat.numel() % world == 0 for b in red4.buckets)
        for step in range(2):
            red4.zero_grad()
            ((net(X[rank::world]) - Y[rank::world]) ** 2).mean().backward()
            red4.wait()
            for (n, p), (_, pr) in zip(net.named_parameters(), ref.named_parameters()):
                if p.requires_grad:
                    assert torch.allclose(p.grad, pr.grad, atol=1e-6), n
                    assert p.grad.data_ptr() == red4._view[p].data_ptr()
        red4.close()
        # SparseEmbedSink with RAGGED row counts (find_and_remove_tokens pads each rank's batch to its own longest sample): the MAX of the
        # counts is exchanged at forward time (exchange_counts), apply() pads to it without a collective of its own and adds every rank's rows
        from otter_amd.train import SparseEmbedSink

        V, D = 11, 4
        emb = torch.nn.Parameter(torch.zeros(V, D))
        sink = SparseEmbedSink()
        sink.anchor(emb.device)
        n_rows = 3 + 2 * rank                                    # 3 rows on rank 0, 5 on rank 1
        ids = (torch.arange(n_rows) * 2 + rank) % V
        rows = torch.full((n_rows, D), float(rank + 1))
        sink.expect(n_rows)
        sink.exchange_counts(None, world)
        assert int(sink._nmax[0][0]) == 3 + 2 * (world - 1)
        sink.add(emb, ids, rows)
        sink.apply(None, world)
        want = torch.zeros(V, D)
        for r in range(world):
            nr = 3 + 2 * r
            want.index_add_(0, (torch.arange(nr) * 2 + r) % V, torch.full((nr, D), float(r + 1)), alpha=1.0 / world)
        assert torch.allclose(emb.grad, want), (emb.grad, want)
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


def _otter_worker(rank, world, port, q):
    """Tiny OtterForConditionalGeneration through TrainStep on 2 ranks: the weight gradients of the fusion modules reach the flat
    buckets through functional.grad_sink (take / ready), everything else through the post-accumulate hooks; the averaged gradients
    equal the single-process mean over the two rank losses (DDP semantics, instruction_following.py:311-314,491-494), and with
    `mask_lm_head` the tied embedding leaves the buckets and only its <answer> row is reduced.  The arithmetic of the fusion
    modules comes from the numpy oracle (tests/_cpu_backend.py): this checks the PLUMBING of the DP path on the real model graph."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OTTER_STUB_TOKENIZER="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np

        from oracle import synth
        from otter_amd import functional as OF
        from otter_amd.train import TrainStep
        from tests import _golden as G
        from tests._cpu_backend import oracle_backend
        from tests.test_host_contract import tiny_model

        def build():
            model = tiny_model()
            m = G.meta()["otter_tiny"]
            sd = synth.state_dict_for(m["seed"], {k: tuple(v) for k, v in m["state_dict_shapes"].items()})
            model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
            return model.train(), m

        model, m = build()
        vision_x, ids, mask, labels = (torch.from_numpy(a) for a in synth.tiny_batch(m["seed"]))
        assert ids.shape[0] == world
        with oracle_backend():
            # single-process reference: mean over the rank losses
            ref, _ = build()
            tot = 0
            for r in range(world):
                tot = tot + ref(vision_x=vision_x[r:r + 1], lang_x=ids[r:r + 1], attention_mask=mask[r:r + 1], labels=labels[r:r + 1])[0] / world
            tot.backward()
            ref_g = {n: p.grad.clone() for n, p in ref.named_parameters() if p.requires_grad}

            sl = slice(rank, rank + 1)
            step = TrainStep(model, lr=1e-3, autocast_dtype=None, bucket_bytes=64 << 10, hip_optimizer=False, max_grad_norm=None)
            red = step.reducer
            assert red is not None and OF.grad_sink is red and len(red.buckets) >= 3
            before = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
            step.optimizer.step = lambda: None          # look at the reduced gradients before any update
            step(vision_x[sl], ids[sl], mask[sl], labels[sl])
            names = {id(p): n for n, p in model.named_parameters()}
            for p, v in red._view.items():
                n = names[id(p)]
                assert p.grad is not None and p.grad.data_ptr() == v.data_ptr(), n
                assert torch.allclose(p.grad, ref_g[n], rtol=1e-4, atol=1e-6), (n, float((p.grad - ref_g[n]).abs().max()))
            assert all(b.work is not None for b in red.buckets)
            step.close()

            # --mask_lm_head: the tied embedding is row-only
            ans = model.text_tokenizer.encode("<answer>")[-1]
            step2 = TrainStep(model, lr=1e-3, autocast_dtype=None, bucket_bytes=64 << 10, hip_optimizer=False, max_grad_norm=None,
                              mask_lm_head=True, answer_token_id=ans)
            wte = model.lang_encoder.transformer.wte.weight
            assert wte not in step2.reducer._owner and wte in step2.reducer.row_only
            step2.optimizer.step = lambda: None
            step2(vision_x[sl], ids[sl], mask[sl], labels[sl])
            want = torch.zeros_like(wte)
            want[ans] = ref_g["lang_encoder.transformer.wte.weight"][ans]
            assert torch.allclose(wte.grad, want, rtol=1e-4, atol=1e-7)
            assert float(wte.grad[ans].abs().max()) > 0
            step2.close()
            assert OF.grad_sink is None

            # a real optimizer step keeps the replicas identical
            step3 = TrainStep(model, lr=1e-3, autocast_dtype=None, bucket_bytes=64 << 10, hip_optimizer=False)
            step3(vision_x[sl], ids[sl], mask[sl], labels[sl])
            chk = torch.stack([p.detach().double().sum() for p in model.parameters() if p.requires_grad])
            both = [torch.zeros_like(chk) for _ in range(world)]
            dist.all_gather(both, chk)
            assert torch.equal(both[0], both[1])
            assert any(not torch.equal(p.detach(), before[n]) for n, p in model.named_parameters() if p.requires_grad)
            step3.close()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_otter_trainstep_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_otter_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=280) for _ in procs]
    for p in procs:
        p.join(timeout=30)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res
