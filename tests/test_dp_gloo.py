"""Data-parallel plumbing on CPU with the gloo backend, world_size 2 (the N > 1 path of bench.py:
batch sharding, bucketed averaged all-reduce with start()/wait(), max/sum over ranks)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, numel, n_buckets, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from motionpriorcmax_amd import dp
        # 1. shards: every sample exactly once across ranks
        mine = dp.shard_indices(29, rank, world)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        flat = sorted(i for part in gathered for i in part)
        assert flat == list(range(29)), flat
        # 2. bucketed averaged all-reduce, two consecutive steps (start of step k+1 after wait of step k)
        red = dp.GradAllReducer(numel=numel, n_buckets=n_buckets, device='cpu')
        assert red.bounds[0][0] == 0 and red.bounds[-1][1] == numel
        assert all(a[1] == b[0] for a, b in zip(red.bounds[:-1], red.bounds[1:]))
        for step in range(2):
            red.flat.copy_(torch.arange(numel, dtype=torch.float32) * (rank + 1) + step)
            red.start()
            red.wait()
            want = torch.arange(numel, dtype=torch.float32) * (sum(range(1, world + 1)) / world) + step
            assert torch.allclose(red.flat, want), (red.flat[:4], want[:4])
        # 3. step-time and event-count reductions
        assert dp.max_over_ranks(1.0 + rank) == float(world)
        assert dp.sum_over_ranks(10.0 * (rank + 1)) == 10.0 * sum(range(1, world + 1))
        out.put((rank, 'ok'))
    except Exception as e:  # surfaced by the parent
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('numel,n_buckets', [(1000, 4), (31_044_610 // 64, 4)])
def test_grad_allreduce_world2(numel, n_buckets):
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, numel, n_buckets, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def test_bucket_bounds_and_single_process_noop():
    from motionpriorcmax_amd import dp
    b = dp.bucket_bounds(dp.UNET_GRAD_NUMEL, 4)
    assert len(b) == 4 and b[0][0] == 0 and b[-1][1] == dp.UNET_GRAD_NUMEL
    assert all((e - s) % 256 == 0 for s, e in b[:-1])
    red = dp.GradAllReducer(numel=100, n_buckets=3, device='cpu')   # no process group: world = 1
    red.flat.fill_(2.0)
    red.start()
    red.wait()
    assert torch.all(red.flat == 2.0)
    assert dp.max_over_ranks(3.5) == 3.5
    assert dp.shard_indices(5, 0, 1) == [0, 1, 2, 3, 4]


def _producer_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from motionpriorcmax_amd import dp
        torch.manual_seed(0)                             # the same weights on every rank (DDP broadcasts them), its own input
        red = dp.GradAllReducer(device='cpu')
        fired = []
        start_bucket = red.start_bucket
        red.start_bucket = lambda i: (fired.append(i), start_bucket(i))
        prod = dp.OverlappedGradProducer(red, rows=16, seed=100 + rank)
        assert sum(p.numel() for p in prod.net.parameters()) == dp.UNET_GRAD_NUMEL == red.flat.numel()
        # this rank's own gradient, without the exchange -- ALREADY divided by the world size: the producer scales the network's
        # output by 1 / world (as DDP divides in its hook), so that the exchange is a plain sum with no pass of its own to divide
        prod.net.zero_grad(set_to_none=False)
        red.skip = True
        prod.step()
        own = red.flat.clone()
        assert float(own.abs().sum()) > 0
        red.skip = False
        for _ in range(2):                               # two steps: the counters re-arm
            fired.clear()
            prod.step()
            prod.wait()
            assert fired == list(range(len(red.bounds))), fired      # every bucket once, in production order
            both = [torch.empty_like(own) for _ in range(world)]
            dist.all_gather(both, own)
            want = sum(both)                                  # = the mean over the ranks of the undivided gradients
            assert torch.allclose(red.flat, want, rtol=1e-5, atol=1e-8), float((red.flat - want).abs().max())
            # the parameters' .grad ARE the flat buffer (views): what the optimizer reads is the averaged gradient
            p0 = next(iter(prod.net.parameters()))
            assert p0.grad.data_ptr() >= red.flat.data_ptr() and p0.grad.data_ptr() < red.flat.data_ptr() + 4 * red.flat.numel()
        out.put((rank, 'ok'))
    except Exception as e:  # surfaced by the parent
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_overlapped_grad_producer_world2():
    """The stand-in network (31 044 610 parameters, the reference UNet's count) fills the reducer's flat buffer through gradient
    views; every bucket is all-reduced once, as soon as its last gradient is written; the result is the mean over the ranks."""
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_producer_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res
