"""Data-parallel plumbing for training with the CMax loss: batch sharding and a bucketed,
overlappable gradient all-reduce (one process per GPU, torch.distributed; backend 'nccl' is RCCL
over xGMI on MI355X, 'gloo' in CPU tests).

Reference semantics (Lightning DDP behind scripts/flow_training.py:125-130): every rank computes
the loss on ITS OWN local batch, network gradients are AVERAGED across ranks; the loss itself
needs no collective.  The all-reduce below therefore carries the network
gradient (31 044 610 fp32 for the reference UNet) and runs on a side stream so that it overlaps
the next batch's event kernels."""
from __future__ import annotations

import torch
import torch.distributed as dist

UNET_GRAD_NUMEL = 31_044_610   # reference UNet(15 -> 2) parameter count


def shard_indices(num_samples: int, rank: int, world: int):
    """Sample indices of `rank` (DistributedSampler-style round robin, no padding)."""
    return list(range(rank, num_samples, world))


def bucket_bounds(numel: int, n_buckets: int):
    """Split [0, numel) into n_buckets contiguous, 256-element aligned ranges."""
    n_buckets = max(1, min(n_buckets, numel))
    per = -(-numel // n_buckets)
    per = -(-per // 256) * 256
    out, s = [], 0
    while s < numel:
        e = min(s + per, numel)
        out.append((s, e))
        s = e
    return out


class GradAllReducer:
    """Averages a flat gradient buffer across ranks in a few large buckets.

    xGMI is point-to-point (7 links per GPU), so ring collectives are per-link bound: few large
    messages beat DDP's default 25 MB buckets.  `start()` enqueues the all-reduces on a side
    stream (after the producer stream's current work), `wait()` makes the consumer stream wait."""

    def __init__(self, numel: int = UNET_GRAD_NUMEL, n_buckets: int = 4, device=None, group=None,
                 force_collective: bool = False):
        self.device = torch.device(device) if device is not None else torch.device('cpu')
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # force_collective: issue the all-reduce even in a group of one (exercises the RCCL path on a 1-GPU box)
        self.skip = (self.world == 1) and not (force_collective and dist.is_initialized())
        self.flat = torch.zeros(numel, dtype=torch.float32, device=self.device)
        self.bounds = bucket_bounds(numel, n_buckets)
        self.is_cuda = self.device.type == 'cuda'
        self.stream = torch.cuda.Stream(device=self.device) if self.is_cuda else None
        self._works = []

    def start(self):
        if self.skip:
            return
        if self.is_cuda:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                for s, e in self.bounds:
                    self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM,
                                                       group=self.group, async_op=True))
                for w in self._works:
                    w.wait()            # stream-level wait only (no host block) on NCCL/RCCL
                self._works = []
                self.flat.mul_(1.0 / self.world)
        else:
            for s, e in self.bounds:
                self._works.append(dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM,
                                                   group=self.group, async_op=True))

    def wait(self):
        if self.skip:
            return
        if self.is_cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        else:
            for w in self._works:
                w.wait()
            self._works = []
            self.flat.mul_(1.0 / self.world)


def max_over_ranks(value: float, device=None, force_collective: bool = False) -> float:
    """MAX of a host float over all ranks (step-time reduction of the benchmark)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None, force_collective: bool = False) -> float:
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
