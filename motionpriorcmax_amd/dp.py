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
                 force_collective: bool = False, prescaled: bool = False):
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
        self._scale_pending = False
        # The division by the world size.  Round 5: a `mul_` pass over every bucket behind its all-reduce, on the side stream -- 124 MB
        # read and written beside the loss kernels.  Round 6 tried ReduceOp.AVG instead (RCCL builds a pre-multiplied sum per call):
        # the one-rank leg of the benchmark went from +0.26 to +0.90 ms per step -- worse.  Now the PRODUCER pre-scales: a caller
        # whose gradients arrive already divided by the world size (OverlappedGradProducer scales the network's output, as DDP
        # divides in its hook) sets `prescaled` and the exchange is the SUM alone.
        self.avg = bool(prescaled)          # "nothing left to divide": the buffer holds gradients / world already
        self.op = dist.ReduceOp.SUM

    def start(self):
        if self.skip:
            return
        if self.is_cuda:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.stream):
                for s, e in self.bounds:
                    self._works.append(dist.all_reduce(self.flat[s:e], op=self.op,
                                                       group=self.group, async_op=True))
                for w in self._works:
                    w.wait()            # stream-level wait only (no host block) on NCCL/RCCL
                self._works = []
                if not self.avg:
                    self.flat.mul_(1.0 / self.world)
        else:
            for s, e in self.bounds:
                self._works.append(dist.all_reduce(self.flat[s:e], op=self.op,
                                                   group=self.group, async_op=True))
            self._scale_pending = not self.avg

    def start_bucket(self, i: int):
        """Enqueue the all-reduce of bucket i alone, behind what the producing (current) stream has enqueued so far: the
        producer calls this the moment the bucket's last gradient has been written (OverlappedGradProducer)."""
        if self.skip:
            return
        s, e = self.bounds[i]
        if self.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                w = dist.all_reduce(self.flat[s:e], op=self.op, group=self.group, async_op=True)
                w.wait()                # stream-level wait only
                if not self.avg:
                    self.flat[s:e].mul_(1.0 / self.world)
        else:
            dist.all_reduce(self.flat[s:e], op=self.op, group=self.group)
            if not self.avg:
                self.flat[s:e].mul_(1.0 / self.world)

    def wait(self):
        if self.skip:
            return
        if self.is_cuda:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)
        else:
            for w in self._works:
                w.wait()
            self._works = []
            if self._scale_pending:          # (start(): the whole buffer at once; start_bucket() scales its own bucket)
                self.flat.mul_(1.0 / self.world)
                self._scale_pending = False


class StandInNetwork(torch.nn.Module):
    """A gradient PRODUCER of the reference network's size for the data-parallel leg of the benchmark: a plain stack of dense layers
    (ReLU between them) with exactly UNET_GRAD_NUMEL = 31 044 610 fp32 parameters -- the parameter count of the reference's
    UNet(15 -> 2) (src/models/unet/unet_model.py:6, src/modules/trajectory_net.py:27-28) -- so that its backward writes 124 MB of
    gradients through rocBLAS / hipBLASLt kernels that compete with the loss kernels for CUs and HBM while the buckets of the
    all-reduce fly.  NOT the reference's architecture (the UNet itself is out of scope; dense layers instead of
    convolutions so that a fresh box pays no MIOpen kernel search inside the benchmark): a stand-in of the same parameter and
    gradient volume on a small input, so that the benchmark step stays short."""
    WIDTHS = (1024, 4096, 4096, 2400, 96, 2)

    def __init__(self):
        super().__init__()
        layers = []
        for i, (a, b) in enumerate(zip(self.WIDTHS[:-1], self.WIDTHS[1:])):
            layers.append(torch.nn.Linear(a, b))
            if i < len(self.WIDTHS) - 2:
                layers.append(torch.nn.ReLU(inplace=True))
        self.body = torch.nn.Sequential(*layers)
        have = sum(p.numel() for p in self.parameters())
        self.pad = torch.nn.Parameter(torch.zeros(UNET_GRAD_NUMEL - have))         # (1 408 values: the count is the UNet's exactly)
        assert sum(p.numel() for p in self.parameters()) == UNET_GRAD_NUMEL

    def forward(self, x):
        return self.body(x).mean() + (self.pad * self.pad).sum()


class OverlappedGradProducer:
    """The network side of a data-parallel training step around the loss: `step()` runs the stand-in network's forward and
    backward; every parameter's gradient is a VIEW into the GradAllReducer's flat buffer (as DDP's gradient_as_bucket_view),
    and the moment the last gradient of a bucket has been written (post-accumulate hooks; backward produces them last layer
    first) that bucket's all-reduce is enqueued on the reducer's side stream behind an event on the producing stream --
    buckets are exchanged while the rest of the backward, and then the next step's loss, still run.  `wait()` before the
    optimizer point.  Reference: Lightning DDP behind scripts/flow_training.py:125-130 (bucketed all-reduce during backward)."""

    def __init__(self, reducer: GradAllReducer, rows: int = 512, seed: int = 0):
        self.r = reducer
        reducer.avg = True                        # (step() scales the network's output by 1 / world: the gradients arrive divided)
        dev = reducer.device
        g = torch.Generator().manual_seed(seed)
        self.net = StandInNetwork().to(dev)
        self.x = torch.randn(rows, StandInNetwork.WIDTHS[0], generator=g).to(dev)
        # parameters in the order their gradients are produced (last layer first), laid out back to back in the flat buffer
        params = list(self.net.parameters())[::-1]
        off = 0
        self._bucket_of, self._left0 = {}, [0] * len(reducer.bounds)
        for p_ in params:
            n = p_.numel()
            p_.grad = reducer.flat[off:off + n].view_as(p_)
            # every bucket this parameter's gradient overlaps waits for it
            over = [i for i, (s, e) in enumerate(reducer.bounds) if s < off + n and off < e]
            self._bucket_of[p_] = over
            for i in over:
                self._left0[i] += 1
            p_.register_post_accumulate_grad_hook(self._hook)
            off += n
        assert off == reducer.flat.numel()
        self._left = list(self._left0)
        self._fired = 0

    def _hook(self, p_):
        for b in self._bucket_of[p_]:
            self._left[b] -= 1
        # buckets complete in order (the parameters are laid out in production order): fire every bucket that is now whole
        while self._fired < len(self._left) and self._left[self._fired] == 0:
            self.r.start_bucket(self._fired)
            self._fired += 1

    def step(self):
        self._left = list(self._left0)
        self._fired = 0
        # (the previous step's buckets may still be in flight on the reducer's stream if the caller did not wait(): zeroing the
        # buffer under them would corrupt the exchange silently -- the producing stream waits for them first; free if wait() ran)
        self.r.wait()
        self.r.flat.zero_()                       # (gradients accumulate into the views: the optimizer's zero_grad)
        (self.net(self.x) * (1.0 / self.r.world)).backward()
        while self._fired < len(self._left):      # (buckets without a parameter of their own end)
            self.r.start_bucket(self._fired)
            self._fired += 1

    def wait(self):
        self.r.wait()


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


# ------------------------------------------------------------------------------------------
# Event-axis sharding: a batch smaller than the number of ranks ("optional finer split"; the reference has
# batch DDP only, scripts/flow_training.py:125-128).  Every rank holds ALL trajectories and 1/world of every sample's event
# rows.  Exchange steps: one all-reduce(SUM) of the raw IWE as int64 Q33.30 accumulators before the blur (integer sums:
# the image, and with it the loss, is bit for bit the single-rank one), one all-reduce(SUM) of dL/dLUT in the backward.
# ------------------------------------------------------------------------------------------
def _all_reduce_sum(t: torch.Tensor, group=None):
    """all-reduce(SUM) in place.  RCCL takes the device tensor as it is; the gloo debugging backend (ranks sharing one GPU,
    CPU tests) gets a host copy of a device tensor."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == 'gloo':
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def shard_event_rows(events: torch.Tensor, num_pos: int, rank: int, world: int):
    """Rows of `events` [B, M, 6] this rank warps: every world-th row of the positive block, then every world-th row of the
    negative block (the polarity split is by row index, focus.py:216-227).  Returns (local events, local num_pos)."""
    M = events.shape[1]
    if num_pos < 0:
        return events[:, rank:M:world].contiguous(), -1
    pos = events[:, rank:num_pos:world]
    neg = events[:, num_pos + rank:M:world]
    return torch.cat((pos, neg), dim=1).contiguous(), pos.shape[1]


class _HipStages:
    """The stage calls of the library (ops.py) the sharded loss is built from."""

    def __init__(self, cfg, B, M, Mp, n, device):
        from . import ops
        self.ops, self.cfg = ops, cfg
        self.shape = ops.make_shape(cfg, B, M, Mp, n)
        self.ws = ops.alloc_workspace(self.shape, device)

    def knn_fwd(self, traj):
        lut, nxt, state, _ = self.ops.knn_lut_fwd(self.cfg, self.shape, traj, self.ws)
        return lut, nxt, state

    def smooth(self, field, nimg, C_, want_grad):
        return self.ops.lut_smooth(self.shape, field, nimg, C_, self.cfg.smooth_weight, self.ws, want_grad)

    def splat_fixed(self, ev, lut, tr):
        return self.ops.event_splat_fwd_fixed(self.shape, ev, lut, tr, self.ws)

    def from_fixed(self, fixed):
        return self.ops.iwe_from_fixed(fixed)

    def contrast(self, raw, want_grad):
        return self.ops.contrast_fwd(self.shape, raw, self.ws, want_grad)

    def finalize(self, nimg, C_, device):
        return self.ops.finalize(self.shape, nimg, C_, self.cfg.smooth_weight, self.ws, device)

    def splat_bwd(self, ev, lut, tr, gimg, scal, g):
        g_lut = torch.empty_like(lut)
        self.ops.event_splat_bwd(self.shape, ev, lut, tr, gimg, scal, g, g_lut, None, self.ws)
        return g_lut

    def scale(self, x, a):
        return self.ops.scale(x, a)

    def knn_bwd(self, traj, g_lut, g_next, state):
        return self.ops.knn_lut_bwd(self.shape, traj, g_lut, g_next, state, self.ws)


class EventShardedFocusFn(torch.autograd.Function):
    """FocusLoss.calc (reference focus.py:66-113) with the EVENT axis sharded over the ranks of `group`."""

    @staticmethod
    def forward(ctx, trajectories, events_local, t_ref, cfg, num_pos_local, group, stages_factory):
        traj = trajectories.detach().float().contiguous()
        ev = events_local.detach().float().contiguous()
        tr = t_ref.detach().float().to(traj.device).contiguous()
        B, M = ev.shape[0], ev.shape[1]
        Mp = num_pos_local if cfg.polarity_split else M
        need_grad = trajectories.requires_grad
        K = (stages_factory or _HipStages)(cfg, B, M, Mp, traj.shape[2], traj.device)
        lut, nxt, state = K.knn_fwd(traj)                          # replicated: every rank has all trajectories
        g_field, s_nimg, s_C = None, 0, 0
        if cfg.smooth_weight > 0:
            field, s_nimg, s_C = (nxt, B * (cfg.num_bins - 1), 2) if cfg.smooth_on_next else (lut, B * cfg.num_bins, 2 * cfg.num_tref)
            if s_nimg > 0:
                g_field = K.smooth(field, s_nimg, s_C, need_grad)
        fixed = K.splat_fixed(ev, lut, tr)                         # this rank's rows only
        _all_reduce_sum(fixed, group)                              # exchange step 1: int64, exact
        raw = K.from_fixed(fixed)
        blur, gimg = K.contrast(raw, need_grad)
        scal = K.finalize(s_nimg, s_C, traj.device)
        ctx.K, ctx.cfg, ctx.group = K, cfg, group
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(traj, ev, tr, lut, state, gimg, scal, g_field)
        out = scal[:3].clone()
        loss, focus, smooth = out[0], out[1], out[2]
        ctx.mark_non_differentiable(focus, smooth, blur)
        return loss, focus, smooth, blur

    @staticmethod
    def backward(ctx, g_loss, g_focus, g_smooth, g_iwes):
        if g_loss is None:
            return (None,) * 7
        K, cfg = ctx.K, ctx.cfg
        traj, ev, tr, lut, state, gimg, scal, g_field = ctx.saved_tensors
        g = g_loss.reshape(1).float().contiguous()
        g_lut = K.splat_bwd(ev, lut, tr, gimg, scal, g)            # partial: this rank's rows
        _all_reduce_sum(g_lut, ctx.group)                          # exchange step 2
        g_next = None
        if g_field is not None:                                    # the smoothness gradient is replicated: added once, after the sum
            if cfg.smooth_on_next:
                g_next = K.scale(g_field, g)
            else:
                g_lut = g_lut + K.scale(g_field, g)
        return K.knn_bwd(traj, g_lut, g_next, state), None, None, None, None, None, None


def event_sharded_calc(loss_obj, trajectories, times, local_batch, group=None, stages_factory=None):
    """`loss_obj.calc(trajectories, times, batch)` for a batch whose EVENT rows are sharded over the ranks (local_batch from
    shard_event_rows): same return contract as FocusLoss.calc; loss, focus term and IWEs equal the unsharded ones bit for
    bit, the gradient up to the fp32 rounding of one sum over ranks."""
    cfg = loss_obj._cfg
    assert cfg.num_tref == 1, 'event-axis sharding serves num_tref == 1 (the shipped configurations)'
    num_pos = int(local_batch['num_pos_events']) if 'num_pos_events' in local_batch else -1
    loss, focus, smooth, iwes = EventShardedFocusFn.apply(trajectories, local_batch['events'], times[:cfg.num_tref], cfg, num_pos,
                                                          group, stages_factory)
    h, w = cfg.image_shape
    b = local_batch['events'].shape[0]
    iwes = iwes.reshape(b, 1, 2, h, w) if cfg.polarity_split else iwes.reshape(b, 1, h, w)
    return loss, {'focus_loss': focus.detach(), 'smoothness_loss': smooth.detach()}, {'iwes': iwes.detach()}
