"""Event-axis sharding (SURVEY.md 8e "optional finer split", dp.event_sharded_calc) on CPU over gloo, world_size 2.

No HIP kernel runs here: the library's stage calls are replaced by `_OracleStages`, an emulation built from the CPU oracle
(test infrastructure) that keeps the one property the design rests on -- the raw IWE of a shard is a sum of Q33.30 INTEGER
taps -- so that what the test checks is the host logic of dp.py: the row sharding (every event exactly once, polarity blocks
kept), the int64 all-reduce before the blur, the fp32 all-reduce of dL/dLUT, the smoothness gradient added once.  The
all-reduced image must equal the single-rank image BIT FOR BIT; loss and gradient must match the unsharded oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

FIX = 30


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


class _OracleStages:
    """dp._HipStages with the oracle's arithmetic (float64 taps rounded once to Q33.30, as the HIP accumulators hold them)."""

    def __init__(self, cfg, B, M, Mp, n, device):
        from oracle import focus_oracle as O
        self.O, self.cfg, self.Mp = O, cfg, Mp

    def knn_fwd(self, traj):
        c = self.cfg
        lut, nxt = self.O.interpolate_flow(traj[:, :1], traj[:, 1:], c.image_shape, c.sp, c.num_knn, 'l1' if c.dist_l1 else 'l2',
                                           'iwd' if c.scheme_iwd else 'mean', c.smooth_on_next and c.smooth_weight > 0)
        self.traj = traj
        return lut.detach(), None if nxt is None else nxt.detach(), None

    def _field(self, field):
        f = field.permute(0, 1, 4, 5, 2, 3)
        return f.reshape(-1, f.shape[3], f.shape[4], f.shape[5])

    def smooth(self, field, nimg, C_, want_grad):
        with torch.enable_grad():          # (the stages are called inside autograd.Function.forward / backward: grad mode is off there)
            f = field.detach().clone().requires_grad_(True)
            self.smooth_val = self.cfg.smooth_weight * self.O.smoothness(self._field(f))
            return torch.autograd.grad(self.smooth_val, f)[0] if want_grad else None

    def _raw(self, ev, lut, tr):
        c = self.cfg
        warped = self.O.warp_events(ev, lut, c.sp)
        _, raw = self.O.make_iwes(ev, warped, tr, c.image_shape, c.scale_by_dt, c.mask_border, c.polarity_split, self.Mp)
        return raw

    def splat_fixed(self, ev, lut, tr):
        # integer taps: every tap value rounded to Q33.30 BEFORE it is added (what ds_add_u64 accumulates)
        c, O = self.cfg, self.O
        h, w = c.image_shape
        warped = O.warp_events(ev, lut, c.sp)
        wts = O.event_weights(ev, warped, tr, c.image_shape, c.scale_by_dt, c.mask_border)
        b = ev.shape[0]
        out = torch.zeros((b, 2, h * w), dtype=torch.int64)
        pos, wt = warped[:, 0], wts[:, 0]
        fl = torch.floor(pos + 1e-6); fr = pos - fl; fl = fl.long()
        y0, x0, fy, fx = fl[..., 0], fl[..., 1], fr[..., 0], fr[..., 1]
        pol = (torch.arange(ev.shape[1]) >= self.Mp).long()[None].expand(b, -1)
        for yy, xx, v in ((y0, x0, (1 - fy) * (1 - fx) * wt), (y0 + 1, x0, fy * (1 - fx) * wt),
                          (y0, x0 + 1, (1 - fy) * fx * wt), (y0 + 1, x0 + 1, fy * fx * wt)):
            ok = (0 <= xx) & (xx < w) & (0 <= yy) & (yy < h)
            q = torch.round(v.double() * (1 << FIX)).long() * ok
            idx = ((xx + yy * w) * ok).long()
            for p_ in (0, 1):
                out[:, p_].scatter_add_(1, idx, q * (pol == p_))
        return out.reshape(b, 2, h, w)

    def from_fixed(self, fixed):
        return (fixed.double() / (1 << FIX)).float()

    def contrast(self, raw, want_grad):
        with torch.enable_grad():
            r = raw.detach().clone().requires_grad_(True)
            blur = self.O.gaussian_blur3(r)
            val = self.O.contrast_value(blur, 'gradient_magnitude', 'l2' if self.cfg.norm_l2 else 'l1')
            self.focus = 1 / val
            gimg = torch.autograd.grad(self.focus, r)[0] if want_grad else None        # d focus / d raw (already scaled)
        return blur.detach(), gimg

    def finalize(self, nimg, C_, device):
        s = torch.zeros(8)
        sm = float(getattr(self, 'smooth_val', 0.0))
        s[0], s[1], s[2] = float(self.focus) + sm, float(self.focus), sm
        return s

    def splat_bwd(self, ev, lut, tr, gimg, scal, g):
        with torch.enable_grad():
            lt = lut.detach().clone().requires_grad_(True)
            raw = self._raw(ev, lt, tr)                       # this rank's rows
            return torch.autograd.grad(raw, lt, grad_outputs=gimg * g)[0]

    def scale(self, x, a):
        return x * a

    def knn_bwd(self, traj, g_lut, g_next, state):
        c = self.cfg
        with torch.enable_grad():
            t = traj.detach().clone().requires_grad_(True)
            lut, nxt = self.O.interpolate_flow(t[:, :1], t[:, 1:], c.image_shape, c.sp, c.num_knn, 'l1' if c.dist_l1 else 'l2',
                                               'iwd' if c.scheme_iwd else 'mean', g_next is not None)
            outs, grads = [lut], [g_lut]
            if g_next is not None:
                outs.append(nxt); grads.append(g_next)
            return torch.autograd.grad(outs, t, grad_outputs=grads)[0]


def _case(smooth_type):
    from oracle import focus_oracle as O
    shape, B, M, nb, K = (48, 64), 2, 3000, 5, 4
    cfg = dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=K, smooth_weight=0.05, lut_superpixel_size=4,
               focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True, polarity_aware_batching=True,
               interpolation_scheme='mean', smooth_type=smooth_type)
    ev, num_pos = O.synth_events(B, M, shape, nb, seed=3, pad_frac=0.03, num_pos=M // 2 + 11)
    g = torch.Generator().manual_seed(5)
    coeff = torch.randn(B, 1, 2, *shape, generator=g) * 2.0
    times = torch.cat((torch.tensor([0.37]), O.bin_mid_times(nb)))
    traj = O.trajectories_at(coeff, times, O.tile_mask(shape, 4), 1, 'polynomial').detach()
    return cfg, ev, num_pos, traj, times


def _worker(rank, world, port, smooth_type, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from motionpriorcmax_amd import dp
        from motionpriorcmax_amd.losses.focus import FocusLoss
        from oracle import focus_oracle as O
        torch.set_num_threads(2)
        cfg, ev, num_pos, traj, times = _case(smooth_type)
        L = FocusLoss(**cfg)                                   # host object only: no kernel is called through it here
        # 1. the shards: every row exactly once, positive rows first
        ev_l, np_l = dp.shard_event_rows(ev, num_pos, rank, world)
        assert (ev_l[:, :np_l, 3] == ev[:, rank:num_pos:world, 3]).all()
        cnt = torch.tensor([ev_l.shape[1], np_l], dtype=torch.int64)
        dist.all_reduce(cnt)
        assert cnt.tolist() == [ev.shape[1], num_pos]
        # 2. sharded loss + gradient
        t = traj.clone().requires_grad_(True)
        loss, log, misc = dp.event_sharded_calc(L, t, times, {'events': ev_l, 'num_pos_events': np_l}, stages_factory=_OracleStages)
        loss.backward()
        # 3. the single-rank run of the same emulation: the image must be bit-identical, loss and gradient equal
        t1 = traj.clone().requires_grad_(True)
        dist_backup = dist.all_reduce
        loss1, log1, misc1 = None, None, None
        K = _OracleStages(L._cfg, ev.shape[0], ev.shape[1], num_pos, traj.shape[2], 'cpu')
        lut, nxt, _ = K.knn_fwd(traj)
        full = K.from_fixed(K.splat_fixed(ev, lut, times[:1]))
        blur_full, _ = K.contrast(full, False)
        assert torch.equal(misc['iwes'].reshape(blur_full.shape), blur_full), 'all-reduced IWE differs from the single-rank one'
        # ... and the unsharded oracle (float accumulation): loss to 1e-6, gradient to 1e-4 relative L2
        lo, _, _ = O.FocusLossOracle(**cfg).calc(t1, times, {'events': ev, 'num_pos_events': num_pos})
        lo.backward()
        assert abs(loss.item() - lo.item()) <= 1e-6 * abs(lo.item()), (loss.item(), lo.item())
        rel = float((t.grad - t1.grad).norm() / t1.grad.norm())
        assert rel < 1e-4, rel
        # every rank holds the same gradient
        gsum = t.grad.clone(); dist.all_reduce(gsum)
        assert torch.allclose(gsum, world * t.grad, rtol=0, atol=1e-6 * float(t.grad.abs().max()))
        out.put((rank, 'ok'))
    except Exception as e:  # surfaced by the parent
        import traceback
        out.put((rank, repr(e) + traceback.format_exc()[-800:]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('smooth_type', ['on_flow_to_tref', 'on_flow_to_next'])
def test_event_axis_shard_world2(smooth_type):
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, smooth_type, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, 'ok'), (1, 'ok')], res


def test_shard_event_rows_partitions_every_block():
    from motionpriorcmax_amd import dp
    ev = torch.arange(2 * 11 * 6, dtype=torch.float32).reshape(2, 11, 6)
    seen = []
    for r in range(3):
        e, npos = dp.shard_event_rows(ev, 4, r, 3)
        assert e.shape[1] == len(range(r, 4, 3)) + len(range(4 + r, 11, 3)) and npos == len(range(r, 4, 3))
        seen.append(e[0, :, 0])
    allrows = torch.sort(torch.cat(seen)).values
    assert torch.equal(allrows, ev[0, :, 0])
    e, npos = dp.shard_event_rows(ev, -1, 1, 2)            # no polarity batching
    assert npos == -1 and torch.equal(e, ev[:, 1::2])
