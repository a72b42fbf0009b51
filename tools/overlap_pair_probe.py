#!/usr/bin/env python3
"""Would a bin-group / sample-group PIPELINE shorten the step?  ("overlap what is HBM-bound with what is
VALU-bound": the event kernels of group g beside the KNN kernels of group g + 1 on a second stream.)

A pipeline gains exactly what two UNLIKE stages gain from running side by side, so that is what is measured, in isolation:
two halves of the C3 batch (7 samples each, own workspaces), stage X of one half on stream A, stage Y of the other half on
stream B -- alone, then together from a common start event; `together` is the time until BOTH are done.

    python tools/overlap_pair_probe.py [workload]        (rocprofv3 --kernel-trace -- python3 tools/overlap_pair_probe.py for the timeline)
pairs: KNN forward || event forward (+ contrast);  KNN forward || smoothness;  KNN backward || event backward;  KNN forward || KNN backward
(the control: two LIKE stages)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from motionpriorcmax_amd import LossFactory, ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
wl = dict(bench.WORKLOADS[name])
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
cfg = L._cfg
B = traj.shape[0]
Bg = max(1, B // 2)
tr = times[:1].to(dev)


class Half:
    def __init__(self, lo):
        self.tj = traj[lo:lo + Bg].to(dev).contiguous()
        self.ev = ev[lo:lo + Bg].to(dev).contiguous()
        self.shape = ops.make_shape(cfg, Bg, self.ev.shape[1], num_pos, self.tj.shape[2])
        self.ws = ops.alloc_workspace(self.shape, dev)
        # everything a later stage needs, computed once
        self.lut, self.nxt, self.state, _ = ops.knn_lut_fwd(cfg, self.shape, self.tj, self.ws)
        self.field, self.nimg = (self.nxt, Bg * (cfg.num_bins - 1)) if cfg.smooth_on_next else (self.lut, Bg * cfg.num_bins)
        self.gfield = ops.lut_smooth(self.shape, self.field, self.nimg, 2, cfg.smooth_weight, self.ws, True)
        self.raw = ops.event_splat_fwd(self.shape, self.ev, self.lut, tr, self.ws)
        self.blur, self.gimg = ops.contrast_fwd(self.shape, self.raw, self.ws, True)
        self.scal = ops.finalize(self.shape, self.nimg, 2, cfg.smooth_weight, self.ws, dev)
        self.g_lut = torch.empty_like(self.lut)
        ops.event_splat_bwd(self.shape, self.ev, self.lut, tr, self.gimg, self.scal, None, self.g_lut, None if cfg.smooth_on_next else self.gfield, self.ws)

    def knn_fwd(self):
        ops.knn_lut_fwd(cfg, self.shape, self.tj, self.ws)

    def ev_fwd(self):
        raw = ops.event_splat_fwd(self.shape, self.ev, self.lut, tr, self.ws)
        ops.contrast_fwd(self.shape, raw, self.ws, True)

    def smooth(self):
        ops.lut_smooth(self.shape, self.field, self.nimg, 2, cfg.smooth_weight, self.ws, True)

    def ev_bwd(self):
        ops.event_splat_bwd(self.shape, self.ev, self.lut, tr, self.gimg, self.scal, None, self.g_lut, None if cfg.smooth_on_next else self.gfield, self.ws)

    def knn_bwd(self):
        ops.knn_lut_bwd(self.shape, self.tj, self.g_lut, self.gfield if cfg.smooth_on_next else None, self.state, self.ws)


h0, h1 = Half(0), Half(Bg)
sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def timed(fa, fb):
    """(fa on stream A, fb on stream B; either may be None) -> us until both are done, median of 15."""
    res = []
    for _ in range(19):
        torch.cuda.synchronize()
        st = torch.cuda.Event(enable_timing=True)
        ea = torch.cuda.Event(enable_timing=True); eb = torch.cuda.Event(enable_timing=True)
        st.record(torch.cuda.current_stream(dev))
        sA.wait_event(st); sB.wait_event(st)
        if fa is not None:
            with torch.cuda.stream(sA):
                fa()
        if fb is not None:
            with torch.cuda.stream(sB):
                fb()
        ea.record(sA); eb.record(sB)
        torch.cuda.synchronize()
        res.append(1e3 * max(st.elapsed_time(ea), st.elapsed_time(eb)))
    res = sorted(res[4:])
    return res[len(res) // 2]


pairs = [('KNN forward (half 1)', h1.knn_fwd, 'event forward + contrast (half 0)', h0.ev_fwd),
         ('KNN forward (half 1)', h1.knn_fwd, 'smoothness (half 0)', h0.smooth),
         ('KNN backward (half 0)', h0.knn_bwd, 'event backward (half 1)', h1.ev_bwd),
         ('KNN forward (half 1)', h1.knn_fwd, 'KNN backward (half 0)', h0.knn_bwd)]
print(f'{name}: halves of {Bg} samples; us, median of 15')
for na, fa, nb_, fb in pairs:
    ta, tb, tab = timed(fa, None), timed(None, fb), timed(fa, fb)
    print(f'  {na:24s} {ta:7.1f}   {nb_:34s} {tb:7.1f}   together {tab:7.1f}   sum {ta + tb:7.1f}   gain {ta + tb - tab:6.1f} us ({100 * (ta + tb - tab) / (ta + tb):.0f} %)')
