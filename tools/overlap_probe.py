"""Would the smoothness kernel run usefully BESIDE the event kernels of the forward?  (They only share their input, the flow
table.)  Times, with HIP events around both streams: k_lut_smooth_march then the event forward on one stream, against the two on
separate streams.  python tools/overlap_probe.py [workload]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from motionpriorcmax_amd import LossFactory, ops  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
wl = bench.WORKLOADS[name]
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
cfg = L._cfg
batch = L.order_events({'events': ev.to(dev), 'num_pos_events': num_pos})
evd = batch['events']
B = traj.shape[0]
shape = ops.make_shape(cfg, B, evd.shape[1], num_pos, traj.shape[2])
ws = ops.alloc_workspace(shape, dev)
ws2 = ops.alloc_workspace(shape, dev)
lut, nxt, state, _ = ops.knn_lut_fwd(cfg, shape, traj.to(dev), ws)
tr = times[:1].to(dev)
field, nimg = (nxt, B * (cfg.num_bins - 1)) if cfg.smooth_on_next else (lut, B * cfg.num_bins)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def run(two):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record(s1)
    if two:
        s2.wait_event(a)
        with torch.cuda.stream(s2):
            ops.lut_smooth(shape, field, nimg, 2, cfg.smooth_weight, ws2, True)
            e2 = torch.cuda.Event(); e2.record(s2)
        with torch.cuda.stream(s1):
            raw = ops.event_splat_fwd(shape, evd, lut, tr, ws)
            ops.contrast_fwd(shape, raw, ws, True)
            s1.wait_event(e2)
            b.record(s1)
    else:
        with torch.cuda.stream(s1):
            ops.lut_smooth(shape, field, nimg, 2, cfg.smooth_weight, ws2, True)
            raw = ops.event_splat_fwd(shape, evd, lut, tr, ws)
            ops.contrast_fwd(shape, raw, ws, True)
            b.record(s1)
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3


for two in (False, True, False, True):
    ts = sorted(run(two) for _ in range(15))
    print(f'{name}: smoothness + event forward + contrast on {"two streams" if two else "one stream "}: median {ts[7]:.1f} us, best {ts[0]:.1f}')
