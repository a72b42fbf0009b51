"""Does splitting the batch into sample groups on separate HIP streams shorten the step?  

The groups are independent except for the scalar `val` (mean over the images of the whole batch), so for a TIMING probe
every group runs the whole chain of stage calls on its own stream with its own buffers:
    python tools/pipeline_probe.py [workload] [groups ...]       e.g.  C3 1 2 7
Per configuration: wall time per step (HIP events around all streams), fwd+bwd.  With `--stagger` the streams are
released one KNN-forward apart, so that the event kernels of group g run beside the KNN of group g + 1."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import ops, LossFactory

args = [a for a in sys.argv[1:] if not a.startswith('--')]
stagger = '--stagger' in sys.argv
wl = dict(bench.WORKLOADS[args[0] if args else 'C3'])
group_counts = [int(a) for a in args[1:]] or [1, 2]
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
cfg = L._cfg
B = traj.shape[0]
evd, trajd, tr = ev.to(dev), traj.to(dev), times[:1].to(dev)


def chain(tj, e, npos, ws_cache):
    Bg = tj.shape[0]
    shape = ops.make_shape(cfg, Bg, e.shape[1], npos, tj.shape[2])
    key = Bg
    if key not in ws_cache:
        ws_cache[key] = ops.alloc_workspace(shape, dev)
    ws = ws_cache[key]
    lut, nxt, state, _ = ops.knn_lut_fwd(cfg, shape, tj, ws)
    g_field = None
    s_nimg = s_C = 0
    if cfg.smooth_weight > 0:
        field, s_nimg, s_C = (nxt, Bg * (cfg.num_bins - 1), 2) if cfg.smooth_on_next else (lut, Bg * cfg.num_bins, 2)
        g_field = ops.lut_smooth(shape, field, s_nimg, s_C, cfg.smooth_weight, ws, True)
    raw = ops.event_splat_fwd(shape, e, lut, tr, ws)
    blur, gimg = ops.contrast_fwd(shape, raw, ws, True)
    scal = ops.finalize(shape, s_nimg, s_C, cfg.smooth_weight, ws, dev)
    g_lut = torch.empty_like(lut)
    ops.event_splat_bwd(shape, e, lut, tr, gimg, scal, None, g_lut, g_field if not cfg.smooth_on_next else None, ws)
    g_next = g_field if cfg.smooth_on_next else None
    return ops.knn_lut_bwd(shape, tj, g_lut, g_next, state, ws)


for ng in group_counts:
    if B % ng:
        continue
    Bg = B // ng
    streams = [torch.cuda.Stream(dev) for _ in range(ng)]
    caches = [dict() for _ in range(ng)]
    parts = [(trajd[g * Bg:(g + 1) * Bg].contiguous(), evd[g * Bg:(g + 1) * Bg].contiguous()) for g in range(ng)]
    res = []
    for it in range(12):
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(torch.cuda.current_stream(dev))
        for st in streams:
            st.wait_event(a)
        prev = None
        for g, st in enumerate(streams):
            with torch.cuda.stream(st):
                if stagger and prev is not None:
                    st.wait_event(prev)
                tj, e = parts[g]
                if stagger:
                    # release the next stream once this group's KNN forward is queued: approximated by an event recorded
                    # right after the first stage call of the chain
                    shape = ops.make_shape(cfg, Bg, e.shape[1], num_pos, tj.shape[2])
                chain(tj, e, num_pos, caches[g])
                mark = torch.cuda.Event(); mark.record(st); prev = mark
        for st in streams:
            e2 = torch.cuda.Event(); e2.record(st)
            torch.cuda.current_stream(dev).wait_event(e2)
        b.record(torch.cuda.current_stream(dev))
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b))
    res = sorted(res[4:])
    print(f'{args[0] if args else "C3"} groups={ng} (B per group {Bg}){" stagger" if stagger else ""}: median {1e3 * res[len(res) // 2]:.0f} us  min {1e3 * res[0]:.0f} us')
