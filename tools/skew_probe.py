"""Event-path time when the events are spatially concentrated (a few buckets hold most records: the tail of the busiest workgroups) (diagnostics)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import ops, LossFactory
B = 14
wl = dict(bench.WORKLOADS['C3']); wl['B'] = B
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
g = torch.Generator().manual_seed(5)
lut = (torch.randn(B, wl['nb'], 120, 160, 1, 2, generator=g) * 2).to(dev)
tr = times[:1].to(dev)
for frac, rows in ((0.0, 480), (0.5, 120), (0.5, 30), (0.9, 30), (1.0, 8)):
    e = ev.clone()
    if frac > 0:
        m = torch.rand(e.shape[:2], generator=g) < frac
        e[..., 0] = torch.where(m, 200 + torch.rand(e.shape[:2], generator=g) * rows, e[..., 0])
    evd = e.to(dev)
    ts = []
    for it in range(6):
        lt = lut.clone().requires_grad_(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f, blur, raw = ops.EventFocusFn.apply(lt, evd, tr, L._cfg, num_pos)
        f.backward()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f'{int(frac*100):3d}% of the events in a band of {rows:3d} rows: event path fwd+bwd {1e3*sorted(ts)[2]:.3f} ms', flush=True)
