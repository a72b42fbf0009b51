"""Per-step wall time of the first steps of a fresh process (diagnostics for one-off stalls)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import LossFactory
wl = bench.WORKLOADS['C3']
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
t = traj.to(dev).requires_grad_(True)
batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
td = times.to(dev)
torch.cuda.synchronize()
out = []
for i in range(40):
    t0 = time.perf_counter()
    loss, _, _ = L.calc(t, td, batch); loss.backward(); t.grad = None
    torch.cuda.synchronize()
    out.append(1e3 * (time.perf_counter() - t0))
print(' '.join(f'{x:.2f}' for x in out))
print('reserved MB', torch.cuda.memory_reserved() / 1e6)
