"""Time of FocusLoss.order_events (mpc_event_bucket_order) per workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import LossFactory
dev = torch.device('cuda:0')
for name in ('C3', 'C2', 'C4'):
    wl = bench.WORKLOADS[name]
    ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
    L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
    for _ in range(3):
        L.order_events(batch)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        L.order_events(batch)
    torch.cuda.synchronize()
    print(name, f'{1e6 * (time.perf_counter() - t0) / 50:.1f} us per batch')
