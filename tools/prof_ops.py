import sys, os
sys.path.insert(0, '/root/repo')
import torch, bench
from motionpriorcmax_amd import LossFactory
wl = bench.WORKLOADS['C2']
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
evd, td = ev.to(dev), times.to(dev)
trajd = traj.to(dev).requires_grad_(True)
batch = {'events': evd, 'num_pos_events': num_pos}
def step():
    loss, _, _ = L.calc(trajd, td, batch)
    loss.backward()
    trajd.grad = None
for _ in range(3): step()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=30, max_name_column_width=60))
