"""Per-stage HIP-event times of the C3 step on the time-ordered and on the bucket-ordered event tensor."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import LossFactory, ops

name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
dev = torch.device('cuda:0')
wl = bench.WORKLOADS[name]
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
ob = L.order_events(batch)
t = traj.to(dev).requires_grad_(True)
times = times.to(dev)
for label, b in (('time-ordered', batch), ('bucket-ordered', ob), ('bucket-ordered rows, no table', {'events': ob['events'], 'num_pos_events': num_pos})):
    for _ in range(5):
        l, _, _ = L.calc(t, times, b); l.backward(); t.grad = None
    ops.STAGE_TIMER = ops.StageTimer()
    for _ in range(20):
        l, _, _ = L.calc(t, times, b); l.backward(); t.grad = None
    s = ops.STAGE_TIMER.summary()
    ops.STAGE_TIMER = None
    print(label, {k: round(v['avg_us'], 1) for k, v in s.items()})
