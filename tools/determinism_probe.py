import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
from motionpriorcmax_amd import ops, LossFactory
wl = dict(bench.WORKLOADS['C3']); wl['B'] = 4
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
t = traj.to(dev)
ref = None
nd = 0
for i in range(10):
    lut, _ = ops.KnnLutFn.apply(t, L._cfg)
    if ref is None: ref = lut.clone()
    else: nd += int((lut != ref).sum())
print('differing LUT entries over 9 repeats:', nd)
tt = t.clone().requires_grad_(True)
batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
gs = []
for i in range(4):
    tt.grad = None
    loss, _, _ = L.calc(tt, times.to(dev), batch); loss.backward()
    gs.append((loss.item(), tt.grad.clone()))
print('loss equal:', all(g[0] == gs[0][0] for g in gs), 'grad equal:', all(torch.equal(g[1], gs[0][1]) for g in gs))
