"""Host-side cost of one eager step (calc + backward) at B = 1 (C2): cProfile of 200 steps."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import LossFactory

name = sys.argv[1] if len(sys.argv) > 1 else 'C2'
dev = torch.device('cuda:0')
wl = bench.WORKLOADS[name]
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
t = traj.to(dev).requires_grad_(True)
times = times.to(dev)


def step():
    l, _, _ = L.calc(t, times, batch)
    l.backward()
    t.grad = None


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host issue {1e6 * (t1 - t0) / 200:.1f} us/step, with drain {1e6 * (t2 - t0) / 200:.1f} us/step')
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(22)
# forward and backward separately
t0 = time.perf_counter(); fw = 0.0
for _ in range(200):
    a = time.perf_counter(); l, _, _ = L.calc(t, times, batch); fw += time.perf_counter() - a
    l.backward(); t.grad = None
tot = time.perf_counter() - t0
print(f'calc {1e6 * fw / 200:.1f} us, backward {1e6 * (tot - fw) / 200:.1f} us (host issue, per step)')
