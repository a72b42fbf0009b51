"""Timing of the event-path stages alone at the C3 shape (diagnostics)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import ops, LossFactory, _lib as C
if os.environ.get('MPC_AB_LIB'):          # A/B timing of two builds on the same box
    C.LIB_PATH = os.path.abspath(os.environ['MPC_AB_LIB'])

B = int(sys.argv[1]) if len(sys.argv) > 1 else 14
wl = dict(bench.WORKLOADS['C3']); wl['B'] = B
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
g = torch.Generator().manual_seed(5)
lut = (torch.randn(B, wl['nb'], 120, 160, 1, 2, generator=g) * 2).to(dev)
evd = ev.to(dev); tr = times[:1].to(dev)
for it in range(5):
    lt = lut.clone().requires_grad_(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f, blur, raw = ops.EventFocusFn.apply(lt, evd, tr, L._cfg, num_pos)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    f.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'event fwd {1e6*(t1-t0):.0f} us  bwd {1e6*(t2-t1):.0f} us')
