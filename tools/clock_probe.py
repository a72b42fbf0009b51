"""Does a B=1 KNN launch run faster right after heavy work (clock ramp hypothesis)? (diagnostics)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import ops, LossFactory
dev = torch.device('cuda:0')
def setup(B):
    wl = dict(bench.WORKLOADS['C3']); wl['B'] = B
    ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
    L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    shape = ops.make_shape(L._cfg, B, 0, 0, traj.shape[2])
    return L._cfg, shape, traj.to(dev), ops.alloc_workspace(shape, dev)
c1, s1, t1, w1 = setup(1)
c14, s14, t14, w14 = setup(14)
def timed(cfg, shape, t, ws, reps=1):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.knn_lut_fwd(cfg, shape, t, ws)
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps
for _ in range(3):
    timed(c1, s1, t1, w1); timed(c14, s14, t14, w14)
print('B=1 alone, 20 back to back: %.1f us each' % timed(c1, s1, t1, w1, 20))
for _ in range(3):
    for _ in range(10):
        ops.knn_lut_fwd(c14, s14, t14, w14)       # ~5 ms of heavy work, no sync
    print('B=1 right after 10 x B=14: %.1f us' % timed(c1, s1, t1, w1, 1))
print('B=14: %.1f us' % timed(c14, s14, t14, w14, 5))
