"""Timing + fallback statistics of the KNN LUT kernels at the C3 shape (diagnostics)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import ops, LossFactory, _lib as C
if os.environ.get('MPC_AB_LIB'):          # A/B timing of two builds on the same box
    C.LIB_PATH = os.path.abspath(os.environ['MPC_AB_LIB'])

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wl = dict(bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else 'C3'])
wl['B'] = B
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
cfg = L._cfg
trajd = traj.to(dev)
shape = ops.make_shape(cfg, B, 0, 0, traj.shape[2])
ws = ops.alloc_workspace(shape, dev)
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lut, nxt, state, _ = ops.knn_lut_fwd(cfg, shape, trajd, ws)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    g = torch.randn_like(lut)
    gn = torch.randn_like(nxt) if nxt is not None else None
    torch.cuda.synchronize(); t2 = time.perf_counter()
    gt = ops.knn_lut_bwd(shape, trajd, g, gn, state, ws)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f'fwd {1e6*(t1-t0):.0f} us  bwd {1e6*(t3-t2):.0f} us')
