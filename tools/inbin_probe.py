"""Statistics of the strip KNN kernel's fast path (needs a library built with -DKS_DEBUG_INBIN, see tools/ubench):
distribution of the number of keys at the K-th distance level and of the candidate slots per query."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from motionpriorcmax_amd import ops, LossFactory, _lib as C
C.LIB_PATH = os.path.abspath(os.environ.get('MPC_AB_LIB', 'tools/ubench/libdbg.so'))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
wl = dict(bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else 'C3']); wl['B'] = B
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
cfg = L._cfg
shape = ops.make_shape(cfg, B, 0, 0, traj.shape[2])
ws = ops.alloc_workspace(shape, dev)
lut, nxt, state, _ = ops.knn_lut_fwd(cfg, shape, traj.to(dev), ws)
torch.cuda.synchronize()
BQ = B * cfg.num_bins * shape.hq * shape.wq
v = state[2 * BQ:3 * BQ].cpu().numpy().reshape(B * cfg.num_bins, shape.hq, shape.wq)
inbin = (v % 100).astype(int); nsl = (v // 100).astype(int)
inner = inbin[:, 8:-8, 8:-8]
print('inner queries: inbin hist', np.bincount(inner.ravel(), minlength=12).tolist())
print('all queries:   inbin hist', np.bincount(inbin.ravel(), minlength=12).tolist())
print('P(inbin > 4) inner %.4f all %.4f ; slots per query: inner mean %.1f max %d, all mean %.1f max %d' % (
    (inner > 4).mean(), (inbin > 4).mean(), nsl[:, 8:-8, 8:-8].mean(), nsl[:, 8:-8, 8:-8].max(), nsl.mean(), nsl.max()))
per_bin = [(inbin[t::cfg.num_bins] > 4).mean() for t in range(cfg.num_bins)]
print('P(inbin > 4) per time bin', np.round(per_bin, 4).tolist())
