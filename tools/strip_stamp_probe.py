"""Per-workgroup durations of k_knn_strip (diagnostics build -DKS_STAMP: the stamps land in plane 2 of the KNN state)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from motionpriorcmax_amd import ops, LossFactory, _lib as C
C.LIB_PATH = os.path.abspath(os.environ['MPC_AB_LIB'])
B = int(sys.argv[1]); name = sys.argv[2]
wl = dict(bench.WORKLOADS[name]); wl['B'] = B
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
cfg = L._cfg
trajd = traj.to(dev)
shape = ops.make_shape(cfg, B, 0, 0, traj.shape[2])
ws = ops.alloc_workspace(shape, dev)
for it in range(3):
    lut, nxt, state, _ = ops.knn_lut_fwd(cfg, shape, trajd, ws)
torch.cuda.synchronize()
hq, wq, nb = shape.hq, shape.wq, cfg.num_bins
BQ = B * nb * hq * wq
pl = state[2 * BQ:3 * BQ].cpu().numpy().reshape(B * nb, hq, wq)
dur = pl[:, 0, 0::2] / 100.0            # [bt, strips]
pre = pl[:, 0, 1::2] / 100.0
wv = np.stack([pl[:, 1 + w, 0::2] for w in range(4)], -1) / 100.0      # [bt, strips, 4 waves]
n = dur.size
print(f'{n} workgroups; duration us: mean {dur.mean():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}; staging {pre.mean():.1f}')
sx = np.arange(dur.shape[1])
edge = (sx < 2) | (sx >= dur.shape[1] - 2)
print(f'  strips at the left/right border: mean {dur[:, edge].mean():.1f}; others {dur[:, ~edge].mean():.1f}')
print('  wavefronts 0..3 (0 = both image borders): mean', np.round(wv.mean((0, 1)), 1), ' others-only strips:', np.round(wv[:, ~edge].mean((0, 1)), 1))
print(f'  sum of durations / slots: 6 per CU -> {dur.sum() / (256 * 6):.0f} us, 5 -> {dur.sum() / (256 * 5):.0f} us')
