"""Phases of the strip workgroups of k_knn_strip's MAIN launch (diagnostics build -DKS_STAMP0, csrc/diag/stamps.h: the stamps land far
inside the KNN forward's `fail` list):  MPC_AB_LIB=build_ab/ksstamp0.so python tools/strip_stamp_probe.py [B] [workload] [family]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from motionpriorcmax_amd import ops, LossFactory, _lib as C
from motionpriorcmax_amd.utils import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 14
name = sys.argv[2] if len(sys.argv) > 2 else 'C3'
fam = sys.argv[3] if len(sys.argv) > 3 else 'white'
wl = dict(bench.WORKLOADS[name]); wl['B'] = B
if fam == 'white':
    _, _, traj, _ = bench.synth_inputs(wl, seed=1)
else:
    traj, _ = synth.synth_trajectories(B, 3, wl['nb'], (bench.H, bench.W), bench.PATCH, fam, seed=11)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
cfg = L._cfg
trajd = traj.to(dev)
shape = ops.make_shape(cfg, B, 0, 0, traj.shape[2])
ws = ops.alloc_workspace(shape, dev)
for it in range(3):
    ops.knn_lut_fwd(cfg, shape, trajd, ws)
torch.cuda.synchronize()
off = C.lib().mpc_knn_fail_list_offset(ctypes.byref(shape))
gx, gy = -(-shape.wq // 2), -(-shape.hq // 128)
nblk = (gx * gy * B * cfg.num_bins + 7) // 8 * 8
raw = ws[off + 4 * 200001: off + 4 * 200001 + 48 * nblk].view(torch.int32).cpu().numpy().reshape(nblk, 12).astype(np.int64)
ok = raw[:, 0] > 0
st = raw[ok]
ph = st[:, 1:5] / 100.0                       # us after the workgroup's start: radius done, row table, staged, end
names = ['radius from the table', 'row table', 'staging', 'search + lists']
prev = np.zeros(len(st))
print(f'{fam}: {ok.sum()} strip workgroups stamped; lifetime us: mean {ph[:, 3].mean():.1f} median {np.median(ph[:, 3]):.1f} p90 {np.percentile(ph[:, 3], 90):.1f} max {ph[:, 3].max():.1f}')
for k, nm in enumerate(names):
    d = ph[:, k] - prev
    print(f'  {nm:24s} mean {d.mean():5.2f}  median {np.median(d):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():6.2f}')
    prev = ph[:, k]
t0 = st[:, 0].min()
start = ((st[:, 0] - t0) % (1 << 31)) / 100.0
print(f'  launch: first start 0, last start {start.max():.1f} us, last end {(start + ph[:, 3]).max():.1f} us; sum of lifetimes / (256 CUs x 6) = {ph[:, 3].sum() / 1536:.0f} us')
