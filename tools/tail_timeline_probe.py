#!/usr/bin/env python3
"""Timeline of one k_knn_tail launch (diagnostics build -DKT_TIMELINE, csrc/diag/stamps.h): when its strip workgroups and its
one-wavefront-per-query workgroups start, finish their first list, stop waiting and end.
    MPC_EXTRA_HIPCC_FLAGS=-DKT_TIMELINE python motionpriorcmax_amd/build.py && python tools/tail_timeline_probe.py unet
(or a diagnostics library beside the product one: MPC_AB_LIB=...; restore the product build afterwards)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from motionpriorcmax_amd import LossFactory, ops, _lib as C  # noqa: E402
from motionpriorcmax_amd.utils import synth  # noqa: E402
if os.environ.get('MPC_AB_LIB'):
    C.LIB_PATH = os.path.abspath(os.environ['MPC_AB_LIB'])

fam = sys.argv[1] if len(sys.argv) > 1 else 'unet'
wl = bench.WORKLOADS['C3']
B = wl['B']
if fam == 'white':
    _, _, traj, _ = bench.synth_inputs(wl, seed=1)
else:
    traj, _ = synth.synth_trajectories(B, 3, wl['nb'], (bench.H, bench.W), bench.PATCH, fam, seed=11)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
shape = ops.make_shape(L._cfg, B, 0, 0, traj.shape[2])
ws = ops.alloc_workspace(shape, dev)
trajd = traj.to(dev)
for _ in range(3):
    ops.knn_lut_fwd(L._cfg, shape, trajd, ws)
torch.cuda.synchronize()
off = C.lib().mpc_knn_fail_list_offset(ctypes.byref(shape))
nwg = 2048
raw = ws[off + 4 * 300001: off + 4 * 300001 + 32 * nwg].view(torch.int32).cpu().numpy().reshape(nwg, 8).astype(np.int64)
t0 = raw[:, 0][raw[:, 0] > 0].min()
us = lambda a: ((a - t0) % (1 << 31)) / 100.0
st, fb = raw[:1024], raw[1024:]
pct = lambda a: 'p10 %.1f  median %.1f  p90 %.1f  max %.1f' % (np.percentile(a, 10), np.median(a), np.percentile(a, 90), a.max())
print(f'{fam}: times in us after the first workgroup of the launch started')
print('  strip workgroups      start:', pct(us(st[:, 0])))
print('  strip workgroups        end:', pct(us(st[:, 3])), '  (lifetime: ' + pct(us(st[:, 3]) - us(st[:, 0])) + ')')
print('  fallback workgroups   start:', pct(us(fb[:, 0])))
print('  ... main list done         :', pct(us(fb[:, 1])), '  (its entries per wavefront: max %d; duration: ' % fb[:, 4].max() + pct(us(fb[:, 1]) - us(fb[:, 0])) + ')')
w = fb[fb[:, 2] > 0]
if len(w):
    print('  ... wait for the strips over:', pct(us(w[:, 2])))
    print('  ... late list done (= end)  :', pct(us(w[:, 3])), '  (its entries per wavefront: max %d; duration: ' % w[:, 5].max() + pct(us(w[:, 3]) - us(w[:, 2])) + ')')
print('  last workgroup ends at %.1f us' % max(us(st[:, 3]).max(), us(fb[:, 3]).max()))
