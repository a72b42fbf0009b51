#!/usr/bin/env python3
"""Phases of a work item of the far pass of k_knn_tail (strip workgroups) (diagnostics build -DKS_STAMP2 of the in-tree library: thread 0
stamps behind the barriers; the last item of every workgroup stays).  On the GPU box:
    MPC_EXTRA_HIPCC_FLAGS=-DKS_STAMP2 python motionpriorcmax_amd/build.py && python tools/more_stamp_probe.py translate40
(restore the product build afterwards: python motionpriorcmax_amd/build.py)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from motionpriorcmax_amd import LossFactory, ops, _lib as C  # noqa: E402
from motionpriorcmax_amd.utils import synth  # noqa: E402
if os.environ.get('MPC_AB_LIB'):          # a diagnostics build beside the product library
    C.LIB_PATH = os.path.abspath(os.environ['MPC_AB_LIB'])

fam = sys.argv[1] if len(sys.argv) > 1 else 'translate40'
nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 768
wl = bench.WORKLOADS['C3']
B = wl['B']
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
raw = ws[off + 4 * 200001: off + 4 * 200001 + 48 * nblk].view(torch.int32).cpu().numpy().reshape(nblk, 12).astype(np.int64)
raw = raw[raw[:, 7] > 0]
ph = np.diff(np.concatenate([np.zeros((len(raw), 1)), raw[:, 1:8]], axis=1), axis=1) / 100.0
names = ['marks + radius (table bisection)', 'compaction + chord pushes', 'row table', 'staging', 'search (passes, bisection, ranking)',
         'outputs + list pushes', 'far list + tile marks + flush']
print(f'{fam}: {len(raw)} workgroups with a far-pass item; last item of each: lifetime us mean {raw[:, 7].mean() / 100:.2f} max {raw[:, 7].max() / 100:.2f}; '
      f'staged slots mean {raw[:, 8].mean():.0f} max {raw[:, 8].max()}; marked queries mean {raw[:, 9].mean():.1f} max {raw[:, 9].max()}')
for k, nme in enumerate(names):
    print(f'  {nme:36s} mean {ph[:, k].mean():6.2f} us   p90 {np.percentile(ph[:, k], 90):6.2f}   max {ph[:, k].max():6.2f}')
