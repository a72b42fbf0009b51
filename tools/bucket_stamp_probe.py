"""Phases of a workgroup of k_knn_bucket (diagnostics build -DKNN_BK_STAMP of the in-tree library: thread 0 stamps the phases,
every stamp behind an s_waitcnt(0); they land far inside the fallback list).  On the GPU box:
    MPC_EXTRA_HIPCC_FLAGS=-DKNN_BK_STAMP python -m motionpriorcmax_amd.build && python tools/bucket_stamp_probe.py C3 14
(restore the product build afterwards: python -m motionpriorcmax_amd.build)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from motionpriorcmax_amd import LossFactory, ops, _lib as C  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
wl = dict(bench.WORKLOADS[name])
if len(sys.argv) > 2:
    wl['B'] = int(sys.argv[2])
B = wl['B']
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
cfg = L._cfg
shape = ops.make_shape(cfg, B, 0, 0, traj.shape[2])
ws = ops.alloc_workspace(shape, dev)
trajd = traj.to(dev)
for _ in range(3):
    ops.knn_lut_fwd(cfg, shape, trajd, ws)
torch.cuda.synchronize()
off = C.lib().mpc_knn_fail_list_offset(ctypes.byref(shape))
nb = cfg.num_bins
S = max(1, min(8, 256 // (B * nb)))
nwg = B * nb * S
raw = ws[off + 4 * 100001: off + 4 * 100001 + 32 * nwg].view(torch.int32).cpu().numpy().reshape(nwg, 8).astype(np.int64)
t0 = (raw[:, 0] - raw[:, 0].min()) % (1 << 32) / 100.0          # start of every workgroup after the first one's
st = raw / 100.0
st[:, 0] = 0.0
ph = np.diff(st, axis=1)
names = ['point loads + cells', 'zero + count atomics + barrier', 'scan', 'cell_start written', 'rank atomics + index scatter', 'per-cell index order', 'gather + write']
print(f'{name} B={B}: {nwg} workgroups ({S} per (sample, bin)); lifetime us mean {st[:, 7].mean():.2f} max {st[:, 7].max():.2f}')
for k, nm in enumerate(names):
    print(f'  {nm:34s} mean {ph[:, k].mean():6.2f} us   max {ph[:, k].max():6.2f}')
print(f'  workgroup starts after the first: median {np.median(t0):.2f} us, p90 {np.percentile(t0, 90):.2f}, max {t0.max():.2f}; last end {np.max(t0 + st[:, 7]):.2f} us after the first start')
