#!/usr/bin/env python3
"""Which queries does the strip kernel hand to the fallback, and why?  python tools/fail_probe.py [family] [B] [loss key=value ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from motionpriorcmax_amd import LossFactory, ops, _lib as C  # noqa: E402
from motionpriorcmax_amd.utils import synth  # noqa: E402

fam = sys.argv[1] if len(sys.argv) > 1 else 'translate40'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda', 0)
wl = bench.WORKLOADS['C3']
if fam == 'white':
    _, _, traj, _ = bench.synth_inputs(dict(wl, B=B), seed=1)
else:
    traj, _ = synth.synth_trajectories(B, 3, wl['nb'], (bench.H, bench.W), bench.PATCH, fam, seed=11)
over = dict(kv.split('=') for kv in sys.argv[3:])          # e.g. dist_norm=l1
L = LossFactory.get_loss_calculator('FOCUS', dict(bench.loss_config(wl), **over))
shape = ops.make_shape(L._cfg, B, 0, 0, traj.shape[2])
ws = ops.alloc_workspace(shape, dev)
ops.knn_lut_fwd(L._cfg, shape, traj.to(dev), ws)
torch.cuda.synchronize()
off = C.lib().mpc_knn_fail_list_offset(ctypes.byref(shape))
n = int(ws[off:off + 4].view(torch.int32).item())
ent = ws[off + 4:off + 4 + 4 * n].view(torch.int32).cpu().numpy().astype(np.uint32)
G = shape.hq * shape.wq
q = ent & (0x00ffffff if B * shape.nb * G < (1 << 24) else 0x3fffffff)
why = ent >> 30
bt = q // G
cell = q % G
cy, cx = cell // shape.wq, cell % shape.wq
print(f'{fam}: {n} of {B * shape.nb * G} queries on the list ({100.0 * n / (B * shape.nb * G):.3f} %)')
print('why (0 few candidates / beyond the far radius, 1 too many slots, 2 staging overflow):', np.bincount(why, minlength=4))
print('by bin:', np.bincount(bt % shape.nb, minlength=shape.nb))
for w in range(3):
    m = why == w
    if m.sum():
        print(f' why {w}: rows min/max {cy[m].min()}..{cy[m].max()}  cols {cx[m].min()}..{cx[m].max()};  row histogram (16-row bands):',
              np.bincount(cy[m] // 16, minlength=8), ' col histogram (16-col bands):', np.bincount(cx[m] // 16, minlength=10))

offs = (ctypes.c_int64 * 6)()
C.lib().mpc_knn_list_offsets(ctypes.byref(shape), offs)
i32 = lambda o, n=1: ws[o:o + 4 * n].view(torch.int32).cpu().numpy()
nstrips = B * shape.nb * ((shape.wq + 1) // 2) * ((shape.hq + 127) // 128)
aw = B * shape.nb * shape.hq * ((shape.wq + 31) // 32)
ag = ws[offs[2]:offs[2] + 4 * aw].view(torch.int32).cpu().numpy().view(np.uint32)
gr = ws[offs[3]:offs[3] + 4 * aw].view(torch.int32).cpu().numpy().view(np.uint32)
pop = lambda a: int(np.unpackbits(a.view(np.uint8)).sum())
far_b = gr & ~ag
print(f"second-launch blocks (2 x 128 queries) listed: {int(i32(offs[1])[0])} of {B * shape.nb * ((shape.wq + 1) // 2) * ((shape.hq + 127) // 128)}"); print(f'strips {nstrips}: searched again in quarters {int(i32(offs[0])[0])}, on the second launch\'s list {int(i32(offs[1])[0])};  '
      f'queries marked for the second launch: far {pop(far_b)} ({100.0 * pop(far_b) / (B * shape.nb * G):.3f} %), unfinished {pop(ag)} '
      f'({100.0 * pop(ag) / (B * shape.nb * G):.3f} %; {pop(ag & gr)} of them for more rings)')
if offs[4] >= 0:
    fl = ws[offs[4]:offs[4] + 4 * B * shape.nb * (G + 1)].view(torch.int32).cpu().numpy().reshape(B * shape.nb, G + 1)
    print(f'far lists: {int(fl[:, 0].sum())} queries ({100.0 * fl[:, 0].sum() / (B * shape.nb * G):.3f} %); work items (tiles) of the far backward: {int(i32(offs[5])[0])}')
