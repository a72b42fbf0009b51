#!/usr/bin/env python3
"""Per-kernel times of the next-row operators (voxel grid, event ingest) at the C3 batch shape:
    python tools/nextrow_probe.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from motionpriorcmax_amd import ops  # noqa: E402
from motionpriorcmax_amd.utils import voxel_grids, ingest_events  # noqa: E402
from oracle import voxel_oracle as VO, ingest_oracle as IO  # noqa: E402

dev = torch.device('cuda:0')
wl = bench.WORKLOADS['C3']
H, W = bench.H, bench.W


def timed(fn, reps=11):
    for _ in range(4):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    with ops.KernelTimer() as kt:
        for _ in range(reps):
            fn()
    ks = {k: round(v['total_us'] / reps, 1) for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]['total_us'])}
    return sorted(ts)[len(ts) // 2], ks


Bv, Nv, vshape = wl['B'], wl['M'], (wl['nb'], H, W)
xs = [torch.stack(VO.synth_raw_events(Nv, vshape, seed=900 + b), -1) for b in range(Bv)]
evv = torch.stack(xs).to(dev)
cntv = torch.full((Bv,), Nv, dtype=torch.int32, device=dev)
for norm in ('mean_std', 'max', None):
    t, ks = timed(lambda: voxel_grids(evv, cntv, vshape, norm))
    print(f'voxel {str(norm):9s} {1e3 * t:.4f} ms  ' + ' '.join(f'{k}={v}' for k, v in ks.items()))
t, ks = timed(lambda: voxel_grids(evv, cntv, vshape, 'mean_std', 0.02))
print(f'voxel q=0.02   {1e3 * t:.4f} ms  ' + ' '.join(f'{k}={v}' for k, v in ks.items()))

raws = [IO.synth_raw(Nv, H, W, seed=950 + b) for b in range(Bv)]
tx = [torch.from_numpy(np.stack([r[k] for r in raws])).to(dev) for k in range(4)]
cnti = torch.full((Bv,), Nv, dtype=torch.int32)
t, ks = timed(lambda: ingest_events(tx[0], tx[1], tx[2], tx[3], cnti, (H, W), wl['nb']))
print(f'ingest         {1e3 * t:.4f} ms  ' + ' '.join(f'{k}={v}' for k, v in ks.items()))
from motionpriorcmax_amd import LossFactory  # noqa: E402
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
t, ks = timed(lambda: ingest_events(tx[0], tx[1], tx[2], tx[3], cnti, (H, W), wl['nb'], order_for=L))
print(f'ingest ordered {1e3 * t:.4f} ms  ' + ' '.join(f'{k}={v}' for k, v in ks.items()))
