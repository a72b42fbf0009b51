#!/usr/bin/env python3
"""Step time and per-kernel times of the UNPINNED per-event basis warp (FocusLoss.calc_per_event_basis), fused kernels and the
plain-torch form around the vote / objective kernels:  python tools/per_event_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from motionpriorcmax_amd import LossFactory, ops  # noqa: E402

dev = torch.device('cuda:0')
for name in ('C3', 'C2'):
    wl = bench.WORKLOADS[name]
    kb = wl['k']
    ev, npos, _, _ = bench.synth_inputs(wl, seed=1)
    g = torch.Generator().manual_seed(8)
    cg = (torch.randn(wl['B'], 1, 2 * kb, bench.H, bench.W, generator=g)).to(dev).requires_grad_(True)
    L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    b = {'events': ev.to(dev), 'num_pos_events': npos}
    ob = L.order_events(b)
    for fused in ('ordered', True, False):
        bb = ob if fused == 'ordered' else b

        def st():
            l, _, _ = L.calc_per_event_basis(cg, 0.41, bb, kb, fused=bool(fused))
            l.backward()
            cg.grad = None
        for _ in range(4):
            st()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            st()
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 10
        with ops.KernelTimer() as kt:
            for _ in range(5):
                st()
        ks = {k: round(v['total_us'] / 5, 1) for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]['total_us'])}
        print(f'{name} per-event basis fused={fused} {1e3 * t:.4f} ms  ' + ' '.join(f'{k}={v}' for k, v in list(ks.items())[:9]), flush=True)
