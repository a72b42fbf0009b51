#!/usr/bin/env python3
"""C3-shaped steps on the inputs training produces instead of white-noise coefficients: smooth flow fields (translation,
divergence, rotation, shear, UNet-like mixtures), zero flow (exact lattice), ragged batches (30-60 % padding rows).  Per
variant: step time, share of the queries the KNN fast path handed to the fallback, and the per-kernel times.

    python tools/realistic_probe.py [--workload C3] [--steps 10] [--families a,b,...] [--json out.json]
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from motionpriorcmax_amd import LossFactory, ops, utils, _lib as C  # noqa: E402
from motionpriorcmax_amd.utils import synth  # noqa: E402
if os.environ.get('MPC_AB_LIB'):          # A/B timing of two builds on the same box
    C.LIB_PATH = os.path.abspath(os.environ['MPC_AB_LIB'])


def trajectories(wl, family, seed, B):
    k = wl['k'] if wl['k'] <= 5 else 3
    return synth.synth_trajectories(B, k, wl['nb'], (bench.H, bench.W), bench.PATCH, family, seed=seed)


def fail_fraction(L, shape, traj, dev):
    ws = ops.alloc_workspace(shape, dev)
    ops.knn_lut_fwd(L._cfg, shape, traj, ws)
    torch.cuda.synchronize()
    off = C.lib().mpc_knn_fail_list_offset(ctypes.byref(shape))
    n = int(ws[off:off + 4].view(torch.int32).item())
    return n / float(shape.B * shape.nb * shape.hq * shape.wq)


def run_variant(wl, name, traj, times, ev, num_pos, steps, dev, layout):
    L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    evd, td = ev.to(dev), times.to(dev)
    trajd = traj.to(dev).requires_grad_(True)
    batch = {'events': evd, 'num_pos_events': num_pos}
    if layout == 'bucket':
        batch = L.order_events(batch)

    def step():
        loss, _, _ = L.calc(trajd, td, batch)
        loss.backward()
        trajd.grad = None
        return loss
    for _ in range(6):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps)
    with ops.KernelTimer() as kt:
        for _ in range(steps):
            step()
    ks = {k: round(v['total_us'] / steps, 1) for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]['total_us'])}
    shape = ops.make_shape(L._cfg, traj.shape[0], 0, 0, traj.shape[2])
    ff = fail_fraction(L, shape, trajd.detach(), dev)
    return {'variant': name, 'ms_per_step': round(1e3 * sorted(ts)[1], 4), 'loss': float(last.item()),
            'valid_events': float(ev[..., 5].sum()), 'knn_fail_frac': ff, 'kernels_us_per_step': ks}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='C3')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--families', default=','.join(('white',) + synth.FLOW_FAMILIES + ('ragged',)))
    ap.add_argument('--layout', default='time', choices=['time', 'bucket'])
    ap.add_argument('--json', default='')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    wl = bench.WORKLOADS[args.workload]
    B = wl['B']
    ev0, np0, traj0, times0 = bench.synth_inputs(wl, seed=1)
    out = []
    for fam in args.families.split(','):
        if fam == 'white':
            r = run_variant(wl, fam, traj0, times0, ev0, np0, args.steps, dev, args.layout)
        elif fam == 'ragged':
            ev, npos = synth.synth_events_ragged(B, wl['M'], (bench.H, bench.W), wl['nb'], seed=3)
            r = run_variant(wl, fam, traj0, times0, ev, npos, args.steps, dev, args.layout)
        else:
            traj, times = trajectories(wl, fam, 11, B)
            r = run_variant(wl, fam, traj, times, ev0, np0, args.steps, dev, args.layout)
        out.append(r)
        top = list(r['kernels_us_per_step'].items())[:6]
        print(f"{fam:12s} {r['ms_per_step']:.4f} ms  fail {100 * r['knn_fail_frac']:.3f} %  " +
              ' '.join(f'{k}={v}' for k, v in top), flush=True)
    base = out[0]['ms_per_step']
    for r in out:
        r['vs_first'] = round(r['ms_per_step'] / base, 3)
    if args.json:
        json.dump(out, open(args.json, 'w'), indent=1)
    print(json.dumps({r['variant']: [r['ms_per_step'], r['vs_first'], round(r['knn_fail_frac'], 5)] for r in out}))


if __name__ == '__main__':
    main()
