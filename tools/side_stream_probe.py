#!/usr/bin/env python3
"""A/B of the library's side stream (round 6) in ONE process on one box: the same steps with mpc_side_stream_enable(0) and (1).

Per input family: ms per step either way, the gain, and whether loss, trajectory gradient and images are bit for bit the same.

    python tools/side_stream_probe.py [--workload C3] [--steps 20] [--families white,unet,...] [--json out.json]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from motionpriorcmax_amd import LossFactory, ops, _lib as C  # noqa: E402
from motionpriorcmax_amd.utils import synth  # noqa: E402
if os.environ.get('MPC_AB_LIB'):
    C.LIB_PATH = os.path.abspath(os.environ['MPC_AB_LIB'])


def timed(step, steps, blocks=3):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(blocks):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / steps)
    return 1e3 * sorted(ts)[len(ts) // 2]


def run(wl, name, traj, times, ev, num_pos, steps, dev, layout, kernels):
    L = LossFactory.get_loss_calculator('FOCUS', dict(bench.loss_config(wl), auto_static_shapes=False))
    evd, td = ev.to(dev), times.to(dev)
    trajd = traj.to(dev).requires_grad_(True)
    batch = {'events': evd, 'num_pos_events': num_pos}
    if layout == 'bucket':
        batch = L.order_events(batch)
    keep = {}

    def step():
        loss, _, misc = L.calc(trajd, td, batch)
        loss.backward()
        keep['loss'], keep['grad'], keep['iwe'] = loss.detach(), trajd.grad, misc['iwes']
        trajd.grad = None
    res = {'variant': name}
    outs = {}
    for on in (0, 1, 0, 1):
        C.lib().mpc_side_stream_enable(on)
        ms = timed(step, steps)
        key = 'side_on_ms' if on else 'side_off_ms'
        res[key] = round(min(ms, res.get(key, 1e9)), 4)
        torch.cuda.synchronize()
        outs[on] = (keep['loss'].clone(), keep['grad'].clone(), keep['iwe'].clone())
        if kernels and key + '_kernels' not in res:
            with ops.KernelTimer() as kt:
                for _ in range(steps):
                    step()
            res[key + '_kernels'] = {k: round(v['total_us'] / steps, 1) for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]['total_us'])}
    C.lib().mpc_side_stream_enable(1)
    res['gain_us'] = round(1e3 * (res['side_off_ms'] - res['side_on_ms']), 1)
    res['bitwise_equal'] = all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    res['loss'] = float(outs[1][0])
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='C3')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--families', default='white,unet,translate40,diverge-30,zero,ragged')
    ap.add_argument('--layout', default='time', choices=['time', 'bucket'])
    ap.add_argument('--kernels', action='store_true', help='per-kernel times (HIP events around every launch) either way')
    ap.add_argument('--json', default='')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    wl = bench.WORKLOADS[args.workload]
    B = wl['B']
    ev0, np0, traj0, times0 = bench.synth_inputs(wl, seed=1)
    out = []
    for fam in args.families.split(','):
        if fam == 'white':
            r = run(wl, fam, traj0, times0, ev0, np0, args.steps, dev, args.layout, args.kernels)
        elif fam == 'ragged':
            ev, npos = synth.synth_events_ragged(B, wl['M'], (bench.H, bench.W), wl['nb'], seed=3)
            r = run(wl, fam, traj0, times0, ev, npos, args.steps, dev, args.layout, args.kernels)
        else:
            k = wl['k'] if wl['k'] <= 5 else 3
            traj, times = synth.synth_trajectories(B, k, wl['nb'], (bench.H, bench.W), bench.PATCH, fam, seed=11)
            r = run(wl, fam, traj, times, ev0, np0, args.steps, dev, args.layout, args.kernels)
        out.append(r)
        print(f"{fam:12s} off {r['side_off_ms']:.4f}  on {r['side_on_ms']:.4f} ms  gain {r['gain_us']:6.1f} us  bitwise {r['bitwise_equal']}", flush=True)
        if args.kernels:
            for key in ('side_off_ms_kernels', 'side_on_ms_kernels'):
                print('   ', key, ' '.join(f'{k}={v}' for k, v in list(r[key].items())[:13]), flush=True)
    if args.json:
        json.dump(out, open(args.json, 'w'), indent=1)
    assert all(r['bitwise_equal'] for r in out), 'side stream changed a result'


if __name__ == '__main__':
    main()
