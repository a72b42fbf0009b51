#!/usr/bin/env python3
"""One workload of bench.py, eager calc + backward: ms per step and the per-kernel times (HIP events around every launch).
For A/B timing of two builds on one box:  MPC_AB_LIB=build_ab/x.so python tools/step_probe.py C4b6 [--steps 20] [--over key=value ...]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from motionpriorcmax_amd import LossFactory, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('workload')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--over', nargs='*', default=[], help='loss settings to override, e.g. interpolation_scheme=iwd')
    ap.add_argument('--top', type=int, default=8)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    wl = bench.WORKLOADS[args.workload]
    over = {}
    for kv in args.over:
        k, v = kv.split('=')
        over[k] = int(v) if v.lstrip('-').isdigit() else (v == 'True' if v in ('True', 'False') else v)
    ev, npos, tr, tm = bench.synth_inputs(wl, seed=1)
    L = LossFactory.get_loss_calculator('FOCUS', dict(bench.loss_config(wl), auto_static_shapes=False, **over))
    evd, tmd = ev.to(dev), tm.to(dev)
    trd = tr.to(dev).requires_grad_(True)
    b = {'events': evd, 'num_pos_events': npos}

    def st():
        loss, _, _ = L.calc(trd, tmd, b)
        loss.backward()
        trd.grad = None
    for _ in range(5):
        st()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(args.steps):
            st()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / args.steps)
    with ops.KernelTimer() as kt:
        for _ in range(args.steps):
            st()
    ks = {k: round(v['total_us'] / args.steps, 1) for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]['total_us'])}
    tag = os.path.basename(os.environ.get('MPC_AB_LIB', 'product'))
    print(f'{args.workload:6s} {tag:24s} {1e3 * sorted(ts)[1]:.4f} ms  ' + ' '.join(f'{k}={v}' for k, v in list(ks.items())[:args.top]), flush=True)


if __name__ == '__main__':
    main()
