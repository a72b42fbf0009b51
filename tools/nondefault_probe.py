import sys, time, torch
sys.path.insert(0, '/root/repo')
import bench
from motionpriorcmax_amd import LossFactory, ops
dev = torch.device('cuda:0')
wl = bench.WORKLOADS['C3']
for tag, over, trefs in (('l2', {}, (0.41,)), ('iwd', {'interpolation_scheme': 'iwd'}, (0.41,)), ('dist_l1', {'dist_norm': 'l1'}, (0.41,)),
                         ('num_tref_2', {'num_tref': 2, 'scale_iwe_by_dt': False, 'polarity_aware_batching': False}, (0.41, 0.77))):
    ev, npos, tr, tm = bench.synth_inputs(wl, seed=1, trefs=trefs)
    L = LossFactory.get_loss_calculator('FOCUS', dict(bench.loss_config(wl), **over))
    evd, tmd = ev.to(dev), tm.to(dev)
    trd = tr.to(dev).requires_grad_(True)
    b = {'events': evd, 'num_pos_events': npos}
    def st():
        l, _, _ = L.calc(trd, tmd, b); l.backward(); trd.grad = None
    for _ in range(4): st()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): st()
    torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 10
    with ops.KernelTimer() as kt:
        for _ in range(5): st()
    ks = {k: round(v['total_us'] / 5, 1) for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]['total_us'])}
    print(f'{tag:12s} {1e3*t:.4f} ms  ' + ' '.join(f'{k}={v}' for k, v in list(ks.items())[:7]), flush=True)
