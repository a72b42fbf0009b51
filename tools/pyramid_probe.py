"""C3 step with pyramid_levels 1 and 3 (the unpinned IWE-pyramid extension): ms per step and the per-kernel HIP events (us x launches per step)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from motionpriorcmax_amd import LossFactory, ops
dev = torch.device('cuda:0')
wl = bench.WORKLOADS['C3']
ev, npos, tr, tm = bench.synth_inputs(wl, seed=1)
for lv in (1, 3):
    L = LossFactory.get_loss_calculator('FOCUS', dict(bench.loss_config(wl), pyramid_levels=lv, auto_static_shapes=False))
    evd, tmd = ev.to(dev), tm.to(dev)
    trd = tr.to(dev).requires_grad_(True)
    b = {'events': evd, 'num_pos_events': npos}
    def st():
        loss, _, _ = L.calc(trd, tmd, b); loss.backward(); trd.grad = None
    for _ in range(5): st()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): st()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 20 * 1e3
    with ops.KernelTimer() as kt:
        for _ in range(10): st()
    ks = {k: (round(v['total_us'] / 10, 1), v['launches'] // 10) for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]['total_us'])}
    print(lv, round(ms, 4), ' '.join(f'{k}={v[0]}x{v[1]}' for k, v in ks.items()), 'sum', round(sum(v[0] for v in ks.values()), 1))
