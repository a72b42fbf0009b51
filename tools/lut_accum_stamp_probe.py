"""Where does a workgroup of k_lut_accum<ordered> spend its life?  Diagnostics build (-DEV_LA_STAMP: wavefront 0 stamps
its phases with the 100 MHz wall clock, every stamp behind an s_waitcnt(0), and leaves them in the first cells of its
strip of the output).  Build + run on the GPU box:
    MPC_EXTRA_HIPCC_FLAGS=-DEV_LA_STAMP python -m motionpriorcmax_amd.build && python tools/lut_accum_stamp_probe.py C3
(restore the product build afterwards: python -m motionpriorcmax_amd.build)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from motionpriorcmax_amd import LossFactory, ops, _lib as C  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
    dev = torch.device('cuda:0')
    wl = bench.WORKLOADS[name]
    ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
    L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    cfg = L._cfg
    batch = L.order_events({'events': ev.to(dev), 'num_pos_events': num_pos})
    evd, offs = batch['events'], batch['event_offsets']
    B, M = evd.shape[0], evd.shape[1]
    shape = ops.make_shape(cfg, B, M, num_pos, traj.shape[2])
    ws = ops.alloc_workspace(shape, dev)
    lut, _, _, _ = ops.knn_lut_fwd(cfg, shape, traj.to(dev), ws)
    t_ref = times.to(dev)[:1]
    P = 2
    H, W = bench.H, bench.W
    gimg = torch.randn(B, 1, P, H, W, device=dev)
    scal = torch.ones(C.SCAL_COUNT, device=dev)
    g_lut = torch.empty_like(lut)
    add = torch.randn_like(lut)
    for _ in range(3):
        ops.event_splat_bwd(shape, evd, lut, t_ref, gimg, scal, None, g_lut, add, ws, offs)
    torch.cuda.synchronize()
    nb, hq, wq = lut.shape[1], lut.shape[2], lut.shape[3]
    ncs = int(C.lib().mpc_event_lut_strips(__import__('ctypes').byref(shape)))
    csr = -(-hq // ncs)
    raw = g_lut.view(torch.int32).reshape(B * nb, hq, wq * 2).cpu().numpy()
    st = np.stack([raw[:, c * csr, :9] for c in range(ncs)], 1).reshape(-1, 9).astype(np.int64)      # [workgroups, 9]
    d = st[:, :7] / 100.0
    n = st[:, 7]
    ph = np.diff(d, axis=1)
    names = ['zero + table strip -> LDS + barrier', 'offsets entries', 'event rows + warp', 'adjoint-image taps', 'LDS atomics + barrier', 'scale + add term + store']
    print(f'{name}: {len(d)} workgroups, rows per workgroup mean {n.mean():.0f} max {n.max()}; lifetime us mean {d[:, 6].mean():.2f} p90 {np.percentile(d[:, 6], 90):.2f} max {d[:, 6].max():.2f}')
    for k, nm in enumerate(names):
        print(f'  {nm:40s} mean {ph[:, k].mean():6.2f} us   p90 {np.percentile(ph[:, k], 90):6.2f}')
    t0 = st[:, 8].astype(np.int64)
    t0 = (t0 - t0.min()) % (1 << 32)
    end = t0 / 100.0 + d[:, 6]
    print(f'  first start -> last end: {end.max():.1f} us; sum of lifetimes / (2 per CU x 256): {d[:, 6].sum() / 512:.1f} us')
    order = np.argsort(t0)
    ts = t0 / 100.0
    print('  workgroups alive at t =', {t: int(((ts <= t) & (end > t)).sum()) for t in (0.5, 1, 2, 4, 6, 8, 12, 20, 30, 40, 50)})
    print('  started within the first 0.5 us:', int((ts < 0.5).sum()))
    print('  start times (us) of every 128th workgroup:', np.round(t0[order][::128] / 100.0, 1))


if __name__ == '__main__':
    main()
