"""Timing + fallback statistics of the KNN LUT kernels at the C3 shape (diagnostics)."""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from motionpriorcmax_amd import ops, LossFactory, _lib as C
if os.environ.get('MPC_AB_LIB'):          # A/B timing of two builds on the same box
    C.LIB_PATH = os.path.abspath(os.environ['MPC_AB_LIB'])

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wl = dict(bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else 'C3'])
wl['B'] = B
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
cfg = L._cfg
trajd = traj.to(dev)
shape = ops.make_shape(cfg, B, 0, 0, traj.shape[2])
ws = ops.alloc_workspace(shape, dev)
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lut, nxt, state, _ = ops.knn_lut_fwd(cfg, shape, trajd, ws)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    g = torch.randn_like(lut)
    gn = torch.randn_like(nxt) if nxt is not None else None
    torch.cuda.synchronize(); t2 = time.perf_counter()
    gt = ops.knn_lut_bwd(shape, trajd, g, gn, state, ws)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f'fwd {1e6*(t1-t0):.0f} us  bwd {1e6*(t3-t2):.0f} us')
# fallback statistics of the strip kernel (diagnostics entry point)
off = C.lib().mpc_knn_fail_list_offset(ctypes.byref(shape))
lst = ws[off:off + 4 * (1 + B * cfg.num_bins * shape.hq * shape.wq)].view(torch.int32)
n = int(lst[0].item())
ent = lst[1:1 + n].cpu().numpy().astype('uint32')
why = ent >> 30
qq = ent & 0x3fffffff
import numpy as np
print(f'fallback queries: {n} of {B * cfg.num_bins * shape.hq * shape.wq} ({100.0 * n / (B * cfg.num_bins * shape.hq * shape.wq):.3f} %), by reason {np.bincount(why, minlength=4).tolist()}')
cyq = (qq % (shape.hq * shape.wq)) // shape.wq
cxq = qq % shape.wq
print('rows of reason-2 fails (count):', [(int(i), int(c)) for i, c in enumerate(np.bincount(cyq[why == 2], minlength=shape.hq)) if c][:60])
print('cols of reason-2 fails (count):', [(int(i), int(c)) for i, c in enumerate(np.bincount(cxq[why == 2], minlength=shape.wq)) if c][:40])
btq = qq // (shape.hq * shape.wq)
print('bins of reason-2 fails:', np.bincount(btq[why == 2] % cfg.num_bins, minlength=cfg.num_bins).tolist())
print('rows of reason-1 fails:', np.bincount(cyq[why == 1], minlength=shape.hq).nonzero()[0].tolist()[:40])
print('cols of reason-1 fails:', np.bincount(cxq[why == 1], minlength=shape.wq).nonzero()[0].tolist()[:40])
# emulate some interior reason-2 queries on the host
sel = np.nonzero((why == 2) & (cyq > 10) & (cyq < shape.hq - 10) & (cxq > 10) & (cxq < shape.wq - 10))[0][:4]
tr = traj.numpy()
for j in sel:
    bt_ = int(qq[j]) // (shape.hq * shape.wq); b_, t_ = bt_ // cfg.num_bins, bt_ % cfg.num_bins
    qy_, qx_ = int(cyq[j]), int(cxq[j])
    pts = tr[b_, 1 + t_]
    pcy = np.clip(np.floor((pts[:, 0] + np.float32(0.5)) / 4), 0, shape.hq - 1).astype(int)
    pcx = np.clip(np.floor((pts[:, 1] + np.float32(0.5)) / 4), 0, shape.wq - 1).astype(int)
    m = (abs(pcy - qy_) <= 3) & (pcx >= (qx_ & ~1) - 3) & (pcx <= (qx_ | 1) + 3)
    P = pts[m]
    d = (np.float32(qy_ * 4 + 1.5) - P[:, 0]) ** 2 + (np.float32(qx_ * 4 + 1.5) - P[:, 1]) ** 2
    up = np.float32(13.99) ** 2
    lv = np.clip(np.floor(np.minimum(d * (np.float32(64) / up) - 32, 32) + 0.5), 0, 32).astype(int)
    h = np.bincount(lv, minlength=33)
    print('query', b_, t_, qy_, qx_, 'cands', m.sum(), 'valid', int((lv < 32).sum()), 'levels hist', h.tolist())
