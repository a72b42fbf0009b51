"""Per-workgroup durations of k_knn_bwd_tile (diagnostics build -DKNN_BW_STAMP: the stamps land in the fallback list)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from motionpriorcmax_amd import ops, LossFactory, _lib as C
C.LIB_PATH = os.path.abspath(os.environ['MPC_AB_LIB'])
B = int(sys.argv[1]); name = sys.argv[2]
wl = dict(bench.WORKLOADS[name]); wl['B'] = B
ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
dev = torch.device('cuda:0')
L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
cfg = L._cfg
trajd = traj.to(dev)
shape = ops.make_shape(cfg, B, 0, 0, traj.shape[2])
ws = ops.alloc_workspace(shape, dev)
for it in range(3):
    lut, nxt, state, _ = ops.knn_lut_fwd(cfg, shape, trajd, ws)
    g = torch.randn_like(lut); gn = torch.randn_like(nxt) if nxt is not None else None
    gt = ops.knn_lut_bwd(shape, trajd, g, gn, state, ws)
torch.cuda.synchronize()
off = C.lib().mpc_knn_fail_list_offset(ctypes.byref(shape))
gx, gy = -(-shape.wq // 16), -(-shape.hq // 16)
nblk = gx * gy * B * cfg.num_bins
st = ws[off:off + 16 * nblk].view(torch.int32).cpu().numpy().reshape(nblk, 4)
dur, pre, fl, tot = st[:, 0] / 100.0, st[:, 1] / 100.0, st[:, 2], st[:, 3]
bxy = np.arange(nblk) % (gx * gy); by, bx = bxy // gx, bxy % gx
border = (by == 0) | (by == gy - 1) | (bx == 0) | (bx == gx - 1)
print(f'{nblk} workgroups; duration us: mean {dur.mean():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}')
print(f'  inner  tiles ({(~border).sum()}): mean {dur[~border].mean():.1f} max {dur[~border].max():.1f}  pre-phase {pre[~border].mean():.1f}  RQ mean {(fl[~border] & 255).mean():.2f}')
print(f'  border tiles ({border.sum()}): mean {dur[border].mean():.1f} max {dur[border].max():.1f}  pre-phase {pre[border].mean():.1f}  RQ mean {(fl[border] & 255).mean():.2f}')
print(f'  global path {((fl & 256) != 0).sum()}, tie tiles {((fl & 512) != 0).sum()}, tiles with an exact-loop wavefront {((fl & 1024) != 0).sum()}; points per tile mean {tot.mean():.0f} max {tot.max()}')
for name_, m in (('corner', ((by == 0) | (by == gy - 1)) & ((bx == 0) | (bx == gx - 1))), ('top/bottom', ((by == 0) | (by == gy - 1)) & ~((bx == 0) | (bx == gx - 1))), ('left/right', ~((by == 0) | (by == gy - 1)) & ((bx == 0) | (bx == gx - 1)))):
    print(f'  {name_}: n {m.sum()} mean {dur[m].mean():.1f} max {dur[m].max():.1f} RQ {(fl[m] & 255).mean():.2f} slow {((fl[m] & 1024) != 0).mean():.2f}')
full = (by < gy - 1) | (shape.hq % 16 == 0)
for name_, m in (('full tiles, <= 256 points', full & (tot <= 256)), ('full tiles, > 256 points (second round of points)', full & (tot > 256))):
    if m.sum():
        print(f'  {name_}: n {m.sum()} mean {dur[m].mean():.1f} us  inner only {dur[m & ~border].mean() if (m & ~border).sum() else float("nan"):.1f}')
