"""Does the ORDER OF THE ROWS INSIDE a (time bin, LUT strip) bucket matter to the event kernels?  (round 3)

The bucketed layout (mpc_event_bucket_order / mpc_ingest_scatter_ordered) fixes which rows a bucket holds, not their order
inside it -- any order gives the same loss bit for bit (integer accumulation).  This probe re-sorts the rows inside every
bucket by an image tile of TILE x TILE pixels (done here with torch: an experiment, not a product path) and times the
step and its kernels for each tile size.  Usage: python tools/intra_bucket_order_probe.py [workload] [tile sizes ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from motionpriorcmax_amd import LossFactory, ops  # noqa: E402


def resort(batch, L, tile, mode):
    ev, offs = batch['events'], batch['event_offsets']
    B, M, _ = ev.shape
    out = ev.clone()
    W = L._cfg['image_shape'][1] if isinstance(L._cfg, dict) else 640
    for b in range(B):
        e = ev[b]
        row = torch.arange(M, device=ev.device)
        # bucket id of every row from the offsets table (both polarity blocks): searchsorted over the table's entries
        tab = offs[b].reshape(-1).to(torch.int64)            # [2 * (NK + 1)] non-decreasing inside each block
        nk1 = offs.shape[-1]
        bid = torch.empty(M, dtype=torch.int64, device=ev.device)
        for pol in range(2):
            t = tab[pol * nk1:(pol + 1) * nk1]
            lo, hi = int(t[0]), (batch['num_pos_events'] if pol == 0 else M)
            r = row[lo:hi]
            bid[lo:hi] = torch.searchsorted(t, r, right=True) + pol * (nk1 + 1)      # padding rows: beyond the last entry
        ty = torch.div(e[:, 0], tile, rounding_mode='floor').to(torch.int64)
        tx = torch.div(e[:, 1], tile, rounding_mode='floor').to(torch.int64)
        if mode == 'tile':
            fine = ty * 4096 + tx
        elif mode == 'row':
            fine = torch.div(e[:, 0], 1, rounding_mode='floor').to(torch.int64) * 4096 + tx
        else:
            fine = tx * 4096 + ty
        key = bid * (1 << 26) + fine
        perm = torch.argsort(key, stable=True)
        out[b] = e[perm]
    return {'events': out, 'num_pos_events': batch['num_pos_events'], 'event_offsets': offs}


def measure(L, trajd, times_d, batch, steps=30):
    def step():
        loss, _, _ = L.calc(trajd, times_d, batch)
        loss.backward()
        trajd.grad = None
        return loss
    for _ in range(8):
        last = step()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(steps):
        last = step()
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / steps
    with ops.KernelTimer() as kt:
        for _ in range(5):
            step()
    ks = {k: round(v['avg_us'], 1) for k, v in kt.summary().items()}
    return ms, float(last), ks


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'C3'
    tiles = [int(a) for a in sys.argv[2:]] or [64, 32, 16, 8]
    dev = torch.device('cuda:0')
    wl = bench.WORKLOADS[name]
    ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
    L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    trajd = traj.to(dev).requires_grad_(True)
    times_d = times.to(dev)
    base = L.order_events({'events': ev.to(dev), 'num_pos_events': num_pos})
    keep = ('k_ev_bin', 'k_iwe_accum', 'k_lut_accum', 'k_contrast_march')
    ms, loss0, ks = measure(L, trajd, times_d, base)
    print(f'{name} bucket order as delivered: {ms:.4f} ms/step loss {loss0!r}', {k: ks.get(k) for k in keep}, flush=True)
    for mode in ('tile', 'row'):
        for t in tiles:
            b2 = resort(base, L, t, mode)
            ms, loss, ks = measure(L, trajd, times_d, b2)
            print(f'{name} rows of a bucket by {mode} {t:3d}: {ms:.4f} ms/step loss_equal {loss == loss0}', {k: ks.get(k) for k in keep}, flush=True)


if __name__ == '__main__':
    main()
