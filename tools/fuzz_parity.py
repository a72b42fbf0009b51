"""Randomised differential test: the HIP path against the CPU oracle over random small configurations
(image sizes, superpixel, patch, K, bins, flags, event counts, flow magnitudes).  Diagnostics; run on a GPU box:

    python tools/fuzz_parity.py [n_cases] [seed]"""
import os
import sys
import random

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from motionpriorcmax_amd import LossFactory
from oracle import focus_oracle as O


def rel_l2(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device('cuda:0')
    bad = 0
    for case in range(n_cases):
        sp = rng.choice([2, 3, 4, 4, 8])
        patch = rng.choice([2, 3, 4, 4, 8])
        H = rng.randrange(24, 140)
        W = rng.randrange(24, 180)
        B = rng.choice([1, 1, 2, 3])
        nb = rng.choice([1, 3, 5, 15])
        M = rng.choice([0, 1, 50, 2000, 20000, 20000])
        sigma = rng.choice([0.0, 0.5, 3.0, 12.0])
        T = rng.choice([1, 1, 1, 2])
        cfg = dict(image_shape=(H, W), num_tref=T, num_bins=nb, num_knn=1, smooth_weight=rng.choice([0.0, 0.003, 0.06]),
                   lut_superpixel_size=sp, focus_loss_norm=rng.choice(['l1', 'l2']), dist_norm=rng.choice(['l1', 'l2']),
                   scale_iwe_by_dt=(T == 1 and rng.random() < 0.7), mask_image_border=rng.random() < 0.7,
                   polarity_aware_batching=(T == 1 and rng.random() < 0.7),
                   interpolation_scheme=rng.choice(['mean', 'iwd']),
                   smooth_type='on_flow_to_tref')
        if T == 1 and rng.random() < 0.3:
            cfg['smooth_type'] = 'on_flow_to_next'
        if nb == 1:
            cfg['smooth_type'] = 'on_flow_to_tref'
        if rng.random() < 0.15:
            cfg['loss_type'] = 'variance'
        mask = O.tile_mask((H, W), patch)
        n = int(mask.sum())
        cfg['num_knn'] = K = max(1, min(n, rng.choice([1, 4, 8, 32, 48])))
        seed = rng.randrange(1 << 30)
        g = torch.Generator().manual_seed(seed)
        np_choice = rng.choice([None, None, 0, M])          # polarity blocks: balanced, all negative, all positive
        ev, num_pos = O.synth_events(B, M, (H, W), nb, seed=seed, pad_frac=rng.choice([0.0, 0.05]), num_pos=np_choice)
        dist_kind = rng.choice(['uniform', 'uniform', 'band', 'spot'])
        if dist_kind != 'uniform' and M > 0:
            # concentrated events: one image strip / LUT strip receives (almost) everything -> one bucket holds (almost) all records
            y0c, x0c = rng.random() * (H - 2), rng.random() * (W - 2)
            rows = ev[..., 5] > 0
            ev[..., 0] = torch.where(rows, y0c + torch.rand(ev.shape[:2], generator=g) * 1.5, ev[..., 0])
            if dist_kind == 'spot':
                ev[..., 1] = torch.where(rows, x0c + torch.rand(ev.shape[:2], generator=g) * 1.5, ev[..., 1])
        coeff = torch.randn(B, 1, 2, H, W, generator=g) * sigma
        t_ref = torch.tensor([rng.random()]) if T == 1 else torch.linspace(0, 1, T)
        times = torch.cat((t_ref, O.bin_mid_times(nb)))
        traj = O.trajectories_at(coeff, times, mask, 1, 'polynomial')
        tag = f'case {case}: {dist_kind} {H}x{W} sp{sp} patch{patch} B{B} nb{nb} M{M} K{K} T{T} sigma{sigma} ' + \
              f"Mp{num_pos} {cfg.get('loss_type', 'gradmag')} " + ' '.join(f'{k}={cfg[k]}' for k in ('focus_loss_norm', 'dist_norm', 'interpolation_scheme', 'smooth_type',
                                                 'scale_iwe_by_dt', 'mask_image_border', 'polarity_aware_batching'))
        if os.environ.get('FUZZ_VERBOSE'):
            print(tag, flush=True)
        if case < int(os.environ.get('FUZZ_FROM', 0)) or case > int(os.environ.get('FUZZ_TO', 1 << 30)):
            continue
        if os.environ.get('FUZZ_LIST') and str(case) not in os.environ['FUZZ_LIST'].split(','):
            continue
        try:
            Lo = O.FocusLossOracle(**cfg)
            tr_o = traj.clone().requires_grad_(True)
            batch = {'events': ev, 'num_pos_events': num_pos}
            lo, _, misco = Lo.calc(tr_o, times, batch)
            lo.backward()
            L = LossFactory.get_loss_calculator('FOCUS', cfg)
            tr_g = traj.to(dev).requires_grad_(True)
            lg, _, miscg = L.calc(tr_g, times.to(dev), {'events': ev.to(dev), 'num_pos_events': num_pos})
            lg.backward()
            lo_v, lg_v = float(lo.detach()), float(lg.detach())
            io = misco['iwes'].numpy()
            ig = miscg['iwes'].cpu().numpy()
            ok = True
            if np.isfinite(lo_v):
                ok &= abs(lg_v - lo_v) <= 2e-5 * abs(lo_v)
                ok &= bool(np.abs(ig - io).max() <= 1e-5 * max(1.0, np.abs(io).max()))
                # sign flips of near-zero Sobel responses are legitimate for 'l1' 
                # (with a single event the whole 'l1' gradient hangs on the signs of a few responses that are zero up
                # to rounding: not compared)
                if not (cfg['focus_loss_norm'] == 'l1' and M <= 1):
                    ok &= rel_l2(tr_g.grad.cpu().numpy(), tr_o.grad.numpy()) < (3e-2 if cfg['focus_loss_norm'] == 'l1' else 1e-3)
            else:
                ok &= not np.isfinite(lg_v)
            # bucket-ordered events (mpc_event_bucket_order): the same loss, IWE and gradient bit for bit, with and
            # without the offsets table (num_tref == 1 and the LDS-tiled event path only)
            if T == 1:
                ob = L.order_events({'events': ev.to(dev), 'num_pos_events': num_pos})
                for bt_ in (ob, {'events': ob['events'], 'num_pos_events': num_pos}):
                    tr_2 = traj.to(dev).requires_grad_(True)
                    l2_, _, m2_ = L.calc(tr_2, times.to(dev), bt_)
                    l2_.backward()
                    same = torch.equal(l2_.detach(), lg.detach()) or (not np.isfinite(lg_v) and not np.isfinite(float(l2_.detach())))
                    same &= torch.equal(m2_['iwes'], miscg['iwes']) and (torch.equal(tr_2.grad, tr_g.grad) or not np.isfinite(lg_v))
                    if not same:
                        ok = False
                        print('ORDERED-EVENTS MISMATCH', tag, float(l2_.detach()), lg_v)
            if not ok:
                bad += 1
                print('MISMATCH', tag, 'loss', lo_v, lg_v, 'iwe', float(np.abs(ig - io).max()),
                      'grad rel', rel_l2(tr_g.grad.cpu().numpy(), tr_o.grad.numpy()))
        except Exception as e:      # noqa: BLE001 -- report and go on
            bad += 1
            print('ERROR', tag, repr(e)[:300])
    print(f'{n_cases} cases, {bad} bad')
    # (the bounds-checked debug build, tools/bounds_run.sh: out-of-range indices its accessors recorded; -1 = product build)
    from motionpriorcmax_amd import _lib as _C
    _n = _C.lib().mpc_bounds_check()
    print('mpc_bounds_check:', _n, _C.lib().mpc_last_error_string().decode() if _n > 0 else '')


if __name__ == '__main__':
    main()
