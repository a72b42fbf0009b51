"""Randomised differential test of the KNN LUT (forward and backward) against a brute-force K-nearest search in
torch on the device, at sizes up to the full DSEC grid (image, superpixel, patch, K, bins, flow roughness, distance
norm, weighting, flow-to-next).  Diagnostics; run on a GPU box:

    python tools/fuzz_knn.py [n_cases] [seed]"""
import os
import sys
import random

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from motionpriorcmax_amd import LossFactory, ops
from oracle import focus_oracle as O


def brute(traj, shape, sp, K, dist, scheme, want_next):
    """traj [B, 1+nb, n, 2] on the device -> (lut [B,nb,hq,wq,1,2], nxt or None), differentiable."""
    grid, hq, wq = O.lut_grid_points(shape, sp)
    q = grid.to(traj.device)
    B, nb = traj.shape[0], traj.shape[1] - 1
    luts, nxts = [], []
    for b in range(B):
        lb, nbk = [], []
        for t in range(nb):
            pts = traj[b, 1 + t]
            diff = q[:, None, :] - pts.detach()[None, :, :]
            d = diff.abs().sum(-1) if dist == 'l1' else (diff ** 2).sum(-1)
            idx = torch.sort(d, dim=1, stable=True).indices[:, :K]
            f = (traj[b, 0] - pts)[idx]                                  # [Q, K, 2]
            if scheme == 'iwd':
                w = 1.0 / (torch.gather(d, 1, idx) + 1e-9)
                w = (w / w.sum(1, keepdim=True)).detach()
                lb.append((f * w[..., None]).sum(1))
            else:
                lb.append(f.mean(1))
            if want_next and t < nb - 1:
                nbk.append((traj[b, 2 + t] - pts)[idx].mean(1))
        luts.append(torch.stack(lb))
        if want_next and nb > 1:
            nxts.append(torch.stack(nbk))
    lut = torch.stack(luts).reshape(B, nb, hq, wq, 1, 2)
    nxt = torch.stack(nxts).reshape(B, nb - 1, hq, wq, 1, 2) if (want_next and nb > 1) else None
    return lut, nxt


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    dev = torch.device('cuda:0')
    bad = 0
    for case in range(n_cases):
        big = rng.random() < 0.25
        H, W = (rng.choice([(480, 640), (260, 346), (360, 480)]) if big else (rng.randrange(16, 200), rng.randrange(16, 260)))
        sp = rng.choice([2, 3, 4, 4, 4, 8]) if not big else 4
        patch = rng.choice([2, 3, 4, 4, 4, 8]) if not big else rng.choice([4, 4, 8])
        nb = rng.choice([1, 2, 5])
        B = rng.choice([1, 2])
        sigma = rng.choice([0.0, 0.3, 1.0, 3.0, 8.0, 30.0])
        dist, scheme = rng.choice(['l1', 'l2']), rng.choice(['mean', 'mean', 'iwd'])
        want_next = nb > 1 and rng.random() < 0.3
        mask = O.tile_mask((H, W), patch)
        n = int(mask.sum())
        if n >= 65536 or n < 1:
            continue
        K = max(1, min(n, rng.choice([1, 2, 8, 32, 32, 64])))
        seed = rng.randrange(1 << 30)
        g = torch.Generator().manual_seed(seed)
        k_basis = rng.choice([1, 3])
        coeff = torch.randn(B, 1, 2 * k_basis, H, W, generator=g) * sigma
        if rng.random() < 0.2:                       # smooth field: tiles move coherently (converging / diverging regions)
            coeff = torch.nn.functional.avg_pool2d(coeff.reshape(B, 2 * k_basis, H, W), 31, 1, 15).reshape(B, 1, 2 * k_basis, H, W) * 20
        times = torch.cat((torch.tensor([rng.random()]), O.bin_mid_times(nb)))
        traj = O.trajectories_at(coeff, times, mask, k_basis, 'polynomial')
        tag = f'case {case}: {H}x{W} sp{sp} patch{patch} n{n} B{B} nb{nb} K{K} sigma{sigma} k{k_basis} {dist} {scheme} next={want_next}'
        if os.environ.get('FUZZ_VERBOSE'):
            print(tag, flush=True)
        cfg = dict(image_shape=(H, W), num_tref=1, num_bins=nb, num_knn=K, smooth_weight=0.01, lut_superpixel_size=sp,
                   focus_loss_norm='l1', dist_norm=dist, scale_iwe_by_dt=True, mask_image_border=True,
                   polarity_aware_batching=True, interpolation_scheme=scheme,
                   smooth_type='on_flow_to_next' if want_next else 'on_flow_to_tref')
        L = LossFactory.get_loss_calculator('FOCUS', cfg)
        t1 = traj.to(dev).requires_grad_(True)
        lut, nxt = ops.KnnLutFn.apply(t1, L._cfg)
        t2 = traj.to(dev).requires_grad_(True)
        rl, rn = brute(t2, (H, W), sp, K, dist, scheme, want_next)
        gl = torch.randn(lut.shape, generator=g).to(dev)
        obj1, obj2 = (lut * gl).sum(), (rl * gl).sum()
        if want_next and nxt is not None:
            gn = torch.randn(nxt.shape, generator=g).to(dev)
            obj1, obj2 = obj1 + (nxt * gn).sum(), obj2 + (rn * gn).sum()
        obj1.backward(); obj2.backward()
        scale = max(1.0, float(rl.abs().max()))
        e_lut = float((lut - rl).abs().max()) / scale
        e_nxt = float((nxt - rn).abs().max()) / max(1.0, float(rn.abs().max())) if (want_next and nxt is not None) else 0.0
        e_grad = float((t1.grad - t2.grad).norm() / t2.grad.norm().clamp_min(1e-30))
        if not (e_lut < 2e-5 and e_nxt < 2e-5 and e_grad < 2e-5):
            bad += 1
            print('MISMATCH', tag, 'lut', e_lut, 'next', e_nxt, 'grad', e_grad, flush=True)
    print(f'{n_cases} cases, {bad} bad')
    # (the bounds-checked debug build, tools/bounds_run.sh: out-of-range indices its accessors recorded; -1 = product build)
    from motionpriorcmax_amd import _lib as _C
    _n = _C.lib().mpc_bounds_check()
    print('mpc_bounds_check:', _n, _C.lib().mpc_last_error_string().decode() if _n > 0 else '')


if __name__ == '__main__':
    main()
