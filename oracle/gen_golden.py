#!/usr/bin/env python3
"""Generate golden input/output vectors by importing the UNMODIFIED reference.

    python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]

Runs only in the build container (the reference never travels to the GPU box).  Third-party
packages the reference imports but this image lacks (pykeops, torchvision, pytorch_lightning,
cv2, numba, omegaconf) are satisfied by the stand-ins in oracle/stubs/ -- see their
docstrings; the reference's own files are imported as they are, read-only, without bytecode.
Only DATA (inputs and the reference's outputs) is written; no reference source is copied.
"""
import argparse
import os
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, 'stubs'))
    sys.path.insert(1, args.ref)
    import torch
    from src.losses import LossFactory          # reference, unmodified
    from src import utils as rutils             # reference, unmodified
    from src.utils import loss as rloss

    torch.set_num_threads(8)
    os.makedirs(args.out, exist_ok=True)

    def base_cfg(**kw):
        cfg = dict(image_shape=(48, 64), num_tref=1, num_bins=5, num_knn=4, smooth_weight=0.003,
                   lut_superpixel_size=4, focus_loss_norm='l1', dist_norm='l2',
                   scale_iwe_by_dt=True, mask_image_border=True, polarity_aware_batching=True,
                   interpolation_scheme='mean', smooth_type='on_flow_to_tref')
        cfg.update(kw)
        return cfg

    def synth_events(g, b, m, shape, nb, num_pos, pad):
        h, w = shape
        ev = torch.zeros(b, m, 6)
        ev[..., 0] = torch.rand(b, m, generator=g) * (h - 1)
        ev[..., 1] = torch.rand(b, m, generator=g) * (w - 1)
        t = torch.rand(b, m, generator=g)
        ev[..., 2] = t
        ev[:, :num_pos, 3] = 1
        ev[..., 4] = torch.clamp(torch.floor(t * nb), 0, nb - 1)
        ev[..., 5] = 1
        if pad:
            ev[:, num_pos - pad:num_pos] = 0
            ev[:, m - pad:] = 0
            ev[1:, num_pos - 2 * pad:num_pos] = 0      # ragged: later samples are shorter
        return ev

    def ref_trajectories(coeff_grid, times, shape, patch, num_basis, basis_type):
        """What TrajectoryNet.calculate_trajectories_at_t does (trajectory_net.py:101-119), using
        the reference's own helpers."""
        mask = rutils.get_optical_flow_tile_mask(shape, patch)
        coeffs, pos, _ = rutils.coeffs_grid_to_list(coeff_grid, mask, num_coeffs=num_basis)
        anchor = torch.full((1,), 0.0, dtype=coeffs.dtype)
        traj = rutils.compute_basis(coeffs, times, num_basis, basis_type) - \
            rutils.compute_basis(coeffs, anchor, num_basis, basis_type)
        traj = traj + pos[None, :, None, :]
        return traj.permute(0, 2, 1, 3).contiguous()

    def run_case(name, cfg, b, m, patch, num_basis, basis_type, coeff_sigma, seed, pad=0,
                 t_ref=0.41, const_flow=None, extra=None, store_idx=True):
        g = torch.Generator().manual_seed(seed)
        shape = cfg['image_shape']
        nb, T = cfg['num_bins'], cfg['num_tref']
        num_pos = m // 2 + 7
        ev = synth_events(g, b, m, shape, nb, num_pos, pad)
        coeff_grid = torch.randn(b, 1, 2 * num_basis, *shape, generator=g) * coeff_sigma
        if const_flow is not None:
            coeff_grid = torch.zeros_like(coeff_grid)
            coeff_grid[:, :, 0] = const_flow[0]
            coeff_grid[:, :, num_basis] = const_flow[1]
        coeff_grid.requires_grad_(True)
        L = LossFactory.get_loss_calculator('FOCUS', cfg)
        times = L.get_reconstruction_times('cpu')
        if T == 1:
            times = times.clone()
            times[0] = t_ref
        traj = ref_trajectories(coeff_grid, times, shape, patch, num_basis, basis_type)
        traj.retain_grad()
        batch = {'events': ev}
        if cfg['polarity_aware_batching']:
            batch['num_pos_events'] = num_pos

        # stage-wise (same calls as FocusLoss.calc, focus.py:82-94)
        flow_lut, flow_next = L.interpolate_flow(traj[:, :T], traj[:, T:])
        flow_lut.retain_grad()
        if flow_next is not None:
            flow_next.retain_grad()
        warped = L.warp_events(ev, flow_lut)
        iwes = L.make_iwes(warped, times[:T], batch.get('num_pos_events', -1))
        focus = rutils.calculate_focus_loss(iwes, loss_type='gradient_magnitude',
                                            norm=cfg['focus_loss_norm'])
        smooth = L.calculate_smooth_loss(flow_lut, flow_next)
        loss = focus + smooth
        loss.backward()

        # the public entry point must agree with the stage-wise run
        loss2, log2, misc2 = L.calc(traj.detach(), times, batch)
        assert torch.allclose(loss2, loss.detach(), rtol=1e-6), (loss2, loss)

        out = dict(
            cfg_keys=np.array(sorted(cfg.keys())),
            cfg_vals=np.array([str(cfg[k]) for k in sorted(cfg.keys())]),
            b=b, m=m, patch=patch, num_basis=num_basis, basis_type=basis_type, num_pos=num_pos,
            events=ev.numpy(), coeff_grid=coeff_grid.detach().numpy(), times=times.numpy(),
            trajectories=traj.detach().numpy(), flow_lut=flow_lut.detach().numpy(),
            warped_yx=warped[..., :2].detach().numpy(),
            iwes=misc2['iwes'].numpy(), focus_loss=focus.item(), smooth_loss=float(smooth),
            loss=loss.item(), grad_trajectories=traj.grad.numpy(),
            grad_flow_lut=flow_lut.grad.numpy(), grad_coeff_grid_abs_sum=coeff_grid.grad.abs().sum().item(),
            # the whole gradient of the coefficient grid (trajectory_net.py:101-119,142-161): it is non-zero at the tile
            # centres only (the mask of get_optical_flow_tile_mask), stored there as [b, 1, 2k, tiles]
            grad_coeff_grid_at_tiles=coeff_grid.grad[..., rutils.get_optical_flow_tile_mask(shape, patch)].numpy(),
            grad_coeff_grid_off_tiles_abs_max=coeff_grid.grad[..., ~rutils.get_optical_flow_tile_mask(shape, patch)].abs().max().item(),
        )
        if flow_next is not None:
            out['flow_next'] = flow_next.detach().numpy()
            out['grad_flow_next'] = flow_next.grad.numpy()
        if store_idx:
            # neighbour index sets via the same lazy-tensor expression as focus.py:129-137
            from pykeops.torch import LazyTensor
            h, w = shape
            sp = cfg['lut_superpixel_size']
            off = float(sp) / 2 - 0.5
            gy, gx = torch.meshgrid(torch.arange(0, h, sp, dtype=torch.float32) + off,
                                    torch.arange(0, w, sp, dtype=torch.float32) + off, indexing='ij')
            gp = torch.stack((gy, gx), -1).reshape(-1, 2)
            x_i = LazyTensor(gp[None].contiguous())
            q_j = LazyTensor(traj.detach()[:, T:][..., None, :].contiguous())
            dist = ((x_i - q_j) ** 2).sum(-1) if cfg['dist_norm'] == 'l2' else (x_i - q_j).abs().sum(-1)
            ind = dist.argKmin(cfg['num_knn'], dim=2)
            out['ind_k_sorted'] = np.sort(ind.numpy(), -1).astype(np.int32)
        if extra:
            out.update(extra(L, ev, iwes.detach(), traj.detach(), times, batch))
        np.savez_compressed(os.path.join(args.out, name + '.npz'), **out)
        print(f'{name}: loss={loss.item():.8f} focus={focus.item():.8f} smooth={float(smooth):.8e}')

    # G1: all flags on, padded + ragged rows, large flows (events leave the image)
    def g1_extra(L, ev, iwes, traj, times, batch):
        # second caller of the API: logging.py:76-79 (raw events, weight=1.0)
        img = L.imager.create_iwe(ev[:1], method='bilinear_vote', sigma=1)
        return {'imager_iwe_raw_events': img.numpy()}
    run_case('g1_allflags', base_cfg(), b=2, m=3000, patch=4, num_basis=1, basis_type='polynomial',
             coeff_sigma=9.0, seed=1, pad=40, extra=g1_extra)

    # G2: BASELINE config 1 shape: 128x128, 10k events, K=32, constant flow, + variance objective
    def g2_extra(L, ev, iwes, traj, times, batch):
        v = rutils.calculate_focus_loss(iwes, loss_type='variance')
        return {'variance_focus_loss': v.item()}
    run_case('g2_config1', base_cfg(image_shape=(128, 128), num_bins=15, num_knn=32), b=1, m=10000,
             patch=4, num_basis=1, basis_type='polynomial', coeff_sigma=0.0, seed=2,
             const_flow=(3.5, -6.25), extra=g2_extra, store_idx=False)

    # G3: B*T == 1 squeeze path, no polarity batching, no dt scaling, no border mask, K=1
    run_case('g3_squeeze_k1', base_cfg(polarity_aware_batching=False, scale_iwe_by_dt=False,
                                       mask_image_border=False, num_knn=1), b=1, m=2000, patch=4,
             num_basis=1, basis_type='polynomial', coeff_sigma=4.0, seed=3)

    # G3b: num_tref = 3 (linspace reference times), single channel
    run_case('g3b_tref3', base_cfg(num_tref=3, polarity_aware_batching=False, scale_iwe_by_dt=False,
                                   num_knn=3), b=2, m=1500, patch=4,
             num_basis=1, basis_type='polynomial', coeff_sigma=3.0, seed=33)

    # G4: inverse-distance weights, L1 distance, smoothness on flow_to_next
    run_case('g4_iwd_l1_next', base_cfg(interpolation_scheme='iwd', dist_norm='l1',
                                        smooth_type='on_flow_to_next', smooth_weight=0.06,
                                        num_knn=6, num_bins=7), b=2, m=2500, patch=4,
             num_basis=1, basis_type='polynomial', coeff_sigma=3.0, seed=4, pad=11)

    # G5a/b: DCT and polynomial-k3 bases, L2 contrast norm, patch 8 with superpixel 4 (n != Q)
    run_case('g5a_dct3_l2', base_cfg(focus_loss_norm='l2', num_knn=5), b=2, m=2000, patch=8,
             num_basis=3, basis_type='dct', coeff_sigma=1.0, seed=5)
    run_case('g5b_poly3', base_cfg(num_knn=8, lut_superpixel_size=8), b=1, m=2000, patch=4,
             num_basis=3, basis_type='polynomial', coeff_sigma=1.0, seed=6)

    # G6: Bezier degree-10 flows at 6 timestamps (bezier.py:92-113 via get_flow_from_reference)
    from src.models.raft_spline.curves import BezierCurves
    g = torch.Generator().manual_seed(7)
    params = torch.randn(2, 20, 6, 8, generator=g) * 2.0
    curve = BezierCurves(params)
    ts = np.array([0.0, 0.13, 0.37, 0.5, 0.81, 1.0], dtype='float64')
    flows = curve.get_flow_from_reference(ts)
    np.savez_compressed(os.path.join(args.out, 'g6_bezier10.npz'), params=params.numpy(),
                        timestamps=ts, flows=flows.numpy(),
                        flow_t1=curve.get_flow_from_reference(1.0).numpy())
    print('g6_bezier10: ok', tuple(flows.shape))

    # G7: contrast/smoothness primitives on a random image (loss.py) incl. B*T=1 3-dim input
    g = torch.Generator().manual_seed(8)
    img = torch.rand(3, 2, 20, 28, generator=g) * 4
    fld = torch.randn(4, 2, 12, 16, generator=g)
    np.savez_compressed(
        os.path.join(args.out, 'g7_primitives.npz'), img=img.numpy(), field=fld.numpy(),
        gm_l1=rloss.calculate_focus_loss(img, 'gradient_magnitude', 'l1').item(),
        gm_l2=rloss.calculate_focus_loss(img, 'gradient_magnitude', 'l2').item(),
        var=rloss.calculate_focus_loss(img, 'variance').item(),
        gm_l1_3dim=rloss.calculate_focus_loss(img[:, 0], 'gradient_magnitude', 'l1').item(),
        smooth=rloss.calculate_smoothness_loss(fld).item())
    print('g7_primitives: ok')


if __name__ == '__main__':
    main()
