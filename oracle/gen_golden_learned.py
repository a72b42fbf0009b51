#!/usr/bin/env python3
"""Golden vector for the 'learned' motion basis (reference src/modules/trajectory_net.py:35-47,79-80 == src/utils/basis.py:26-27):
the basis is an MLP 1 -> 64 -> 64 -> 64 -> k of the time.  Imports the UNMODIFIED reference (see oracle/gen_golden.py for the
stand-ins), builds the network exactly as TrajectoryNet.__init__ does (nn.Sequential + utils.initialize_weights), runs the
TrajectoryNet.step harness (coefficient grid -> coeffs_grid_to_list -> compute_basis('learned') -> calc -> backward) and
stores inputs, MLP weights, trajectories, loss and the gradients w.r.t. the grid AND the MLP weights.  Data only.

    python oracle/gen_golden_learned.py [--ref /root/reference] [--out tests/golden]
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
    from torch import nn
    from src.losses import LossFactory          # reference, unmodified
    from src import utils as rutils             # reference, unmodified

    torch.set_num_threads(8)
    torch.manual_seed(11)
    shape, patch, k, b, m, nb = (48, 64), 4, 3, 2, 2500, 5
    cfg = dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=4, smooth_weight=0.003, lut_superpixel_size=4,
               focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True,
               polarity_aware_batching=True, interpolation_scheme='mean', smooth_type='on_flow_to_tref')
    # trajectory_net.py:35-47
    net = nn.Sequential(nn.Linear(1, 64), nn.LeakyReLU(), nn.Linear(64, 64), nn.LeakyReLU(), nn.Linear(64, 64), nn.LeakyReLU(),
                        nn.Linear(64, k))
    rutils.initialize_weights(net)
    g = torch.Generator().manual_seed(12)
    num_pos = m // 2 + 3
    ev = torch.zeros(b, m, 6)
    ev[..., 0] = torch.rand(b, m, generator=g) * (shape[0] - 1)
    ev[..., 1] = torch.rand(b, m, generator=g) * (shape[1] - 1)
    t = torch.rand(b, m, generator=g)
    ev[..., 2] = t
    ev[:, :num_pos, 3] = 1
    ev[..., 4] = torch.clamp(torch.floor(t * nb), 0, nb - 1)
    ev[..., 5] = 1
    coeff_grid = (torch.randn(b, 1, 2 * k, *shape, generator=g) * 4.0).requires_grad_(True)
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    times = L.get_reconstruction_times('cpu').clone()
    times[0] = 0.37
    mask = rutils.get_optical_flow_tile_mask(shape, patch)
    coeffs, pos, _ = rutils.coeffs_grid_to_list(coeff_grid, mask, num_coeffs=k)
    # trajectory_net.py:101-111: basis at the times minus the basis at the anchor t = 0
    traj = rutils.compute_basis(coeffs, times, k, 'learned', net) - rutils.compute_basis(coeffs, torch.zeros(1), k, 'learned', net)
    traj = (traj + pos[None, :, None, :]).permute(0, 2, 1, 3).contiguous()
    traj.retain_grad()
    loss, log, misc = L.calc(traj, times, {'events': ev, 'num_pos_events': num_pos})
    loss.backward()
    out = dict(cfg_keys=np.array(sorted(cfg.keys())), cfg_vals=np.array([str(cfg[kk]) for kk in sorted(cfg.keys())]),
               patch=patch, num_basis=k, num_pos=num_pos, events=ev.numpy(), coeff_grid=coeff_grid.detach().numpy(),
               times=times.numpy(), trajectories=traj.detach().numpy(), loss=loss.item(),
               focus_loss=log['focus_loss'].item(), smooth_loss=float(log['smoothness_loss']),
               basis_at_times=net(times[..., None]).detach().numpy(),
               grad_trajectories=traj.grad.numpy(), grad_coeff_grid_at_tiles=coeff_grid.grad[..., mask].numpy())
    for name, prm in net.state_dict().items():
        out['net_' + name.replace('.', '_')] = prm.numpy()
    for name, prm in net.named_parameters():
        out['grad_net_' + name.replace('.', '_')] = prm.grad.numpy()
    np.savez_compressed(os.path.join(args.out, 'g11_learned_basis.npz'), **out)
    print(f'g11_learned_basis: loss={loss.item():.8f}  |d loss / d net| = {sum(float(p.grad.abs().sum()) for p in net.parameters()):.6e}')


if __name__ == '__main__':
    main()
