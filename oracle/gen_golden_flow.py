#!/usr/bin/env python3
"""Golden vectors for dense flow + flow metrics from the UNMODIFIED reference
(src/utils/flow.py, src/utils/trajectories.py), imported with the third-party stand-ins of oracle/stubs
(torchvision.resize forwards to torch's own anti-aliased bicubic interpolate).

    python oracle/gen_golden_flow.py [--ref /root/reference] [--out tests/golden]"""
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
    from src import utils as rutils             # reference, unmodified
    sys.path.insert(2, os.path.join(here, '..'))
    from oracle.flow_oracle import synth_flow_case

    out = {}
    # dense flow: the validation path of trajectory_net.py:124-140 (coeff grid -> list -> flow at tiles -> dense)
    for tag, (H, W), ps, B, seed in (('a', (48, 64), 4, 2, 1), ('b', (40, 56), 8, 1, 2), ('c', (36, 60), 3, 2, 3)):
        g = torch.Generator().manual_seed(seed)
        mask = rutils.get_optical_flow_tile_mask((H, W), ps)
        coeff_grid = torch.randn(B, 1, 2, H, W, generator=g) * 3
        coeffs, pix, _ = rutils.coeffs_grid_to_list(coeff_grid, mask, num_coeffs=1)
        t0 = rutils.compute_basis(coeffs, torch.zeros(1), 1, 'polynomial')
        t1 = rutils.compute_basis(coeffs, torch.ones(1), 1, 'polynomial')
        traj_flow = (t1 - t0)[..., 0, :]
        dense, patch = rutils.dense_flow_from_traj(traj_flow, pix, ps, (H, W))
        out.update({f'df_{tag}_traj_flow': traj_flow.numpy(), f'df_{tag}_pix': pix.numpy(),
                    f'df_{tag}_ps': np.array(ps), f'df_{tag}_shape': np.array((H, W)),
                    f'df_{tag}_dense': dense.numpy(), f'df_{tag}_patch': patch.numpy()})
        print('dense', tag, tuple(dense.shape), float(dense.abs().sum()))
    # metrics: every argument combination of flow.py:18-70
    for tag, B, (H, W), wm, wsc, seed in (('a', 3, (24, 32), True, True, 5), ('b', 2, (20, 28), False, False, 6),
                                           ('c', 1, (16, 16), True, False, 7), ('d', 2, (12, 20), False, True, 8)):
        gt, pr, em, ts = synth_flow_case(B, H, W, seed, wm, wsc)
        if tag == 'c':
            em = em[:, 0]                      # the 3-dim event-mask form (flow.py:44-45)
        err = rutils.calculate_flow_error(gt, pr, em, ts)
        out.update({f'fe_{tag}_gt': gt.numpy(), f'fe_{tag}_pred': pr.numpy(),
                    f'fe_{tag}_err': np.array([float(err[k]) for k in ('EPE', '1PE', '2PE', '3PE', 'AE')], np.float64)})
        if em is not None:
            out[f'fe_{tag}_mask'] = em.numpy()
        if ts is not None:
            out[f'fe_{tag}_scale'] = ts.numpy()
        print('metrics', tag, {k: float(v) for k, v in err.items()})
    np.savez_compressed(os.path.join(args.out, 'g10_flow.npz'), **out)


if __name__ == '__main__':
    main()
