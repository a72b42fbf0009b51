#!/usr/bin/env python3
"""Golden vectors for the voxel-grid builder from the UNMODIFIED reference
(src/loader/dsec/utils.py imports only torch and numpy, so no stand-ins are needed).

    python oracle/gen_golden_voxel.py [--ref /root/reference] [--out tests/golden]"""
import argparse
import importlib.util
import os
import sys

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(__file__), '..', 'tests', 'golden'))
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location('ref_dsec_utils', os.path.join(args.ref, 'src/loader/dsec/utils.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
    from oracle.voxel_oracle import synth_raw_events
    for name, shape, n, norm, seed, quant in (('g8_voxel_meanstd', (5, 24, 32), 4000, 'mean_std', 1, 0),
                                               ('g8_voxel_max', (3, 20, 28), 1500, 'max', 2, 0),
                                               ('g8_voxel_raw', (15, 30, 40), 6000, None, 3, 0),
                                               ('g8_voxel_q05_meanstd', (5, 24, 32), 9000, 'mean_std', 4, 0.05),
                                               ('g8_voxel_q10_raw', (3, 20, 28), 5000, None, 5, 0.1),
                                               ('g8_voxel_q02_max', (4, 22, 30), 12000, 'max', 6, 0.02)):
        x, y, t, p = synth_raw_events(n, shape, seed)
        vg = mod.VoxelGrid(shape, norm_type=norm, quantile=quant)
        out = vg.convert({'p': p, 't': t, 'x': x, 'y': y})
        np.savez_compressed(os.path.join(args.out, name + '.npz'), x=x.numpy(), y=y.numpy(), t=t.numpy(),
                            p=p.numpy(), shape=np.array(shape), norm=str(norm), quantile=np.float64(quant), grid=out.numpy())
        print(name, tuple(out.shape), float(out.abs().sum()))


if __name__ == '__main__':
    main()
