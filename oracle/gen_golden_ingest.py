#!/usr/bin/env python3
"""Golden vectors for the event collate (loader.py:360-415) from the UNMODIFIED reference.
loader.py imports third-party packages this image lacks (cv2, h5py, numba, imageio, hdf5plugin, pandas);
none of them is used by pad_events / sequence_collate_fn, so empty stand-ins (oracle/stubs) satisfy the
import.  Per-sample event arrays come from oracle/ingest_oracle.sample_events."""
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
    sys.path.insert(2, os.path.join(here, '..'))
    import torch
    from src.loader.dsec.loader import sequence_collate_fn          # reference, unmodified
    from oracle import ingest_oracle as I
    H, W, nb = 48, 64, 5
    raws, batch = [], []
    for b, n in enumerate((1500, 900, 1200)):
        x, y, t, p = I.synth_raw(n, H, W, seed=60 + b)
        pos, neg = I.sample_events(x, y, t, p, H, W, nb)
        raws.append((x, y, t, p))
        batch.append({'pos_events': torch.from_numpy(pos), 'neg_events': torch.from_numpy(neg),
                      'timestamp': torch.tensor([0, 1]), 'voxel': torch.zeros(1), 'file_index': torch.tensor(b)})
    out = sequence_collate_fn(batch)
    N = max(len(r[0]) for r in raws)
    pad = lambda a, dt: np.stack([np.concatenate((r, np.zeros(N - len(r), dt))) for r in a])
    np.savez_compressed(os.path.join(args.out, 'g9_ingest.npz'), H=H, W=W, nb=nb,
                        counts=np.array([len(r[0]) for r in raws], dtype=np.int32),
                        x=pad([r[0] for r in raws], 'float32'), y=pad([r[1] for r in raws], 'float32'),
                        t=pad([r[2] for r in raws], 'int64'), p=pad([r[3] for r in raws], 'float32'),
                        events=out['events'].numpy(), num_pos_events=int(out['num_pos_events']))
    print('g9_ingest', tuple(out['events'].shape), int(out['num_pos_events']))


if __name__ == '__main__':
    main()
