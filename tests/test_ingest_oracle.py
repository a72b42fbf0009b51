"""Event-ingest oracle against the reference's own collate (CPU)."""
import os

import numpy as np

from conftest import GOLDEN
from oracle import ingest_oracle as I


def load_ingest():
    z = np.load(os.path.join(GOLDEN, 'g9_ingest.npz'))
    return {k: z[k] for k in z.files}


def test_ingest_oracle_matches_reference_collate():
    g = load_ingest()
    H, W, nb = int(g['H']), int(g['W']), int(g['nb'])
    samples = []
    for b, n in enumerate(g['counts']):
        samples.append(I.sample_events(g['x'][b, :n], g['y'][b, :n], g['t'][b, :n], g['p'][b, :n], H, W, nb))
    ev, num_pos = I.collate(samples)
    assert num_pos == int(g['num_pos_events'])
    np.testing.assert_array_equal(ev, g['events'])
    # sanity of the restated per-sample half: bins follow (e_i, e_i+1], rows are inside the sensor
    valid = ev[..., 5] == 1
    assert ((ev[..., 0] >= 0) & (ev[..., 0] < H))[valid].all() and ((ev[..., 1] >= 0) & (ev[..., 1] < W))[valid].all()
    t, bins = ev[..., 2][valid].astype(np.float64), ev[..., 4][valid]
    assert (bins == np.clip(np.ceil(t * nb) - 1, 0, nb - 1)).mean() > 0.999
