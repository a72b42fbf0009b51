"""CPU ORACLE for event ingest -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the event wire format the loss consumes:
  * per-sample columns, time normalisation, bin index, in-image filter, polarity split
    (reference src/loader/dsec/loader.py:152-167, Sequence.get_data_sample);
  * zero padding with a valid flag and the positive-block / negative-block layout
    (loader.py:360-415, pad_events / sequence_collate_fn).
The collate half is pinned by tests/golden/g9_ingest.npz (made by the reference's own
sequence_collate_fn, oracle/gen_golden_ingest.py); the per-sample half restates lines 152-161, which
need the dataset files to run in the reference and are therefore not pinned by a reference run."""
import numpy as np


def sample_events(x, y, t_us, p, height, width, num_bins):
    """-> (pos_events [n,5], neg_events [m,5]) float32 columns (y, x, t, p, bin).  loader.py:152-167."""
    t = (t_us - t_us.min()) / (t_us.max() - t_us.min())                       # int64 -> float64
    bins = np.clip(np.searchsorted(np.linspace(0, 1, num_bins + 1), t) - 1, 0, None)
    ev = np.column_stack((y, x, t, p, bins))
    mask = (0 <= ev[:, 0]) & (ev[:, 0] < height) & (0 <= ev[:, 1]) & (ev[:, 1] < width)
    ev = ev[mask].astype('float32')
    return ev[ev[:, 3] == 1], ev[ev[:, 3] == 0]


def collate(samples):
    """samples: list of (pos, neg) -> (events [B, max_pos + max_neg, 6], num_pos_events).  loader.py:360-415."""
    max_pos = max(len(s[0]) for s in samples)
    max_neg = max(len(s[1]) for s in samples)
    out = np.zeros((len(samples), max_pos + max_neg, 6), dtype=np.float32)
    for b, (pos, neg) in enumerate(samples):
        out[b, :len(pos), :5] = pos
        out[b, :len(pos), 5] = 1
        out[b, max_pos:max_pos + len(neg), :5] = neg
        out[b, max_pos:max_pos + len(neg), 5] = 1
    return out, max_pos


def voxel_input(x, y, t_us, p):
    """(x, y, t, p) float32 as loader.py:135-138 hands them to VoxelGrid.convert."""
    t = (t_us - t_us[0]).astype('float32')
    t = t / t[-1]
    return np.stack((x.astype('float32'), y.astype('float32'), t, p.astype('float32')), -1)


def synth_raw(n, height, width, seed=0, spill=3.0, window_us=100000):
    """Raw window as the DSEC slicer + rectification deliver it: float32 coordinates that may leave the
    sensor, increasing int64 microsecond timestamps, polarity in {0, 1}."""
    g = np.random.default_rng(seed)
    x = (g.random(n) * (width - 1 + 2 * spill) - spill).astype('float32')
    y = (g.random(n) * (height - 1 + 2 * spill) - spill).astype('float32')
    t = np.sort(g.integers(0, window_us, n)).astype('int64') + 51_000_000_000
    p = (g.random(n) > 0.45).astype('float32')
    return x, y, t, p
