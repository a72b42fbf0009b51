"""Seeded synthetic DSEC-shaped inputs for the benchmark, smoke test and tests."""
import torch


def bin_mid_times(num_bins: int) -> torch.Tensor:
    e = torch.linspace(0, 1, num_bins + 1)
    return (e[:-1] + e[1:]) / 2


def synth_events(b, m, image_shape, num_bins, seed=0, pad_frac=0.0, time_sorted=False, num_pos=None):
    """[b, m, 6] rows (y, x, t, p, bin, valid) laid out as the DSEC loader + collate do
    (reference loader.py:152-167,360-395): positive block, then negative block, zero padding rows
    at the end of each block; `time_sorted` orders each block by timestamp like real recordings."""
    g = torch.Generator().manual_seed(seed)
    h, w = image_shape
    num_pos = m // 2 if num_pos is None else num_pos
    ev = torch.zeros(b, m, 6)
    ev[..., 0] = torch.rand(b, m, generator=g) * (h - 1)
    ev[..., 1] = torch.rand(b, m, generator=g) * (w - 1)
    t = torch.rand(b, m, generator=g)
    if time_sorted:
        t = torch.cat((torch.sort(t[:, :num_pos], 1).values, torch.sort(t[:, num_pos:], 1).values), 1)
    ev[..., 2] = t
    ev[:, :num_pos, 3] = 1
    ev[..., 4] = torch.clamp(torch.floor(t * num_bins), 0, num_bins - 1)
    ev[..., 5] = 1
    if pad_frac > 0:
        for lo, hi in ((0, num_pos), (num_pos, m)):
            n_pad = int((hi - lo) * pad_frac)
            if n_pad:
                ev[:, hi - n_pad:hi] = 0
    return ev, num_pos
