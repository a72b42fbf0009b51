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


# (translate60 / diverge+-45: scripts/dsec_inference.py:93 clamps the network's flow at 60 px -- flows up to there occur)
FLOW_FAMILIES = ('zero', 'translate10', 'translate20', 'translate40', 'translate60', 'diverge+30', 'diverge-30', 'diverge+45', 'diverge-45',
                 'rotate', 'shear', 'unet')


def _smooth_field(b, c, image_shape, g, cells=(6, 8)):
    """Low-pass random field [b, c, h, w] of unit variance: white noise on a coarse grid, bicubic upsampling (what the
    last up-convolutions of a UNet produce: structure at ~80 px, no pixel noise)."""
    h, w = image_shape
    z = torch.randn(b, c, cells[0] + 3, cells[1] + 3, generator=g)
    f = torch.nn.functional.interpolate(z, size=(h, w), mode='bicubic', align_corners=True)
    return f / f.std(dim=(-2, -1), keepdim=True).clamp_min(1e-6)


def synth_flow_coeff_grid(b, k, image_shape, family, seed=0):
    """Polynomial motion coefficients [b, 1, 2k, h, w] (first k channels y, next k x: utils/trajectories.py:15-32) of a flow
    field a trained network produces (trajectory_net.py:142-161), instead of i.i.d. noise per tile.  The displacement at t = 1
    is  T + A (pos - centre) + low-pass noise;  `family` picks T and A:
      zero            all coefficients 0 (the first training steps: the trajectory points are the exact lattice, every
                      neighbour search has exact distance ties)
      translateD      |T| = D px in a random direction per sample (+ 1 px of low-pass noise)
      diverge+30/-30  A = +-0.15 I  (divergence +-30 %: expansion thins the points to 0.76x, contraction packs them 1.38x
                      and empties a band of up to 48 px at every border)
      rotate          0.1 rad about the centre (32 px at the left / right border)
      shear           d x / d y = 0.15
      unet            T up to 20 px, A random with entries up to 0.1, 6 px of low-pass noise
    Higher orders (k > 1) carry 15 % / 5 % of the first as independent smooth fields, so trajectories curve a little."""
    g = torch.Generator().manual_seed(seed)
    h, w = image_shape
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32) - (h - 1) / 2, torch.arange(w, dtype=torch.float32) - (w - 1) / 2,
                            indexing='ij')
    out = torch.zeros(b, 1, 2 * k, h, w)
    if family == 'zero':
        return out
    for i in range(b):
        T = torch.zeros(2)
        A = torch.zeros(2, 2)
        noise = 1.0
        if family.startswith('translate'):
            mag = float(family[len('translate'):])
            ang = float(torch.rand(1, generator=g)) * 6.2831853
            T = mag * torch.tensor([torch.sin(torch.tensor(ang)), torch.cos(torch.tensor(ang))])
        elif family.startswith('diverge'):
            d = float(family[len('diverge'):]) / 100.0 / 2.0
            A = d * torch.eye(2)
        elif family == 'rotate':
            om = 0.1 * (1 if i % 2 == 0 else -1)
            A = torch.tensor([[0.0, -om], [om, 0.0]])
        elif family == 'shear':
            A = torch.tensor([[0.0, 0.0], [0.15, 0.0]])
        elif family == 'unet':
            mag = 20.0 * float(torch.rand(1, generator=g))
            ang = float(torch.rand(1, generator=g)) * 6.2831853
            T = mag * torch.tensor([torch.sin(torch.tensor(ang)), torch.cos(torch.tensor(ang))])
            A = (torch.rand(2, 2, generator=g) - 0.5) * 0.2
            noise = 6.0
        else:
            raise ValueError(f'unknown flow family {family!r}')
        fy = T[0] + A[0, 0] * yy + A[0, 1] * xx
        fx = T[1] + A[1, 0] * yy + A[1, 1] * xx
        sm = _smooth_field(1, 2 * k, image_shape, g)[0]
        out[i, 0, 0] = fy + noise * sm[0]
        out[i, 0, k] = fx + noise * sm[k]
        for j, frac in zip(range(1, k), (0.15, 0.05, 0.02, 0.01)):
            amp = frac * max(float(fy.abs().max()), float(fx.abs().max()), noise)
            out[i, 0, j] = amp * sm[j]
            out[i, 0, k + j] = amp * sm[k + j]
    return out


def synth_events_ragged(b, m_max, image_shape, num_bins, seed=0, pad_lo=0.3, pad_hi=0.6, time_sorted=True):
    """A batch as the DSEC collate builds it (loader.py:360-415): every sample has its own number of positive and negative
    events; each polarity block is padded with zero rows to the batch maximum of that polarity.  Sample 0 holds the maxima
    (no padding), the others are padded by a fraction in [pad_lo, pad_hi].  Returns ([b, M, 6], num_pos)."""
    g = torch.Generator().manual_seed(seed)
    h, w = image_shape
    half = m_max // 2
    npos = [half] + [int(half * (1 - (pad_lo + (pad_hi - pad_lo) * float(torch.rand(1, generator=g))))) for _ in range(b - 1)]
    nneg = [m_max - half] + [int((m_max - half) * (1 - (pad_lo + (pad_hi - pad_lo) * float(torch.rand(1, generator=g))))) for _ in range(b - 1)]
    ev = torch.zeros(b, m_max, 6)
    for i in range(b):
        for lo, cnt, pol in ((0, npos[i], 1.0), (half, nneg[i], 0.0)):
            y = torch.rand(cnt, generator=g) * (h - 1)
            x = torch.rand(cnt, generator=g) * (w - 1)
            t = torch.rand(cnt, generator=g)
            if time_sorted:
                t = torch.sort(t).values
            rows = ev[i, lo:lo + cnt]
            rows[:, 0] = y; rows[:, 1] = x; rows[:, 2] = t; rows[:, 3] = pol
            rows[:, 4] = torch.clamp(torch.floor(t * num_bins), 0, num_bins - 1)
            rows[:, 5] = 1
    return ev, half


def synth_trajectories(b, k, num_bins, image_shape, patch, family, seed=0, t_ref=0.41):
    """Trajectories [b, 1 + num_bins, n, 2] at (t_ref, bin mid-times) of the tile centres under a flow field of `family`
    (synth_flow_coeff_grid), built as TrajectoryNet.step builds them (trajectory_net.py:101-119)."""
    from .trajectories import get_optical_flow_tile_mask, coeffs_grid_to_list
    from .basis import compute_basis
    times = torch.cat((torch.tensor([t_ref]), bin_mid_times(num_bins)))
    mask = get_optical_flow_tile_mask(image_shape, patch)
    coeff = synth_flow_coeff_grid(b, k, image_shape, family, seed=seed)
    coeffs, pos, _ = coeffs_grid_to_list(coeff, mask, num_coeffs=k)
    traj = compute_basis(coeffs, times, k, 'polynomial') - compute_basis(coeffs, torch.zeros(1), k, 'polynomial')
    traj = (traj + pos[None, :, None, :]).permute(0, 2, 1, 3).contiguous()
    return traj, times
