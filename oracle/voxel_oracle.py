"""CPU ORACLE for the voxel-grid builder -- TEST INFRASTRUCTURE ONLY.

Pure-torch restatement of reference src/loader/dsec/utils.py:29-77 (VoxelGrid.convert: trilinear
`put_(accumulate=True)` over the 8 corners, then mean/std or max normalisation of the non-zero
entries).  Pinned by tests/golden/g8_voxel_*.npz, produced by the unmodified reference
(oracle/gen_golden_voxel.py; that file of the reference imports only torch and numpy)."""
import torch


def voxel_grid(x, y, t, p, shape, norm_type='mean_std', quantile=0.0):
    """x, y, t, p: [N] float tensors (t increasing; normalised with its first and last element as in
    utils.py:35-36), shape (C, H, W).  Returns [C, H, W]."""
    C, H, W = shape
    grid = torch.zeros(C * H * W, dtype=torch.float32)
    tn = (C - 1) * (t - t[0]) / (t[-1] - t[0])
    x0, y0, t0 = x.int(), y.int(), tn.int()              # truncation, utils.py:38-40
    val = 2 * p - 1
    for xl in (x0, x0 + 1):
        for yl in (y0, y0 + 1):
            for tl in (t0, t0 + 1):
                m = (xl < W) & (xl >= 0) & (yl < H) & (yl >= 0) & (tl >= 0) & (tl < C)
                w = val * (1 - (xl - x).abs()) * (1 - (yl - y).abs()) * (1 - (tl - tn).abs())
                idx = H * W * tl.long() + W * yl.long() + xl.long()
                grid.put_(idx[m], w[m], accumulate=True)
    grid = grid.reshape(C, H, W)
    if quantile > 0:                                     # utils.py:57-61: clip at the (1 - quantile) quantile of |grid|
        thr = torch.quantile(grid.abs().view(-1), 1 - quantile)
        grid = torch.where(grid.abs() > thr, grid.sign() * thr, grid)
    if norm_type == 'mean_std':
        nz = torch.nonzero(grid, as_tuple=True)
        if nz[0].numel() > 0:
            mean, std = grid[nz].mean(), grid[nz].std()
            grid[nz] = (grid[nz] - mean) / std if std > 0 else grid[nz] - mean
    elif norm_type == 'max':
        mx = grid.abs().max()
        if mx > 0:
            grid = grid / mx
    return grid


def synth_raw_events(n, shape, seed=0, spill=2.0):
    """Seeded raw events as the DSEC slicer hands them to the voxel grid: rectified float coordinates
    that may fall slightly outside the sensor, increasing timestamps, polarity in {0, 1}."""
    g = torch.Generator().manual_seed(seed)
    C, H, W = shape
    x = torch.rand(n, generator=g) * (W - 1 + 2 * spill) - spill
    y = torch.rand(n, generator=g) * (H - 1 + 2 * spill) - spill
    t = torch.sort(torch.rand(n, generator=g)).values
    t = (t - t[0]) / (t[-1] - t[0])
    p = (torch.rand(n, generator=g) > 0.5).float()
    return x, y, t, p
