"""Motion bases evaluated at the reconstruction times (host side, plain torch: 1 + num_bins
times x ~19k trajectories).  Mirrors reference src/utils/basis.py:4-46 and the Bernstein basis of
src/models/raft_spline/curves/bezier.py:69-107."""
import math

import numpy as np
import torch


def compute_basis(coeffs, times, num_basis, basis_type, basis_network=None):
    """coeffs [b, s, 2, n, k], times [n_t] -> trajectories [b, n, n_t, 2] summed over scales s."""
    if basis_type == "dct":
        k_idx = torch.arange(1, num_basis + 1, device=coeffs.device)
        basis = np.sqrt(2.0) * torch.cos(np.pi / 2.0 * ((2 * times[..., None] + 1) * k_idx[None, None, :]))
    elif basis_type == "learned":
        basis = basis_network(times[..., None])
    elif basis_type == "polynomial":
        k_idx = torch.arange(1, num_basis + 1, device=coeffs.device)
        basis = times[..., None] ** k_idx[None, None, :]
    else:
        raise ValueError
    cy = coeffs[..., 0, :, :]
    cx = coeffs[..., 1, :, :]
    ty = torch.sum(basis[..., None, :, :] * cy[..., None, :], dim=-1)
    tx = torch.sum(basis[..., None, :, :] * cx[..., None, :], dim=-1)
    return torch.sum(torch.stack([ty, tx], dim=-1), dim=1)


def bernstein_basis(times, degree):
    """[n_t] -> [n_t, degree]: C(d,i) (1-t)^(d-i) t^i for i = 1..d (P0 == 0), float64 then fp32."""
    t = np.asarray(times, dtype=np.float64).reshape(-1)
    out = np.zeros((t.size, degree))
    for d_idx in range(degree):
        i = d_idx + 1
        out[:, d_idx] = math.comb(degree, i) * (1 - t) ** (degree - i) * t ** i
    return torch.from_numpy(out).float()
