"""Motion bases evaluated at the reconstruction times (host side, plain torch: 1 + num_bins
times x ~19k trajectories).  Mirrors reference src/utils/basis.py:4-46 and the Bernstein basis of
src/models/raft_spline/curves/bezier.py:69-107."""
import math
import weakref

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


def basis_values(times, num_basis, basis_type, basis_network=None):
    """The basis functions of `compute_basis` at arbitrary times: times [...] -> [..., k] (same formulas, same dtype as `times`).
    Used by the per-event continuous-time warp (FocusLoss.calc_per_event_basis: UNPINNED extension), which evaluates the basis at
    every event's own timestamp instead of at the num_bins bin mid-times."""
    if basis_type == "dct":
        k_idx = torch.arange(1, num_basis + 1, device=times.device)
        return np.sqrt(2.0) * torch.cos(np.pi / 2.0 * ((2 * times[..., None] + 1) * k_idx))
    if basis_type == "learned":
        return basis_network(times[..., None])
    if basis_type == "polynomial":
        k_idx = torch.arange(1, num_basis + 1, device=times.device)
        return times[..., None] ** k_idx
    raise ValueError(basis_type)


_BASIS_CACHE = {}


def _cached_basis(kind, times, *args):
    """The [n_t, d] basis matrix of `kind` for these timestamps: the float64 evaluation runs on the host (as the reference's does,
    bezier.py:104-107) and costs milliseconds -- a training loop asks for the same bin mid-times every step, so the last few
    matrices are kept (keyed on the timestamps' bytes)."""
    t = np.ascontiguousarray(np.asarray(times, dtype=np.float64).reshape(-1))
    key = (kind, args, t.tobytes())
    m = _BASIS_CACHE.get(key)
    if m is None:
        if len(_BASIS_CACHE) > 32:
            _BASIS_CACHE.clear()
        m = _BASIS_CACHE[key] = (_bernstein_basis_eval if kind == 'bernstein' else _bspline_basis_eval)(t, *args)
    return m


_DEV_BASIS_CACHE = []          # [(times tensor, its version, kind, args, device, dtype, matrix on the device)]
_TILE_POS_CACHE = {}


def _device_basis(kind, times, args, device, dtype):
    """The basis matrix on `device`.  A training loop passes the SAME timestamps tensor every step: for a tensor seen before (the
    object itself, unmodified -- the entry holds a reference, so its address cannot be handed to another tensor) the matrix is
    returned without reading the timestamps back to the host (a device synchronisation, the float64 evaluation and an upload: 9 ms
    of a 0.3 ms step)."""
    if torch.is_tensor(times):
        # an entry belongs to a tensor OBJECT (weak reference: the cache keeps no caller tensor alive; a dead entry cannot match, so
        # an address handed to another tensor is no hit), unmodified as far as torch can tell (_version) and still at the same
        # storage address with the same shape (`.data` assignments and set_() do not bump the version)
        _DEV_BASIS_CACHE[:] = [e for e in _DEV_BASIS_CACHE if e[0]() is not None]
        for ent in _DEV_BASIS_CACHE:
            if ent[0]() is times and ent[1] == (times._version, times.data_ptr(), tuple(times.shape)) and ent[2:6] == (kind, args, device, dtype):
                return ent[6]
        t = times.detach().cpu().numpy()
    else:
        t = times
    m = _cached_basis(kind, t, *args).to(device, dtype)
    if torch.is_tensor(times):
        if len(_DEV_BASIS_CACHE) >= 8:
            _DEV_BASIS_CACHE.pop(0)
        _DEV_BASIS_CACHE.append((weakref.ref(times), (times._version, times.data_ptr(), tuple(times.shape)), kind, args, device, dtype, m))
    return m


def _tile_positions(H, W, tile_size, device, dtype):
    """(tile centres as a LongTensor on the host -- what the adapters return --, the same on `device` in `dtype`), built once.
    The adapters hand the host tensor out as it is (a copy per step would be host time of a host-bound step): their docstrings say
    that `pixel_positions` is shared and must not be modified in place."""
    from .trajectories import get_optical_flow_tile_mask
    key = (H, W, tile_size, str(device), dtype)
    ent = _TILE_POS_CACHE.get(key)
    if ent is None:
        if len(_TILE_POS_CACHE) > 16:
            _TILE_POS_CACHE.clear()
        pos = torch.nonzero(get_optical_flow_tile_mask((H, W), tile_size))
        ent = _TILE_POS_CACHE[key] = (pos, pos.to(device, dtype))
    return ent


def _curve_trajectories(params, bm, tile_size, H, W, scale):
    """pos + scale * (basis x control points), (y, x) order: on the GPU one kernel each way (ops.CurveTrajFn, csrc/curves.hip:
    mpc_curve_traj_fwd / _bwd -- a training step calls this every iteration and a B = 1 step is bound by the host); for CPU tensors
    (the host-side tests against the reference's curves) the same in plain torch."""
    B, c2, h, w = params.shape
    d = c2 // 2
    pos, pos_dev = _tile_positions(H, W, tile_size, params.device, params.dtype)
    assert pos.shape[0] == h * w, 'image shape must be a multiple of the tile size'
    if params.is_cuda and params.dtype == torch.float32 and d <= 16:
        from .. import ops
        return ops.CurveTrajFn.apply(params, bm, pos_dev, float(scale)), pos
    flow = torch.einsum('bcdhw,td->btchw', params.view(B, 2, d, h, w), bm) * scale  # [B, n_t, (x, y), h, w]
    disp = torch.stack((flow[:, :, 1], flow[:, :, 0]), dim=-1).reshape(B, bm.shape[0], h * w, 2)
    return disp + pos_dev[None, None], pos


def bernstein_basis(times, degree):
    """[n_t] -> [n_t, degree]: C(d,i) (1-t)^(d-i) t^i for i = 1..d (P0 == 0), float64 then fp32.  (The caller's own copy: the cached
    matrix is shared by every later step.)"""
    return _cached_basis('bernstein', times, int(degree)).clone()


def _bernstein_basis_eval(times, degree):
    t = np.asarray(times, dtype=np.float64).reshape(-1)
    out = np.zeros((t.size, degree))
    for d_idx in range(degree):
        i = d_idx + 1
        out[:, d_idx] = math.comb(degree, i) * (1 - t) ** (degree - i) * t ** i
    return torch.from_numpy(out).float()


def trajectories_from_bezier(params, times, tile_size, image_shape, scale=1.0):
    """RAFT-spline adapter: sample Bezier flow curves at the tile centres as `trajectories`
    for `FocusLoss.calc`.

    params [B, 2*d, h, w] with h = H // tile_size, w = W // tile_size, viewed [B, 2, d, h, w] with dim 1 in
    (x, y) order (reference src/models/raft_spline/curves/base.py:88-89, polynomial.py:60-61); the flow at time
    t is `CurveBase.get_flow_from_reference(t)` = sum_i B_i(t) P_i (bezier.py:92-113), which is zero at the
    anchor t = 0.  Returns (trajectories [B, n_t, n, 2] in (y, x) pixel coordinates, pixel_positions [n, 2]) with
    the tile centres of `get_optical_flow_tile_mask` as start points -- the layout `calc` expects
    (focus.py:66-72); `pixel_positions` is a cached tensor shared by all calls: do not modify it in place.  `scale` multiplies the flow (8.0 if the curve lives on RAFT's 1/8 grid units).
    Differentiable w.r.t. `params` (plain torch: 2*d*n_t multiply-adds per tile)."""
    from .trajectories import get_optical_flow_tile_mask
    B, c2, h, w = params.shape
    H, W = (int(v) for v in image_shape)
    assert c2 % 2 == 0 and h == H // tile_size and w == W // tile_size, (params.shape, image_shape, tile_size)
    d = c2 // 2
    bm = _device_basis('bernstein', times, (int(d),), params.device, params.dtype)    # [n_t, d]
    return _curve_trajectories(params, bm, tile_size, H, W, scale)


def bspline_basis(times, num_ctrl, degree=3):
    """[n_t] -> [n_t, num_ctrl - 1]: the clamped (open uniform) B-spline basis functions N_1 .. N_{m-1} of `degree` on [0, 1]
    (control point 0 is fixed at zero, as P0 of the reference's Bezier curves: the flow from the reference time vanishes at
    t = 0).  Cox-de Boor in float64, then fp32.  UNPINNED EXTENSION: the reference has no B-spline curve
    (src/models/raft_spline/curves holds Bezier and polynomial curves only) -- BASELINE.json's configs[3]
    names a cubic B-spline, so the basis is provided, default OFF, and checked against scipy.interpolate.BSpline."""
    return _cached_basis('bspline', times, int(num_ctrl), int(degree)).clone()          # (the caller's own copy)


def _bspline_basis_eval(times, num_ctrl, degree=3):
    m, p = int(num_ctrl), int(degree)
    assert m >= p + 1, 'need at least degree + 1 control points'
    t = np.clip(np.asarray(times, dtype=np.float64).reshape(-1), 0.0, 1.0)
    inner = np.linspace(0.0, 1.0, m - p + 1)
    knots = np.concatenate((np.zeros(p), inner, np.ones(p)))            # m + p + 1 knots
    # degree 0
    N = np.zeros((t.size, m + p))
    for i in range(m + p):
        lo, hi = knots[i], knots[i + 1]
        if hi > lo:
            N[:, i] = ((t >= lo) & (t < hi)) | ((t == 1.0) & (hi == 1.0))
    for q in range(1, p + 1):
        Nn = np.zeros((t.size, m + p - q))
        for i in range(m + p - q):
            a = knots[i + q] - knots[i]
            b = knots[i + q + 1] - knots[i + 1]
            if a > 0:
                Nn[:, i] += (t - knots[i]) / a * N[:, i]
            if b > 0:
                Nn[:, i] += (knots[i + q + 1] - t) / b * N[:, i + 1]
        N = Nn
    return torch.from_numpy(N[:, 1:m]).float()


def trajectories_from_bspline(params, times, tile_size, image_shape, scale=1.0, degree=3):
    """As trajectories_from_bezier, for a clamped uniform B-spline flow curve (cubic by default) with control points
    P_1 .. P_{m-1} = params [B, 2*(m-1), h, w] ((x, y) channel order) and P_0 = 0.  UNPINNED EXTENSION (see bspline_basis)."""
    from .trajectories import get_optical_flow_tile_mask
    B, c2, h, w = params.shape
    H, W = (int(v) for v in image_shape)
    assert c2 % 2 == 0 and h == H // tile_size and w == W // tile_size, (params.shape, image_shape, tile_size)
    d = c2 // 2
    bm = _device_basis('bspline', times, (int(d + 1), int(degree)), params.device, params.dtype)      # [n_t, d]
    return _curve_trajectories(params, bm, tile_size, H, W, scale)
