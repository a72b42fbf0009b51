"""CPU-only checks: the C-ABI library loads and exports what include/mpcmax.h declares, shape
validation, and the host-side mirror of the reference's loss plugin API."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def _header_functions():
    src = open(os.path.join(ROOT, 'include', 'mpcmax.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(mpc_[a-z_0-9]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from motionpriorcmax_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 10
    for n in names:
        assert hasattr(L, n), f'{n} declared in include/mpcmax.h but not exported'
    assert sorted(_lib.EXPORTS) == names


def _shape(**kw):
    from motionpriorcmax_amd import _lib
    d = dict(B=2, M=1000, Mp=500, nb=5, T=1, H=48, W=64, sp=4, hq=12, wq=16, n=192, K=4, flags=0)
    d.update(kw)
    return _lib.Shape(**d)


def test_workspace_bytes_and_validation_are_host_only():
    from motionpriorcmax_amd import _lib
    L = _lib.lib()
    s = _shape()
    n1 = L.mpc_workspace_bytes(ctypes.byref(s))
    assert n1 > 0
    n2 = L.mpc_workspace_bytes(ctypes.byref(_shape(B=4)))
    assert n2 > n1
    bad = _shape(hq=13)
    assert L.mpc_workspace_bytes(ctypes.byref(bad)) == -2
    assert b'hq/wq' in L.mpc_last_error_string()
    assert L.mpc_workspace_bytes(ctypes.byref(_shape(T=2, flags=_lib.F_SCALE_BY_DT))) == -2
    assert L.mpc_workspace_bytes(ctypes.byref(_shape(H=2))) == -2


def test_null_arguments_are_rejected_without_touching_the_gpu():
    from motionpriorcmax_amd import _lib
    L = _lib.lib()
    s = _shape()
    assert L.mpc_event_splat_fwd(ctypes.byref(s), None, None, None, None, None, None) == -1
    assert L.mpc_knn_lut_fwd(ctypes.byref(s), None, None, None, None, None, None, None) == -1
    assert L.mpc_contrast_fwd(ctypes.byref(s), None, None, None, None, None) == -1


CFG = dict(image_shape=(48, 64), num_tref=1, num_bins=5, num_knn=4, smooth_weight=0.003,
           lut_superpixel_size=4, focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True,
           mask_image_border=True, polarity_aware_batching=True, interpolation_scheme='mean',
           smooth_type='on_flow_to_tref')


def test_factory_and_attributes():
    from motionpriorcmax_amd import LossFactory, TrajectoryLossBase
    L = LossFactory.get_loss_calculator('FOCUS', dict(CFG, some_unknown_key=1), profiler=None)
    assert isinstance(L, TrajectoryLossBase)
    assert L.is_needing_offsets is True
    assert hasattr(L.imager, 'create_iwe')
    with pytest.raises(ValueError, match='Unsupported loss type'):
        LossFactory.get_loss_calculator('NOPE', CFG)
    with pytest.raises(AssertionError):
        LossFactory.get_loss_calculator('FOCUS', dict(CFG, num_tref=2))
    t = L.get_reconstruction_times('cpu')
    assert t.shape == (6,) and 0 <= t[0] < 1
    np.testing.assert_allclose(t[1:].numpy(), [0.1, 0.3, 0.5, 0.7, 0.9], atol=1e-6)
    L3 = LossFactory.get_loss_calculator('FOCUS', dict(CFG, num_tref=3, scale_iwe_by_dt=False,
                                                       polarity_aware_batching=False))
    np.testing.assert_allclose(L3.get_reconstruction_times('cpu')[:3].numpy(), [0, 0.5, 1])


def test_cpu_tensors_fail_loudly():
    from motionpriorcmax_amd import LossFactory
    g = load_golden('g3_squeeze_k1')
    L = LossFactory.get_loss_calculator('FOCUS', g['cfg'])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        L.calc(torch.from_numpy(g['trajectories']), torch.from_numpy(g['times']),
               {'events': torch.from_numpy(g['events'])})
    with pytest.raises(AssertionError):
        Lp = LossFactory.get_loss_calculator('FOCUS', CFG)
        Lp.calc(torch.zeros(1, 6, 192, 2), torch.zeros(6), {'events': torch.zeros(1, 10, 6)})


@pytest.mark.parametrize('name', ['g1_allflags', 'g5a_dct3_l2', 'g5b_poly3'])
def test_host_trajectory_helpers_match_reference(name):
    """The caller-side trajectory construction (TrajectoryNet.calculate_trajectories_at_t,
    trajectory_net.py:101-119) built from this package's utils."""
    from motionpriorcmax_amd import utils
    g = load_golden(name)
    cg = torch.from_numpy(g['coeff_grid'])
    times = torch.from_numpy(g['times'])
    k, bt = int(g['num_basis']), str(g['basis_type'])
    mask = utils.get_optical_flow_tile_mask(g['cfg']['image_shape'], int(g['patch']))
    coeffs, pos, _ = utils.coeffs_grid_to_list(cg, mask, num_coeffs=k)
    traj = utils.compute_basis(coeffs, times, k, bt) - utils.compute_basis(coeffs, torch.zeros(1), k, bt)
    traj = (traj + pos[None, :, None, :]).permute(0, 2, 1, 3)
    np.testing.assert_allclose(traj.numpy(), g['trajectories'], atol=1e-5)


def test_bernstein_basis_matches_reference():
    from motionpriorcmax_amd import utils
    g = load_golden('g6_bezier10')
    bm = utils.bernstein_basis(g['timestamps'], 10)
    p = torch.from_numpy(g['params']).view(2, 2, 10, 6, 8)
    np.testing.assert_allclose(torch.einsum('bdphw,tp->tbdhw', p, bm).numpy(), g['flows'], atol=1e-5)


def test_bezier_adapter_matches_reference_curves():
    """8f-4: trajectories sampled from Bezier curves = tile centres + the reference's
    `get_flow_from_reference` (g6, (x, y) channel order swapped to (y, x)); gradient reaches the parameters."""
    from motionpriorcmax_amd import utils
    g = load_golden('g6_bezier10')
    params = torch.from_numpy(g['params']).requires_grad_(True)          # [2, 20, 6, 8]
    traj, pos = utils.trajectories_from_bezier(params, g['timestamps'], 4, (24, 32))
    assert traj.shape == (2, 6, 48, 2) and pos.shape == (48, 2)
    assert pos[0].tolist() == [2, 2] and pos[-1].tolist() == [22, 30]
    flows = torch.from_numpy(g['flows'])                                 # [n_t, B, 2 (x, y), h, w]
    disp = (traj - pos.float()[None, None]).reshape(2, 6, 6, 8, 2)
    np.testing.assert_allclose(disp[..., 0].detach().numpy(), flows[:, :, 1].permute(1, 0, 2, 3).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(disp[..., 1].detach().numpy(), flows[:, :, 0].permute(1, 0, 2, 3).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(disp[:, 0].detach().numpy(), 0.0, atol=0)            # anchor t = 0
    np.testing.assert_allclose(disp[:, -1, ..., 1].detach().numpy(), g['flow_t1'][:, 0], rtol=1e-6, atol=1e-6)
    traj.sum().backward()
    assert params.grad is not None and float(params.grad.abs().sum()) > 0


def test_cubic_bspline_adapter_matches_scipy():
    """UNPINNED extension (BASELINE.json configs[3] names a cubic B-spline; the reference has none): the clamped uniform
    B-spline basis against scipy.interpolate.BSpline, partition of unity, zero flow at the anchor, autograd to the control points."""
    import numpy as np
    import torch
    from scipy.interpolate import BSpline
    from motionpriorcmax_amd import utils
    rng = np.random.default_rng(0)
    for m, p in ((4, 3), (7, 3), (11, 3), (6, 2)):
        t = np.concatenate(([0.0, 1.0], rng.random(40)))
        inner = np.linspace(0, 1, m - p + 1)
        knots = np.concatenate((np.zeros(p), inner, np.ones(p)))
        ref = np.stack([BSpline(knots, np.eye(m)[i], p, extrapolate=False)(np.minimum(t, 1 - 1e-15)) for i in range(m)], 1)
        got = utils.bspline_basis(t, m, p).numpy()
        np.testing.assert_allclose(got, ref[:, 1:], atol=2e-6)
        np.testing.assert_allclose(got.sum(1) + ref[:, 0], 1.0, atol=2e-6)            # partition of unity with N_0
        assert np.all(got[0] == 0) and got[1, -1] == 1.0                                # t = 0: only N_0; t = 1: only N_{m-1}
    params = torch.randn(2, 2 * 6, 6, 8, requires_grad=True)
    times = torch.tensor([0.0, 0.3, 0.77, 1.0])
    traj, pos = utils.trajectories_from_bspline(params, times, 4, (24, 32))
    assert traj.shape == (2, 4, 48, 2)
    assert torch.equal(traj[:, 0], pos.float()[None].expand(2, -1, -1))               # zero flow at the anchor
    last = params.view(2, 2, 6, 6, 8)[:, :, -1]                                        # t = 1: the last control point, (x, y) -> (y, x)
    want = torch.stack((last[:, 1], last[:, 0]), -1).reshape(2, 48, 2) + pos.float()[None]
    assert torch.allclose(traj[:, 3], want, atol=1e-6)
    traj.sum().backward()
    assert torch.isfinite(params.grad).all() and float(params.grad.abs().sum()) > 0
