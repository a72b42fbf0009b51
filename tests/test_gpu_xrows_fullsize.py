"""The two UNPINNED extensions BASELINE.json's configs name, at the size their config names:
  * configs[2] "multi-scale IWE pyramid" at C3 size (B = 14 x 200k events, poly-k3) -- FocusLoss(pyramid_levels=3);
  * configs[3] "cubic B-spline trajectory basis" at C4 size (500k events, 41 bins) -- utils.trajectories_from_bspline on the device,
    through FocusLoss.calc.
The reference has neither (no IWE pyramid; raft-spline curves are Bezier, bezier.py:92-113), so there is
nothing to pin them to: what is checked is the DEFINITION each one documents, written with the CPU oracle's functions on the
LUT the device computed (the KNN itself is pinned elsewhere), brute-force K-nearest on sampled cells, and the gradient by a
directional finite difference on the device."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_gpu_fullsize import _brute_lut, _fail_fraction

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def _directional_fd(fn, x, d, eps):
    """(f(x + eps d) - f(x - eps d)) / (2 eps) in float64 from fp32 evaluations."""
    with torch.no_grad():
        fp = fn(x + eps * d).double().item()
        fm = fn(x - eps * d).double().item()
    return (fp - fm) / (2.0 * eps)


def test_pyramid_at_c3_size_matches_its_definition():
    import bench
    from motionpriorcmax_amd import LossFactory, ops
    from oracle import focus_oracle as O
    dev = _dev()
    wl = bench.WORKLOADS['C3']
    cfg = bench.loss_config(wl)
    levels = 3
    ev, num_pos, traj, times = bench.synth_inputs(wl, seed=5)
    assert ev.shape == (14, 200000, 6) and traj.shape[1] == 16
    L = LossFactory.get_loss_calculator('FOCUS', dict(cfg, pyramid_levels=levels))
    L1 = LossFactory.get_loss_calculator('FOCUS', cfg)
    evd, timesd = ev.to(dev), times.to(dev)
    batch = {'events': evd, 'num_pos_events': num_pos}
    tg = traj.to(dev).requires_grad_(True)
    loss, log, misc = L.calc(tg, timesd, batch)
    loss.backward()
    assert misc['iwes'].shape == (14, 1, 2, 480, 640)
    # the definition, with the oracle's functions, on the LUT the device computed
    shape = ops.make_shape(L._cfg, 14, 0, 0, traj.shape[2])
    ws = ops.alloc_workspace(shape, dev)
    lut, _, _, _ = ops.knn_lut_fwd(L._cfg, shape, traj.to(dev), ws)
    lut_c = lut.cpu()
    warped = O.warp_events(ev, lut_c, 4)
    _, raw = O.make_iwes(ev, warped, times[:1], (480, 640), True, True, True, num_pos)
    focus, cur = 0.0, raw
    per_level = []
    for lv in range(levels):
        per_level.append(1.0 / O.contrast_value(O.gaussian_blur3(cur), 'gradient_magnitude', cfg['focus_loss_norm']))
        focus = focus + per_level[-1]
        cur = F.avg_pool2d(cur, 2)
    so = O.FocusLossOracle(**cfg).smooth_loss(lut_c, None)
    assert abs(log['focus_loss'].item() - focus.item()) <= 1e-5 * abs(focus.item()), (log['focus_loss'].item(), focus.item())
    assert abs(loss.item() - (focus + so).item()) <= 1e-5 * abs((focus + so).item())
    # level 0 alone is the plain loss of the reference, bit for bit
    l1, log1, misc1 = L1.calc(traj.to(dev), timesd, batch)
    assert abs(log1['focus_loss'].item() - per_level[0].item()) <= 1e-5 * per_level[0].item()
    assert torch.equal(misc1['iwes'], misc['iwes'])
    # gradient: a smooth direction (every trajectory's reference-time point moved by the same vector, i.e. the whole flow table
    # shifted) -- central difference of the loss against <grad, d>
    d = torch.zeros_like(tg)
    d[:, 0, :, 0] = 0.6
    d[:, 0, :, 1] = -0.8
    fd = _directional_fd(lambda t: L.calc(t, timesd, batch)[0], tg.detach(), d, 2e-2)
    an = float((tg.grad.double() * d.double()).sum())
    assert abs(fd - an) <= 0.05 * max(abs(an), abs(fd)) + 1e-9, (fd, an)
    assert torch.isfinite(tg.grad).all()
    # bitwise reproducible from run to run
    t2 = traj.to(dev).requires_grad_(True)
    l2, _, _ = L.calc(t2, timesd, batch)
    l2.backward()
    assert torch.equal(l2, loss) and torch.equal(t2.grad, tg.grad)


def test_bspline_trajectories_at_c4_size_through_calc():
    import bench
    from motionpriorcmax_amd import LossFactory, ops, utils
    from motionpriorcmax_amd.utils.synth import synth_events, bin_mid_times
    from oracle import focus_oracle as O
    dev = _dev()
    wl = bench.WORKLOADS['C4']
    cfg = bench.loss_config(wl)
    assert cfg['num_bins'] == 41 and cfg['smooth_type'] == 'on_flow_to_next'
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    ev, num_pos = synth_events(1, wl['M'], (480, 640), 41, seed=4, pad_frac=0.02, time_sorted=True)
    g = torch.Generator().manual_seed(12)
    nctrl = 10                                                   # P_1 .. P_10 free, P_0 = 0: as many as the Bezier-10 of the shipped yaml
    params_c = torch.randn(1, 2 * nctrl, 120, 160, generator=g) * 2.0
    times = torch.cat((torch.tensor([0.41]), bin_mid_times(41)))
    params = params_c.to(dev).requires_grad_(True)
    traj, pos = utils.trajectories_from_bspline(params, times.to(dev), 4, (480, 640))       # on the device, autograd on
    assert traj.is_cuda and traj.shape == (1, 42, 19200, 2)
    # the basis: scipy's BSpline design matrix (clamped uniform cubic), columns 1..m-1; rows sum to 1 with column 0
    from scipy.interpolate import BSpline
    m, p = nctrl + 1, 3
    knots = np.concatenate((np.zeros(p), np.linspace(0, 1, m - p + 1), np.ones(p)))
    tq = np.minimum(times.numpy().astype(np.float64), 1 - 1e-15)
    dm = np.stack([BSpline(knots, np.eye(m)[i], p, extrapolate=False)(tq) for i in range(m)], 1)
    bm = utils.bspline_basis(times, m, p).numpy()
    np.testing.assert_allclose(bm, dm[:, 1:], atol=2e-6)
    traj_ref = torch.einsum('bcdhw,td->btchw', params_c.view(1, 2, nctrl, 120, 160), torch.from_numpy(dm[:, 1:]).float())
    traj_ref = torch.stack((traj_ref[:, :, 1], traj_ref[:, :, 0]), -1).reshape(1, 42, 19200, 2) + pos.float()[None, None]
    np.testing.assert_allclose(traj.detach().cpu().numpy(), traj_ref.numpy(), rtol=1e-5, atol=2e-4)
    evd, timesd = ev.to(dev), times.to(dev)
    # KNN LUT and flow_to_next against brute force on sampled cells
    shape = ops.make_shape(L._cfg, 1, 0, 0, 19200)
    ws = ops.alloc_workspace(shape, dev)
    td = traj.detach().contiguous()
    lut, nxt, state, _ = ops.knn_lut_fwd(L._cfg, shape, td, ws)
    assert _fail_fraction(L, shape, ws) < 0.02
    grid, hq, wq = O.lut_grid_points((480, 640), 4)
    sel = torch.cat((torch.randperm(hq * wq, generator=g)[:384], torch.tensor([0, 159, 160 * 119, 160 * 120 - 1])))
    q = grid[sel].to(dev)
    for t in (0, 20, 40):
        f, fn = _brute_lut(td[0], q, 32, t)
        assert (lut[0, t].reshape(-1, 2)[sel.to(dev)] - f).abs().max().item() < 1e-5
        if t < 40:
            assert (nxt[0, t].reshape(-1, 2)[sel.to(dev)] - fn).abs().max().item() < 1e-5
    # the loss through the plugin API against the oracle's event path and smoothness on the device's tables
    batch = {'events': evd, 'num_pos_events': num_pos}
    loss, log, misc = L.calc(traj, timesd, batch)
    loss.backward()
    Lo = O.FocusLossOracle(**cfg)
    fo, _, _ = Lo.event_path(ev, lut.cpu(), times[:1], num_pos)
    so = Lo.smooth_loss(lut.cpu(), nxt.cpu())
    assert abs(log['focus_loss'].item() - fo.item()) <= 1e-5 * abs(fo.item())
    assert abs(log['smoothness_loss'].item() - so.item()) <= 1e-5 * abs(so.item())
    assert abs(loss.item() - (fo + so).item()) <= 1e-5 * abs((fo + so).item())
    # the gradient reaches the control points (autograd through the basis): finite, non-zero, and a directional difference agrees
    assert params.grad is not None and torch.isfinite(params.grad).all() and float(params.grad.abs().sum()) > 0
    d = torch.zeros_like(params)
    d[:, nctrl - 1] = 0.5            # last control point of x
    d[:, 2 * nctrl - 1] = -0.5       # last control point of y

    def f(pr):
        tj, _ = utils.trajectories_from_bspline(pr, timesd, 4, (480, 640))
        return L.calc(tj, timesd, batch)[0]
    fd = _directional_fd(f, params.detach(), d, 2e-2)
    an = float((params.grad.double() * d.double()).sum())
    assert abs(fd - an) <= 0.08 * max(abs(an), abs(fd)) + 1e-9, (fd, an)
