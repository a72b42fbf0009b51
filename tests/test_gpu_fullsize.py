"""The shapes `bench.py` reports, at their full size, through the same code path (needs an MI355X).

BASELINE configs[2] (C3: 480x640, batch 14 x 200k events, poly-k3) and configs[3] (C4: 480x640, 500k events, 41 bins,
Bezier degree 10 via `trajectories_from_bezier` ON THE DEVICE, smoothness on flow_to_next, weight 0.06 --
config/exe/trajectory_inference/experiment/raft-spline_evimo2-300ms_ours-selfsup_Tab2L5.yaml:21-35, focus.py:170-178).
Workgroup swizzles, bucket capacities, strip counts and the KNN fast-path / fallback split all depend on B and num_bins,
so these sizes are checked themselves:
  * size-independent properties: every sample of the batch equals a B = 1 run of the same sample bit for bit (samples
    couple only through the scalar `val`), mass conservation of the raw IWE;
  * the oracle where it finishes in seconds: the event path (LUT given) for the whole batch, the smoothness terms, a
    brute-force K-nearest search on sampled cells;
  * the in-library global-atomic path (`debug_atomic_path`) as a second implementation of the event kernels;
  * the device-side training step (coefficient grid -> basis -> calc -> backward to the grid, trajectory_net.py:101-161)
    against the gradient sums the goldens hold.
Tolerances: loss rel 1e-5 (BASELINE target), LUT / IWE atol 1e-5 (x max), gradients by relative L2."""
import ctypes

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def _rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _bench_case(name, seed=1):
    import bench
    from motionpriorcmax_amd import LossFactory
    wl = bench.WORKLOADS[name]
    ev, num_pos, traj, times = bench.synth_inputs(wl, seed=seed)
    cfg = bench.loss_config(wl)
    return wl, cfg, ev, num_pos, traj, times, LossFactory.get_loss_calculator('FOCUS', cfg)


def _fail_fraction(L, shape, ws):
    from motionpriorcmax_amd import _lib as C
    off = C.lib().mpc_knn_fail_list_offset(ctypes.byref(shape))
    n = int(ws[off:off + 4].view(torch.int32).item())
    return n / float(shape.B * shape.nb * shape.hq * shape.wq)


def _brute_lut(traj_b, q, K, t):
    """mean flow to t_ref (and to the next bin) of the K nearest trajectories of bin t, torch on the device."""
    pts = traj_b[1 + t]
    d = ((q[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    idx = torch.sort(d, dim=1, stable=True).indices[:, :K]
    f = (traj_b[0] - pts)[idx].mean(1)
    nxt = (traj_b[2 + t] - pts)[idx].mean(1) if 2 + t < traj_b.shape[0] else None
    return f, nxt


def test_c3_full_batch():
    """configs[2] at B = 14 x 200k: per-sample bitwise equality with B = 1 runs, mass conservation, the loss against the
    oracle's event path on the GPU LUT, the atomic path as cross-check of loss and gradient."""
    from motionpriorcmax_amd import ops, LossFactory
    from oracle import focus_oracle as O
    dev = _dev()
    wl, cfg, ev, num_pos, traj, times, L = _bench_case('C3')
    B = wl['B']
    assert B == 14 and ev.shape == (14, 200000, 6)
    evd, trajd, timesd = ev.to(dev), traj.to(dev), times.to(dev)
    t_ref = timesd[:1]
    # stage level, whole batch
    shape = ops.make_shape(L._cfg, B, 0, 0, traj.shape[2])
    ws = ops.alloc_workspace(shape, dev)
    lut, _, state, _ = ops.knn_lut_fwd(L._cfg, shape, trajd, ws)
    assert _fail_fraction(L, shape, ws) < 0.005          # the strip kernel serves (nearly) every query of this shape
    _, blur, raw = ops.EventFocusFn.apply(lut, evd, t_ref, L._cfg, num_pos)
    # every sample alone: bit-identical LUT and raw IWE (integer accumulators; samples are independent)
    for b in (0, 6, 13):
        shape1 = ops.make_shape(L._cfg, 1, 0, 0, traj.shape[2])
        lut1, _, _, _ = ops.knn_lut_fwd(L._cfg, shape1, trajd[b:b + 1].contiguous(), ops.alloc_workspace(shape1, dev))
        assert torch.equal(lut1[0], lut[b]), f'LUT of sample {b} differs from its B = 1 run'
        _, blur1, raw1 = ops.EventFocusFn.apply(lut1, evd[b:b + 1].contiguous(), t_ref, L._cfg, num_pos)
        assert torch.equal(raw1[0], raw[b]) and torch.equal(blur1[0], blur[b]), f'IWE of sample {b} differs from its B = 1 run'
    # mass conservation over the whole batch (weights: valid * (1 - |t - t_ref|), zero if warped outside, focus.py:201-214)
    bidx = torch.arange(B, device=dev)[:, None]
    it = evd[..., 4].long()
    iy = torch.div(evd[..., 0], 4, rounding_mode='floor').long()
    ix = torch.div(evd[..., 1], 4, rounding_mode='floor').long()
    pos = lut[bidx, it, iy, ix, 0] + evd[..., :2]
    w = evd[..., 5] * (1 - (evd[..., 2] - t_ref[0]).abs().clamp(0, 1))
    oob = (pos[..., 0] > 480) | (pos[..., 1] > 640) | (pos[..., 0] < 0) | (pos[..., 1] < 0)
    w = torch.where(oob, torch.zeros_like(w), w)
    fl = torch.floor(pos + 1e-6)
    fr = pos - fl
    y0, x0 = fl[..., 0].long(), fl[..., 1].long()
    mass = 0.0
    for dy, dx, tw in ((0, 0, (1 - fr[..., 0]) * (1 - fr[..., 1])), (1, 0, fr[..., 0] * (1 - fr[..., 1])),
                       (0, 1, (1 - fr[..., 0]) * fr[..., 1]), (1, 1, fr[..., 0] * fr[..., 1])):
        ok = (y0 + dy >= 0) & (y0 + dy < 480) & (x0 + dx >= 0) & (x0 + dx < 640)
        mass += (tw * w * ok).double().sum().item()
    assert abs(raw.double().sum().item() - mass) <= 1e-5 * mass
    # the whole loss through the plugin API; oracle event path + smoothness on the GPU LUT
    tg = trajd.clone().requires_grad_(True)
    loss, log, misc = L.calc(tg, timesd, {'events': evd, 'num_pos_events': num_pos})
    loss.backward()
    assert torch.equal(misc['iwes'].reshape(blur.shape), blur)
    Lo = O.FocusLossOracle(**cfg)
    lut_c = lut.cpu()
    fo, _, rawo = Lo.event_path(ev, lut_c, times[:1], num_pos)
    so = Lo.smooth_loss(lut_c, None)
    assert abs(log['focus_loss'].item() - fo.item()) <= 1e-5 * abs(fo.item())
    assert abs(log['smoothness_loss'].item() - so.item()) <= 1e-5 * abs(so.item())
    assert abs(loss.item() - (fo + so).item()) <= 1e-5 * abs((fo + so).item())
    np.testing.assert_allclose(raw[:2].cpu().numpy(), rawo[:2].numpy(), atol=1e-5 * float(rawo.max()))
    # second implementation of the event kernels (global float atomics), same process
    La = LossFactory.get_loss_calculator('FOCUS', dict(cfg, debug_atomic_path=True))
    ta = trajd.clone().requires_grad_(True)
    la, _, _ = La.calc(ta, timesd, {'events': evd, 'num_pos_events': num_pos})
    la.backward()
    assert abs(la.item() - loss.item()) <= 2e-6 * abs(loss.item())
    # (float atomics round differently from the exact integer sums: a few near-zero Sobel responses change sign under the
    # 'l1' norm, SURVEY.md section 4 -- seen as 1e-7 or 3e-4 from run to run)
    assert _rel_l2(tg.grad, ta.grad) < 2e-3
    assert torch.isfinite(tg.grad).all() and float(tg.grad.abs().sum()) > 0


def test_c4_full_size_bezier_on_device_flow_to_next():
    """configs[3]: nb = 41, 500k events, trajectories from Bezier control points built ON THE DEVICE (8f-4), smoothness on
    flow_to_next with weight 0.06."""
    import bench
    from motionpriorcmax_amd import ops, utils, LossFactory
    from oracle import focus_oracle as O
    dev = _dev()
    wl = bench.WORKLOADS['C4']
    cfg = bench.loss_config(wl)
    assert cfg['num_bins'] == 41 and cfg['smooth_type'] == 'on_flow_to_next' and cfg['smooth_weight'] == 0.06
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    from motionpriorcmax_amd.utils.synth import synth_events, bin_mid_times
    ev, num_pos = synth_events(1, wl['M'], (480, 640), 41, seed=3, pad_frac=0.02, time_sorted=True)
    g = torch.Generator().manual_seed(11)
    params_c = torch.randn(1, 20, 120, 160, generator=g) * 2.0
    times = torch.cat((torch.tensor([0.41]), bin_mid_times(41)))
    params = params_c.to(dev).requires_grad_(True)
    traj, pos = utils.trajectories_from_bezier(params, times.to(dev), 4, (480, 640))          # device tensors, autograd on
    assert traj.is_cuda and traj.shape == (1, 42, 19200, 2)
    traj_ref, _ = utils.trajectories_from_bezier(params_c, times, 4, (480, 640))              # the same on the host
    np.testing.assert_allclose(traj.detach().cpu().numpy(), traj_ref.numpy(), rtol=1e-5, atol=1e-4)
    evd, timesd = ev.to(dev), times.to(dev)
    # KNN LUT and flow_to_next against brute force on sampled cells
    shape = ops.make_shape(L._cfg, 1, 0, 0, 19200)
    ws = ops.alloc_workspace(shape, dev)
    td = traj.detach().contiguous()
    lut, nxt, state, _ = ops.knn_lut_fwd(L._cfg, shape, td, ws)
    assert nxt is not None and nxt.shape == (1, 40, 120, 160, 1, 2)
    assert _fail_fraction(L, shape, ws) < 0.01
    grid, hq, wq = O.lut_grid_points((480, 640), 4)
    sel = torch.randperm(hq * wq, generator=g)[:384]
    sel = torch.cat((sel, torch.tensor([0, 1, 159, 160 * 119, 160 * 120 - 1, 160 * 2 + 2])))       # corners, edges
    q = grid[sel].to(dev)
    for t in (0, 20, 39, 40):
        f, fn = _brute_lut(td[0], q, 32, t)
        assert (lut[0, t].reshape(-1, 2)[sel.to(dev)] - f).abs().max().item() < 1e-5
        if t < 40:
            assert (nxt[0, t].reshape(-1, 2)[sel.to(dev)] - fn).abs().max().item() < 1e-5
    # full loss; oracle event path on the GPU LUT, oracle smoothness on the GPU flow_to_next
    loss, log, misc = L.calc(traj, timesd, {'events': evd, 'num_pos_events': num_pos})
    loss.backward()
    Lo = O.FocusLossOracle(**cfg)
    fo, iwo, _ = Lo.event_path(ev, lut.cpu(), times[:1], num_pos)
    so = Lo.smooth_loss(lut.cpu(), nxt.cpu())
    assert abs(log['focus_loss'].item() - fo.item()) <= 1e-5 * abs(fo.item())
    assert abs(log['smoothness_loss'].item() - so.item()) <= 1e-5 * abs(so.item())
    assert abs(loss.item() - (fo + so).item()) <= 1e-5 * abs((fo + so).item())
    np.testing.assert_allclose(misc['iwes'][0, 0].cpu().numpy(), iwo[0].detach().numpy(), atol=1e-5 * float(iwo.max()))
    # the gradient reaches the Bezier parameters; the atomic path agrees
    assert params.grad is not None and torch.isfinite(params.grad).all() and float(params.grad.abs().sum()) > 0
    La = LossFactory.get_loss_calculator('FOCUS', dict(cfg, debug_atomic_path=True))
    p2 = params_c.to(dev).requires_grad_(True)
    t2, _ = utils.trajectories_from_bezier(p2, timesd, 4, (480, 640))
    la, _, _ = La.calc(t2, timesd, {'events': evd, 'num_pos_events': num_pos})
    la.backward()
    assert abs(la.item() - loss.item()) <= 2e-6 * abs(loss.item())
    assert _rel_l2(params.grad, p2.grad) < 2e-3          # ('l1' norm: sign flips of near-zero responses, see above)


def test_c4_at_the_yaml_training_batch_of_six():
    """configs[3] at the batch size its yaml trains with (`...Tab2L5.yaml` training.batch_size: 6; SURVEY.md 8d "B=1 (and 6)"):
    6 x 500k events, 41 bins, smoothness on flow_to_next.  Every sample must equal its own B = 1 run bit for bit (LUT,
    flow_to_next, IWEs, gradient of its trajectories up to the 1/val factor shared by the batch), the LUT is checked against
    brute force on sampled cells of every sample, and the global-atomic event kernels give the same loss and gradient."""
    import bench
    from motionpriorcmax_amd import ops, utils, LossFactory
    from oracle import focus_oracle as O
    from motionpriorcmax_amd.utils.synth import synth_events, bin_mid_times
    dev = _dev()
    wl = bench.WORKLOADS['C4']
    cfg = bench.loss_config(wl)
    B = 6
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    ev, num_pos = synth_events(B, wl['M'], (480, 640), 41, seed=13, pad_frac=0.02, time_sorted=True)
    g = torch.Generator().manual_seed(17)
    params = (torch.randn(B, 20, 120, 160, generator=g) * 2.0).to(dev)
    times = torch.cat((torch.tensor([0.41]), bin_mid_times(41)))
    timesd, evd = times.to(dev), ev.to(dev)
    traj, _ = utils.trajectories_from_bezier(params, timesd, 4, (480, 640))
    traj = traj.detach().contiguous()
    # stage level: LUT and flow_to_next of the batch against brute force, every sample
    shape = ops.make_shape(L._cfg, B, 0, 0, 19200)
    ws = ops.alloc_workspace(shape, dev)
    lut, nxt, _, _ = ops.knn_lut_fwd(L._cfg, shape, traj, ws)
    assert _fail_fraction(L, shape, ws) < 0.01
    grid, hq, wq = O.lut_grid_points((480, 640), 4)
    sel = torch.cat((torch.randperm(hq * wq, generator=g)[:96], torch.tensor([0, 159, 160 * 119, 160 * 120 - 1])))
    q = grid[sel].to(dev)
    for b in range(B):
        for t in (0, 23, 40):
            f, fn = _brute_lut(traj[b], q, 32, t)
            assert (lut[b, t].reshape(-1, 2)[sel.to(dev)] - f).abs().max().item() < 1e-5
            if t < 40:
                assert (nxt[b, t].reshape(-1, 2)[sel.to(dev)] - fn).abs().max().item() < 1e-5
    del ws
    # the whole loss at B = 6, and sample 4 alone
    tg = traj.clone().requires_grad_(True)
    loss, log, misc = L.calc(tg, timesd, {'events': evd, 'num_pos_events': num_pos})
    loss.backward()
    b1 = 4
    t1 = traj[b1:b1 + 1].clone().requires_grad_(True)
    l1, log1, misc1 = L.calc(t1, timesd, {'events': evd[b1:b1 + 1].contiguous(), 'num_pos_events': num_pos})
    l1.backward()
    assert torch.equal(misc['iwes'][b1], misc1['iwes'][0])
    # focus = 1 / mean over ALL images of the batch: the per-sample gradients of the focus term differ by the ratio of the
    # squared values; the smoothness term is a mean over the batch as well (ratio B).  Compare directions on the focus-only part:
    Lf = LossFactory.get_loss_calculator('FOCUS', dict(cfg, smooth_weight=0.0))
    tgf = traj.clone().requires_grad_(True); lf, _, _ = Lf.calc(tgf, timesd, {'events': evd, 'num_pos_events': num_pos}); lf.backward()
    t1f = traj[b1:b1 + 1].clone().requires_grad_(True); l1f, _, _ = Lf.calc(t1f, timesd, {'events': evd[b1:b1 + 1].contiguous(), 'num_pos_events': num_pos}); l1f.backward()
    # d(1/val)/dx = -(1/val^2) dval/dx, val = mean over B*P images: grad_batch[b] = grad_single * (val_single^2 / val_batch^2) / B
    ratio = (lf.item() ** 2) / (l1f.item() ** 2) / B
    assert _rel_l2(tgf.grad[b1], t1f.grad[0] * ratio) < 1e-5
    # oracle: event path and smoothness on the GPU LUT of the batch (one sample's IWEs compared; all enter the loss)
    Lo = O.FocusLossOracle(**cfg)
    fo, iwo, _ = Lo.event_path(ev, lut.cpu(), times[:1], num_pos)
    so = Lo.smooth_loss(lut.cpu(), nxt.cpu())
    assert abs(log['focus_loss'].item() - fo.item()) <= 1e-5 * abs(fo.item())
    assert abs(log['smoothness_loss'].item() - so.item()) <= 1e-5 * abs(so.item())
    assert abs(loss.item() - (fo + so).item()) <= 1e-5 * abs((fo + so).item())
    np.testing.assert_allclose(misc['iwes'][2, 0].cpu().numpy(), iwo[2].detach().numpy(), atol=1e-5 * float(iwo.max()))
    # second implementation of the event kernels
    La = LossFactory.get_loss_calculator('FOCUS', dict(cfg, debug_atomic_path=True))
    ta = traj.clone().requires_grad_(True)
    la, _, _ = La.calc(ta, timesd, {'events': evd, 'num_pos_events': num_pos})
    la.backward()
    assert abs(la.item() - loss.item()) <= 2e-6 * abs(loss.item())
    assert _rel_l2(tg.grad, ta.grad) < 2e-3


@pytest.mark.parametrize('norm', ['l2'])
def test_num_tref_2_at_c3_size_two_implementations_agree(norm):
    """num_tref = 2 at the C3 shape (B = 4 x 200k events): the reference times as samples of the num_tref == 1 kernels (strip search,
    tile gather, LDS-tiled event path; `calc`'s default) against the general kernels (one launch per stage for both reference times,
    global-atomic event path): loss, images and gradient of two independent implementations.  The 'l2' focus norm: no sign() whose
    flips would loosen the gradient bound."""
    import bench
    from motionpriorcmax_amd import LossFactory
    dev = _dev()
    wl = dict(bench.WORKLOADS['C3']); wl['B'] = 4
    ev, _, traj, times = bench.synth_inputs(wl, seed=3, trefs=(0.0, 1.0))
    cfg = dict(bench.loss_config(wl), num_tref=2, scale_iwe_by_dt=False, polarity_aware_batching=False, focus_loss_norm=norm)
    outs = []
    for as_samples in (True, False):
        L = LossFactory.get_loss_calculator('FOCUS', dict(cfg, trefs_as_samples=as_samples))
        t = traj.to(dev).requires_grad_(True)
        loss, log, misc = L.calc(t, times.to(dev), {'events': ev.to(dev)})
        loss.backward()
        outs.append((loss.item(), log['smoothness_loss'].item(), misc['iwes'], t.grad))
    a, g = outs
    assert a[2].shape == g[2].shape == (4, 2, 480, 640)
    assert abs(a[0] - g[0]) <= 2e-6 * abs(g[0]) and abs(a[1] - g[1]) <= 1e-5 * abs(g[1]) + 1e-12
    assert float((a[2] - g[2]).abs().max()) <= 1e-5 * float(g[2].abs().max())
    assert _rel_l2(a[3], g[3]) < 1e-4


@pytest.mark.parametrize('name,B', [('C2', 1), ('C3', 2), ('C4', 1)])
def test_strip_kernel_serves_the_benchmark_shapes(name, B):
    """The KNN fast path (strip kernel) must serve nearly every query of the shapes the benchmark times; what it hands to
    the per-query fallback is listed in the workspace (mpc_knn_fail_list_offset)."""
    import bench
    from motionpriorcmax_amd import ops, LossFactory
    dev = _dev()
    wl = dict(bench.WORKLOADS[name]); wl['B'] = B
    _, _, traj, _ = bench.synth_inputs(wl, seed=5)
    L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    shape = ops.make_shape(L._cfg, B, 0, 0, traj.shape[2])
    ws = ops.alloc_workspace(shape, dev)
    ops.knn_lut_fwd(L._cfg, shape, traj.to(dev), ws)
    torch.cuda.synchronize()
    assert _fail_fraction(L, shape, ws) < 0.01


@pytest.mark.parametrize('name', ['g1_allflags', 'g5a_dct3_l2', 'g5b_poly3'])
def test_training_step_on_device_matches_reference_gradient(name):
    """`TrajectoryNet.step` shaped harness on the device (trajectory_net.py:101-119,142-161): coefficient grid
    [B,1,2k,H,W] -> coeffs_grid_to_list -> compute_basis -> + pixel positions -> calc -> backward to the GRID, all tensors
    on the GPU; the goldens hold the reference's loss and the absolute sum of its coefficient-grid gradient."""
    from motionpriorcmax_amd import utils, LossFactory
    g = load_golden(name)
    dev = _dev()
    cfg = g['cfg']
    k, bt, patch = int(g['num_basis']), str(g['basis_type']), int(g['patch'])
    cg = torch.from_numpy(g['coeff_grid']).to(dev).requires_grad_(True)
    times = torch.from_numpy(g['times']).to(dev)
    mask = utils.get_optical_flow_tile_mask(cfg['image_shape'], patch).to(dev)
    coeffs, pos, _ = utils.coeffs_grid_to_list(cg, mask, num_coeffs=k)
    traj = utils.compute_basis(coeffs, times, k, bt) - utils.compute_basis(coeffs, torch.zeros(1, device=dev), k, bt)
    traj = (traj + pos[None, :, None, :]).permute(0, 2, 1, 3)
    assert traj.is_cuda
    np.testing.assert_allclose(traj.detach().cpu().numpy(), g['trajectories'], atol=1e-5)
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    batch = {'events': torch.from_numpy(g['events']).to(dev)}
    if cfg['polarity_aware_batching']:
        batch['num_pos_events'] = int(g['num_pos'])
    loss, _, _ = L.calc(traj, times, batch)
    loss.backward()
    assert abs(loss.item() - g['loss']) <= 1e-5 * abs(g['loss'])
    got = float(cg.grad.abs().sum())
    want = float(g['grad_coeff_grid_abs_sum'])
    assert abs(got - want) <= 1e-3 * want, (got, want)
    # the WHOLE gradient of the grid against the reference's autograd (stored at the tile centres, zero elsewhere): by
    # relative L2, and element by element under the accounting rule of tests/test_gpu_grad_accounting.py -- 2e-5 of the
    # maximum + 1e-4 relative, with at most 0.1 % of the elements beyond it (a vote next to a pixel whose Sobel response is
    # zero up to rounding may take the other sign() in the two implementations)
    gc = cg.grad.cpu().numpy()
    m = mask.cpu().numpy()
    ref = g['grad_coeff_grid_at_tiles']
    assert np.abs(gc[..., ~m]).max() == 0.0
    d = gc[..., m] - ref
    rel = np.linalg.norm(d) / np.linalg.norm(ref)
    bad = np.abs(d) > 2e-5 * np.abs(ref).max() + 1e-4 * np.abs(ref)
    assert rel < 1e-3, rel
    assert bad.mean() <= 1e-3, (bad.mean(), rel)


@pytest.mark.parametrize('scheme', ['mean', 'iwd'])
def test_dsec_size_end_to_end_gradient_tight_with_l2_norm(scheme):
    """The end-to-end gradient bounds of the 'l1' cases are loose by necessity (a near-zero Sobel response may take either sign: 2e-3);
    with `focus_loss_norm: l2` the objective is smooth, so the WHOLE backward -- event kernels, smoothness, KNN gather, far backward,
    combine -- can be held to 1e-4 at DSEC size: against CPU autograd through the oracle's event path and a gather over the neighbour
    sets the device itself reports (the sets are checked against brute force elsewhere; here they make the CPU side affordable)."""
    import bench
    from motionpriorcmax_amd import ops, LossFactory
    from motionpriorcmax_amd.utils import synth
    from oracle import focus_oracle as O
    dev = _dev()
    wl = dict(bench.WORKLOADS['C3'], B=2)
    cfg = dict(bench.loss_config(wl), focus_loss_norm='l2', interpolation_scheme=scheme)
    ev, num_pos, _, times = bench.synth_inputs(wl, seed=21)
    # a UNet-like smooth field: far queries and the tail launch take part
    traj, _ = synth.synth_trajectories(2, 3, wl['nb'], (480, 640), 4, 'unet', seed=17)
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    evd, timesd = ev.to(dev), times.to(dev)
    tg = traj.to(dev).requires_grad_(True)
    loss, log, _ = L.calc(tg, timesd, {'events': evd, 'num_pos_events': num_pos})
    loss.backward()
    idx = ops.knn_indices(L._cfg, traj.to(dev)).cpu().long()                       # [B, nb, Q, K]
    tc = traj.clone().requires_grad_(True)
    B, nb, Q, K = idx.shape
    flow = tc[:, :1] - tc[:, 1:]                                                    # traj(t_ref) - traj(t_mid)  [B, nb, n, 2]
    fk = flow.gather(2, idx.reshape(B, nb, Q * K, 1).expand(B, nb, Q * K, 2)).reshape(B, nb, Q, K, 2)                  # focus.py:151-154
    if scheme == 'mean':
        lut = fk.mean(3)                                                                                                # focus.py:156
    else:
        # 'iwd' (focus.py:157-163): weights 1 / (d + 1e-9), normalised, constants of the backward -- the tile gather's weights are one
        # hardware reciprocal each (1 ulp), the bound below covers their rounding with three orders of magnitude to spare
        with torch.no_grad():
            grid, _, _ = O.lut_grid_points((480, 640), 4)
            pk = traj[:, 1:].gather(2, idx.reshape(B, nb, Q * K, 1).expand(B, nb, Q * K, 2)).reshape(B, nb, Q, K, 2)
            d = ((grid[None, None, :, None, :] - pk) ** 2).sum(-1)
            w = 1.0 / (d + 1e-9)
            w = w / w.sum(3, keepdim=True)
        lut = (w[..., None] * fk).sum(3)
    lut = lut.reshape(B, nb, 120, 160, 1, 2)
    Lo = O.FocusLossOracle(**cfg)
    fo, _, _ = Lo.event_path(ev, lut, times[:1], num_pos)
    so = Lo.smooth_loss(lut, None)
    (fo + so).backward()
    assert abs(loss.item() - (fo + so).item()) <= 1e-5 * abs((fo + so).item())
    assert _rel_l2(tg.grad.cpu(), tc.grad) < 1e-4, _rel_l2(tg.grad.cpu(), tc.grad)
