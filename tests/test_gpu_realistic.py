"""The KNN flow LUT and its backward on the point sets a trained network produces (smooth flow fields: translation up to
40 px, divergence +-30 %, rotation, shear, UNet-like mixtures; zero flow = the exact lattice) instead of white-noise
coefficients: every cell of sampled (sample, bin) slices against a brute-force K-nearest search on the device
(focus.py:129-137 is exact for any point set), the gradient against autograd through that brute-force gather, and the share
of the queries the strip kernel hands to its fallback."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

H, W, SP, PATCH, K, NB = 480, 640, 4, 4, 32, 15


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda', 0)


def _loss(smooth_type='on_flow_to_tref'):
    from motionpriorcmax_amd import LossFactory
    return LossFactory.get_loss_calculator('FOCUS', dict(
        image_shape=(H, W), num_tref=1, num_bins=NB, num_knn=K, smooth_weight=0.003, lut_superpixel_size=SP,
        focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True, polarity_aware_batching=True,
        interpolation_scheme='mean', smooth_type=smooth_type))


def _brute(traj_b, t, q):
    """K nearest of every query among the points of bin t, ties to the lowest index (stable sort), in query chunks."""
    pts = traj_b[1 + t]
    idx = []
    for c in range(0, q.shape[0], 2048):
        d = ((q[c:c + 2048, None, :] - pts[None, :, :]) ** 2).sum(-1)
        idx.append(torch.sort(d, dim=1, stable=True).indices[:, :K])
    return torch.cat(idx)


# (translate60 / diverge+-45: scripts/dsec_inference.py:93 clamps the network's flow at 60 px)
FAMILIES = ['zero', 'translate10', 'translate40', 'translate60', 'diverge+30', 'diverge-30', 'diverge+45', 'diverge-45', 'rotate', 'shear', 'unet']


@pytest.mark.parametrize('family', FAMILIES)
def test_knn_lut_and_gradient_on_smooth_flow_fields(family):
    from motionpriorcmax_amd import ops
    from motionpriorcmax_amd.utils import synth
    from oracle import focus_oracle as O
    dev = _dev()
    B = 2
    traj, _ = synth.synth_trajectories(B, 3, NB, (H, W), PATCH, family, seed=23)
    L = _loss()
    td = traj.to(dev).requires_grad_(True)
    lut, _ = ops.KnnLutFn.apply(td, L._cfg)
    g = torch.Generator().manual_seed(5)
    wgt = torch.randn(lut.shape, generator=g).to(dev)
    (lut * wgt).sum().backward()
    grid, _, _ = O.lut_grid_points((H, W), SP)
    q = grid.to(dev)
    tb = traj.to(dev).requires_grad_(True)
    tot = 0.
    for b, t in ((0, 0), (0, NB - 1), (1, NB // 2)):
        idx = _brute(tb[b].detach(), t, q)
        ref = (tb[b, 0] - tb[b, 1 + t])[idx].mean(1)
        got = lut[b, t].reshape(-1, 2)
        assert torch.allclose(got, ref.detach(), atol=1e-4, rtol=1e-5), (family, b, t, float((got - ref).abs().max()))
        tot = tot + (ref * wgt[b, t].reshape(-1, 2)).sum()
    tot.backward()
    # gradient of the three slices: d traj(t_mid) of each, compared slice by slice (d traj(t_ref) sums over all bins)
    for b, t in ((0, 0), (0, NB - 1), (1, NB // 2)):
        a, r = td.grad[b, 1 + t], tb.grad[b, 1 + t]
        assert float((a - r).norm()) <= 1e-5 * float(r.norm()) + 1e-7, (family, b, t, float((a - r).norm() / r.norm()))


@pytest.mark.parametrize('family,limit', [('white', 0.002), ('zero', 0.002), ('translate10', 0.03), ('translate40', 0.10),
                                          ('diverge+30', 0.10), ('diverge-30', 0.15), ('rotate', 0.06), ('shear', 0.06), ('unet', 0.08)])
def test_strip_kernel_serves_most_queries_of_smooth_flow_fields(family, limit):
    """What the fast path hands to the fallback (the queries inside bands the flow emptied, and the radius estimate's misses)
    stays a small share for every family; rounds 1-3 had one radius from the mean density: 2 % .. 35 %."""
    import bench
    from motionpriorcmax_amd import ops, _lib as C
    from motionpriorcmax_amd.utils import synth
    dev = _dev()
    B = 4
    if family == 'white':
        _, _, traj, _ = bench.synth_inputs(dict(bench.WORKLOADS['C3'], B=B), seed=5)
    else:
        traj, _ = synth.synth_trajectories(B, 3, NB, (H, W), PATCH, family, seed=7)
    L = _loss()
    shape = ops.make_shape(L._cfg, B, 0, 0, traj.shape[2])
    ws = ops.alloc_workspace(shape, dev)
    ops.knn_lut_fwd(L._cfg, shape, traj.to(dev), ws)
    torch.cuda.synchronize()
    off = C.lib().mpc_knn_fail_list_offset(ctypes.byref(shape))
    n = int(ws[off:off + 4].view(torch.int32).item())
    frac = n / float(shape.B * shape.nb * shape.hq * shape.wq)
    assert frac <= limit, (family, frac)


def test_full_step_on_smooth_flow_matches_oracle_event_path():
    """calc + backward on a UNet-like field with a ragged batch (30-60 % padding rows): loss against the oracle's event path on
    the GPU LUT, the gradient finite and bitwise reproducible."""
    from motionpriorcmax_amd.utils import synth
    from oracle import focus_oracle as O
    dev = _dev()
    B, M = 3, 60000
    traj, times = synth.synth_trajectories(B, 3, NB, (H, W), PATCH, 'unet', seed=3)
    ev, num_pos = synth.synth_events_ragged(B, M, (H, W), NB, seed=4)
    L = _loss()
    outs = []
    for _ in range(2):
        td = traj.to(dev).requires_grad_(True)
        loss, log, misc = L.calc(td, times.to(dev), {'events': ev.to(dev), 'num_pos_events': num_pos})
        loss.backward()
        outs.append((loss.detach().clone(), td.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][1]).all()
    from motionpriorcmax_amd import ops
    lut, _ = ops.KnnLutFn.apply(traj.to(dev), L._cfg)
    cfg = dict(L._kwargs) if hasattr(L, '_kwargs') else None
    Lo = O.FocusLossOracle(image_shape=(H, W), num_tref=1, num_bins=NB, num_knn=K, smooth_weight=0.003, lut_superpixel_size=SP,
                           focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True,
                           polarity_aware_batching=True, interpolation_scheme='mean', smooth_type='on_flow_to_tref')
    f, _, _ = Lo.event_path(ev, lut.cpu(), times[:1], num_pos)
    sm = Lo.smooth_loss(lut.cpu(), None)
    ref = float(f + sm)
    assert abs(float(outs[0][0]) - ref) <= 1e-5 * abs(ref), (float(outs[0][0]), ref)


@pytest.mark.parametrize('family', ['unet', 'translate40'])
def test_knn_lut_l1_norm_at_dsec_density(family):
    """dist_norm = 'l1' at the DSEC density: the ball of the ring bound is a diamond (half the square), so the search radius is
    4 cells and the general tile kernel serves the forward (knn_device.h: knn_square_need, knn.hip: mpc_knn_r_init): LUT and
    gradient of sampled slices against a brute-force L1 K-nearest search (focus.py:133-134)."""
    from motionpriorcmax_amd import LossFactory, ops
    from motionpriorcmax_amd.utils import synth
    from oracle import focus_oracle as O
    dev = _dev()
    B = 1
    traj, _ = synth.synth_trajectories(B, 3, NB, (H, W), PATCH, family, seed=31)
    L = LossFactory.get_loss_calculator('FOCUS', dict(
        image_shape=(H, W), num_tref=1, num_bins=NB, num_knn=K, smooth_weight=0.003, lut_superpixel_size=SP,
        focus_loss_norm='l1', dist_norm='l1', scale_iwe_by_dt=True, mask_image_border=True, polarity_aware_batching=True,
        interpolation_scheme='mean', smooth_type='on_flow_to_tref'))
    td = traj.to(dev).requires_grad_(True)
    lut, _ = ops.KnnLutFn.apply(td, L._cfg)
    g = torch.Generator().manual_seed(6)
    wgt = torch.randn(lut.shape, generator=g).to(dev)
    (lut * wgt).sum().backward()
    grid, _, _ = O.lut_grid_points((H, W), SP)
    q = grid.to(dev)
    tb = traj.to(dev).requires_grad_(True)
    tot = 0.
    for t in (0, NB - 1):
        pts = tb[0, 1 + t].detach()
        idx = []
        for c in range(0, q.shape[0], 2048):
            d = (q[c:c + 2048, None, :] - pts[None, :, :]).abs().sum(-1)
            idx.append(torch.sort(d, dim=1, stable=True).indices[:, :K])
        idx = torch.cat(idx)
        ref = (tb[0, 0] - tb[0, 1 + t])[idx].mean(1)
        got = lut[0, t].reshape(-1, 2)
        assert torch.allclose(got, ref.detach(), atol=1e-4, rtol=1e-5), (family, t, float((got - ref).abs().max()))
        tot = tot + (ref * wgt[0, t].reshape(-1, 2)).sum()
    tot.backward()
    for t in (0, NB - 1):
        a, r = td.grad[0, 1 + t], tb.grad[0, 1 + t]
        assert float((a - r).norm()) <= 1e-5 * float(r.norm()) + 1e-7, (family, t, float((a - r).norm() / r.norm()))


@pytest.mark.parametrize('family', ['translate40', 'diverge-30'])
def test_fused_backward_with_far_queries_matches_the_staged_path(family):
    """mpc_focus_fwd / mpc_focus_bwd (the fused calls: KNN forward with the event count riding in it, tile reaches from the event
    backward's kernel, second launch, far backward) against the same loss put together from the stage entry points (KnnLutFn,
    EventFocusFn, LutSmoothFn) on inputs with emptied bands: loss equal, trajectory gradient to rounding."""
    from motionpriorcmax_amd import ops
    from motionpriorcmax_amd.utils import synth
    dev = _dev()
    B, M = 2, 80000
    traj, times = synth.synth_trajectories(B, 3, NB, (H, W), PATCH, family, seed=13)
    ev, num_pos = synth.synth_events_ragged(B, M, (H, W), NB, seed=14)
    L = _loss()
    evd, tmd = ev.to(dev), times.to(dev)
    ta = traj.to(dev).requires_grad_(True)
    loss, _, _ = L.calc(ta, tmd, {'events': evd, 'num_pos_events': num_pos})
    loss.backward()
    tb = traj.to(dev).requires_grad_(True)
    lut, _ = ops.KnnLutFn.apply(tb, L._cfg)
    focus, _, _ = ops.EventFocusFn.apply(lut, evd, tmd[:1], L._cfg, num_pos)
    b, nb, hq, wq, T, d = lut.shape
    field = lut.permute(0, 1, 4, 5, 2, 3).reshape(-1, d, hq, wq).permute(0, 2, 3, 1).contiguous()
    smooth = ops.LutSmoothFn.apply(field, L._cfg, 0.003)
    ref = focus + smooth
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * abs(float(ref)), (float(loss), float(ref))
    a, r = ta.grad, tb.grad
    assert torch.isfinite(a).all()
    assert float((a - r).norm()) <= 2e-5 * float(r.norm()) + 1e-9, float((a - r).norm() / r.norm())


@pytest.mark.parametrize('family,B,forward', [('translate10', 1, True), ('white', 1, True), ('unet', 1, False), ('translate40', 2, False)])
def test_tail_launch_paths_small_and_large_marked_counts(family, B, forward):
    """The tail launch has two ways to serve the queries the main launch marks: with at most 1 024 of them in the whole launch its
    one-wavefront search takes them from the MARKED list and no far pass runs (every B = 1 step of a lattice-like input); otherwise the
    far pass of its strip workgroups, and what that cannot finish goes on the LATE list.  Both against the brute-force search, every
    cell of three bins, and the counters say which path ran (mpc_knn_tail_counters_offset)."""
    import bench
    from motionpriorcmax_amd import ops, _lib as C
    from motionpriorcmax_amd.utils import synth
    from oracle import focus_oracle as O
    dev = _dev()
    if family == 'white':
        _, _, traj, _ = bench.synth_inputs(dict(bench.WORKLOADS['C3'], B=B), seed=5)
    else:
        traj, _ = synth.synth_trajectories(B, 3, NB, (H, W), PATCH, family, seed=31)
    L = _loss()
    shape = ops.make_shape(L._cfg, B, 0, 0, traj.shape[2])
    ws = ops.alloc_workspace(shape, dev)
    td = traj.to(dev)
    lut = ops.knn_lut_fwd(L._cfg, shape, td, ws)[0]
    torch.cuda.synchronize()
    off = C.lib().mpc_knn_tail_counters_offset(ctypes.byref(shape))
    marked, late, done = (int(ws[off + o:off + o + 4].view(torch.int32).item()) for o in (0, 128, 256))
    assert (marked <= 1024) == forward, (family, B, marked)
    if not forward:
        assert done > 0, (family, marked, late, done)          # strip workgroups of the tail launch had work and counted themselves
    grid, _, _ = O.lut_grid_points((H, W), SP)
    q = grid.to(dev)
    for b, t in ((0, 0), (B - 1, NB - 1), (0, NB // 2)):
        idx = _brute(td[b], t, q)
        ref = (td[b, 0] - td[b, 1 + t])[idx].mean(1)
        got = lut.reshape(B, NB, -1, 2)[b, t]
        assert torch.allclose(got, ref, atol=1e-4, rtol=1e-5), (family, b, t, float((got - ref).abs().max()))
