"""UNPINNED extension: FocusLoss(pyramid_levels=L) -- BASELINE.json's configs[2] names a multi-scale IWE pyramid, the reference
has none, so there is nothing to pin it to.  What is checked is that the HIP path computes the
DEFINITION it documents (ops.PyramidFocusFn: 2x2 averages of the raw IWE, the reference's blur + gradient-magnitude objective on
every level, focus = sum over the levels) -- against the same definition written with the CPU oracle's functions -- and that
pyramid_levels=1 is exactly the plain loss."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _case(norm, smooth_type):
    from oracle import focus_oracle as O
    shape, B, M, nb, K = (96, 128), 2, 12000, 5, 8
    cfg = dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=K, smooth_weight=0.01, lut_superpixel_size=4,
               focus_loss_norm=norm, dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True, polarity_aware_batching=True,
               interpolation_scheme='mean', smooth_type=smooth_type)
    ev, num_pos = O.synth_events(B, M, shape, nb, seed=2, pad_frac=0.02)
    g = torch.Generator().manual_seed(3)
    coeff = torch.randn(B, 1, 2, *shape, generator=g) * 2.0
    times = torch.cat((torch.tensor([0.41]), O.bin_mid_times(nb)))
    traj = O.trajectories_at(coeff, times, O.tile_mask(shape, 4), 1, 'polynomial').detach()
    return cfg, ev, num_pos, traj, times


def _oracle_pyramid(cfg, traj, times, ev, num_pos, levels):
    from oracle import focus_oracle as O
    Lo = O.FocusLossOracle(**cfg)
    want_next = cfg['smooth_type'] == 'on_flow_to_next'
    lut, nxt = O.interpolate_flow(traj[:, :1], traj[:, 1:], cfg['image_shape'], 4, cfg['num_knn'], 'l2', 'mean', want_next)
    warped = O.warp_events(ev, lut, 4)
    _, raw = O.make_iwes(ev, warped, times[:1], cfg['image_shape'], True, True, True, num_pos)
    focus = 0.0
    cur = raw
    for lv in range(levels):
        focus = focus + 1 / O.contrast_value(O.gaussian_blur3(cur), 'gradient_magnitude', cfg['focus_loss_norm'])
        cur = F.avg_pool2d(cur, 2)
    return focus + Lo.smooth_loss(lut, nxt), focus


@pytest.mark.parametrize('norm,smooth_type,levels', [('l2', 'on_flow_to_tref', 3), ('l1', 'on_flow_to_next', 2)])
def test_pyramid_matches_its_definition(norm, smooth_type, levels):
    from motionpriorcmax_amd import LossFactory
    dev = torch.device('cuda:0')
    cfg, ev, num_pos, traj, times = _case(norm, smooth_type)
    L = LossFactory.get_loss_calculator('FOCUS', dict(cfg, pyramid_levels=levels))
    tg = traj.to(dev).requires_grad_(True)
    loss, log, misc = L.calc(tg, times.to(dev), {'events': ev.to(dev), 'num_pos_events': num_pos})
    loss.backward()
    to = traj.clone().requires_grad_(True)
    lo, fo = _oracle_pyramid(cfg, to, times, ev, num_pos, levels)
    lo.backward()
    assert abs(loss.item() - lo.item()) <= 1e-5 * abs(lo.item()), (loss.item(), lo.item())
    assert abs(log['focus_loss'].item() - fo.item()) <= 1e-5 * abs(fo.item())
    rel = float((tg.grad.cpu() - to.grad).norm() / to.grad.norm())
    assert rel < (1e-4 if norm == 'l2' else 2e-2), rel          # ('l1': sign() of near-zero Sobel responses)
    assert misc['iwes'].shape == (2, 1, 2, 96, 128)


def test_one_level_is_the_plain_loss_and_bad_shapes_are_refused():
    from motionpriorcmax_amd import LossFactory
    dev = torch.device('cuda:0')
    cfg, ev, num_pos, traj, times = _case('l1', 'on_flow_to_tref')
    batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
    outs = []
    for kw in ({}, {'pyramid_levels': 1}):
        t = traj.to(dev).requires_grad_(True)
        l, _, m = LossFactory.get_loss_calculator('FOCUS', dict(cfg, **kw)).calc(t, times.to(dev), batch)
        l.backward()
        outs.append((l.detach(), t.grad, m['iwes']))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    with pytest.raises(ValueError):
        LossFactory.get_loss_calculator('FOCUS', dict(cfg, pyramid_levels=2, num_tref=2, scale_iwe_by_dt=False, polarity_aware_batching=False))
    L7 = LossFactory.get_loss_calculator('FOCUS', dict(cfg, pyramid_levels=7))        # 96 / 64 is not an integer
    with pytest.raises(ValueError):
        L7.calc(traj.to(dev), times.to(dev), batch)
