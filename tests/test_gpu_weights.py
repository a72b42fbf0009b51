"""Event weights beyond [0, 1] on the LDS-tiled vote (Q33.30 accumulators): the reference accepts any weight --
`imager.create_iwe(weight=<tensor>)` (event_image_converter.py:45-74) and whatever column 5 of the events holds
(focus.py:202) -- so the fixed-point conversion must not saturate.  Checked against the oracle's bilinear_vote."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(shape, nb):
    return dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=8, smooth_weight=0.003, lut_superpixel_size=4,
                focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True,
                polarity_aware_batching=True, interpolation_scheme='mean', smooth_type='on_flow_to_tref')


@pytest.mark.parametrize('wmax', [1.0, 3.7, 5.0, 1000.0])
def test_create_iwe_with_a_weight_tensor(wmax):
    from motionpriorcmax_amd import LossFactory
    from oracle import focus_oracle as O
    shape = (96, 128)
    g = torch.Generator().manual_seed(11)
    n = 20000
    ev = torch.zeros(2, n, 4)
    ev[..., 0] = torch.rand(2, n, generator=g) * (shape[0] + 6) - 3          # some taps fall outside the image
    ev[..., 1] = torch.rand(2, n, generator=g) * (shape[1] + 6) - 3
    w = (torch.rand(2, n, generator=g) * 2 - 1) * wmax                       # both signs, |w| up to wmax
    w[:, :5] = torch.tensor([wmax, -wmax, 2.0, -2.5, 1.9999])
    L = LossFactory.get_loss_calculator('FOCUS', _cfg(shape, 5))
    dev = torch.device('cuda:0')
    for sigma in (0, 1):
        got = L.imager.create_iwe(ev.to(dev), method='bilinear_vote', sigma=sigma, weight=w.to(dev)).cpu()
        want = O.bilinear_vote(ev[..., :2], w, shape)
        if sigma > 0:
            want = O.gaussian_blur3(want[:, None])[:, 0]
        # fp32 scatter_add of ~10 taps of magnitude wmax per pixel vs exact fixed point: 1e-5 * the largest pixel
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=0, atol=1e-5 * max(1.0, want.abs().max().item()))


def test_loss_with_a_weighted_valid_column():
    from motionpriorcmax_amd import LossFactory
    from oracle import focus_oracle as O
    shape, B, M, nb = (96, 128), 2, 12000, 5
    cfg = _cfg(shape, nb)
    ev, num_pos = O.synth_events(B, M, shape, nb, seed=4, pad_frac=0.02)
    g = torch.Generator().manual_seed(9)
    ev[..., 5] = ev[..., 5] * (torch.rand(B, M, generator=g) * 3.7)            # weights in [0, 3.7], padding rows stay 0
    ev[0, 7, 5] = -2.5
    coeff = torch.randn(B, 1, 2, *shape, generator=g) * 3.0
    times = torch.cat((torch.tensor([0.3]), O.bin_mid_times(nb)))
    traj = O.trajectories_at(coeff, times, O.tile_mask(shape, 4), 1, 'polynomial')
    to = traj.clone().requires_grad_(True)
    lo, _, mo = O.FocusLossOracle(**cfg).calc(to, times, {'events': ev, 'num_pos_events': num_pos})
    lo.backward()
    dev = torch.device('cuda:0')
    tg = traj.to(dev).requires_grad_(True)
    lg, _, mg = LossFactory.get_loss_calculator('FOCUS', cfg).calc(tg, times.to(dev), {'events': ev.to(dev), 'num_pos_events': num_pos})
    lg.backward()
    assert abs(lg.item() - lo.item()) <= 1e-5 * abs(lo.item())
    np.testing.assert_allclose(mg['iwes'].cpu().numpy(), mo['iwes'].numpy(), rtol=0, atol=1e-5 * mo['iwes'].abs().max().item())
    gn = (tg.grad.cpu() - to.grad).norm() / to.grad.norm()
    assert gn < 1e-2, gn
