"""UNPINNED extension: the per-event continuous-time basis warp (`FocusLoss.calc_per_event_basis`; BASELINE.json's north_star
names it, the reference has no such path: focus.py:182-195 gathers a binned KNN look-up table).  The HIP path (LDS-tiled vote of
pre-warped rows, contrast kernels, mpc_event_pos_grad + torch for the chain to the coefficient grid) against the DEFINITION
written from the reference's building blocks in oracle/focus_oracle.py (`FocusLossOracle.calc_per_event_basis`)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda', 0)


def _cfg(shape, nb, **over):
    cfg = dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=8, smooth_weight=0.003, lut_superpixel_size=4,
               focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True, polarity_aware_batching=True,
               interpolation_scheme='mean', smooth_type='on_flow_to_tref')
    cfg.update(over)
    return cfg


@pytest.mark.parametrize('over,basis,k', [
    ({}, 'polynomial', 3), ({'polarity_aware_batching': False, 'scale_iwe_by_dt': False}, 'polynomial', 1),
    ({'focus_loss_norm': 'l2', 'smooth_weight': 0.0}, 'dct', 2), ({'mask_image_border': False}, 'polynomial', 5),
    ({'smooth_weight': 0.0}, 'polynomial', 9),        # more orders than the kernels keep in registers: phi from torch, strided reads
])
@pytest.mark.parametrize('fused', [True, False, 'ordered'])
def test_per_event_basis_against_its_definition(over, basis, k, fused):
    from motionpriorcmax_amd import LossFactory
    from oracle import focus_oracle as O
    dev = _dev()
    shape, B, M, nb = (96, 128), 2, 12000, 5
    cfg = _cfg(shape, nb, **over)
    ev, num_pos = O.synth_events(B, M, shape, nb, seed=11, pad_frac=0.1)
    g = torch.Generator().manual_seed(3)
    coeff = torch.randn(B, 1, 2 * k, *shape, generator=g) * 4.0          # some events leave the image
    batch = {'events': ev, 'num_pos_events': num_pos}
    co = coeff.clone().requires_grad_(True)
    lo, _, mo = O.FocusLossOracle(**cfg).calc_per_event_basis(co, 0.41, batch, k, basis)
    lo.backward()
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    cg = coeff.to(dev).requires_grad_(True)
    bg = {'events': ev.to(dev), 'num_pos_events': num_pos}
    if fused == 'ordered':
        bg = L.order_events(bg)                  # rows permuted inside the polarity blocks + offsets: same loss, LDS backward
    lg, log, mg = L.calc_per_event_basis(cg, 0.41, bg, k, basis, fused=bool(fused))
    lg.backward()
    assert abs(lg.item() - lo.item()) <= 1e-5 * abs(lo.item()), (lg.item(), lo.item())
    iw = mo['iwes'].reshape(mg['iwes'].shape)
    assert (mg['iwes'].cpu() - iw).abs().max() <= 1e-5 * iw.abs().max()
    # gradient: relative L2 (an event next to a pixel whose Sobel response is zero up to rounding may take the other sign())
    go, gg = co.grad, cg.grad.cpu()
    assert torch.isfinite(gg).all() and go.abs().max() > 0
    assert (gg - go).norm() / go.norm() < 2e-3, float((gg - go).norm() / go.norm())
    # only the tile centres receive gradient (trajectories.py:3-13: the coefficients are sampled there)
    from motionpriorcmax_amd.utils import get_optical_flow_tile_mask
    assert float(gg[..., ~get_optical_flow_tile_mask(shape, 4)].abs().max()) == 0.0


def test_per_event_basis_is_reproducible_and_zero_coefficients_are_the_identity_warp():
    from motionpriorcmax_amd import LossFactory, ops
    from oracle import focus_oracle as O
    dev = _dev()
    shape, B, M, nb, k = (96, 128), 2, 9000, 5, 3
    cfg = _cfg(shape, nb, smooth_weight=0.0)
    ev, num_pos = O.synth_events(B, M, shape, nb, seed=5)
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
    z = torch.zeros(B, 1, 2 * k, *shape, device=dev)
    l0, _, m0 = L.calc_per_event_basis(z, 0.3, batch, k)
    lut = torch.zeros(B, nb, shape[0] // 4, shape[1] // 4, 1, 2, device=dev)
    f, iw, _ = ops.EventFocusFn.apply(lut, batch['events'], torch.tensor([0.3], device=dev), L._cfg, num_pos)
    assert torch.equal(l0, f) and torch.equal(m0['iwes'].reshape(iw.shape), iw)
    g = torch.Generator().manual_seed(1)
    c = (torch.randn(B, 1, 2 * k, *shape, generator=g) * 3).to(dev)
    outs = []
    for _ in range(2):
        cg = c.clone().requires_grad_(True)
        l, _, _ = L.calc_per_event_basis(cg, 0.3, batch, k)
        l.backward()
        outs.append((l.detach().clone(), cg.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])          # forward: integer accumulation, bitwise reproducible
    assert torch.isfinite(outs[0][1]).all()
    # with the bucket-ordered layout the backward is bitwise reproducible too (LDS fixed point, no global atomics)
    ob = L.order_events(batch)
    outs = []
    for _ in range(2):
        cg = c.clone().requires_grad_(True)
        l, _, _ = L.calc_per_event_basis(cg, 0.3, ob, k)
        l.backward()
        outs.append((l.detach().clone(), cg.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # ... and the forward does not care about the row order inside a polarity block: same loss, bit for bit
    cg = c.clone().requires_grad_(True)
    lp, _, _ = L.calc_per_event_basis(cg, 0.3, batch, k)
    assert torch.equal(lp.detach(), outs[0][0])
    for kk, basis in ((9, 'polynomial'), (8, 'dct')):
        c9 = (torch.randn(B, 1, 2 * kk, *shape, generator=g) * 0.5).to(dev)
        la, _, _ = L.calc_per_event_basis(c9, 0.3, batch, kk, basis)
        lb, _, _ = L.calc_per_event_basis(c9, 0.3, ob, kk, basis)
        assert torch.equal(la, lb), (kk, basis, float(la), float(lb))
