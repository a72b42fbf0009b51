"""Where the end-to-end gradient tolerances come from, counted.

The 'l1' objective differentiates |Sobel response| with sign(): a response that is zero up to the rounding of the IWE
sums (Q33.30 integer sums here, fp32 scatter_add in the reference) can come out with either sign, and every event that
votes within the 5x5 footprint of such a pixel then gets a different -- equally valid -- gradient.  That is why the
golden end-to-end tests allow 1e-3 (rel. L2).  This test makes the allowance accountable: every LUT cell whose gradient
differs from the oracle's is EXPLAINED by a near-zero response next to one of its events, the cells that are not
explained agree tightly, and the explained ones are few.  A backward bug in a rarely taken branch would show up as an
unexplained cell.  All other discontinuities of the path (floor(pos + 1e-6), the strict border comparisons) act on event
positions, which are bit-identical on both sides (same fp32 operations in the same order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cfg(shape, nb, norm):
    return dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=8, smooth_weight=0.0, lut_superpixel_size=4,
                focus_loss_norm=norm, dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True,
                polarity_aware_batching=True, interpolation_scheme='mean', smooth_type='on_flow_to_tref')


@pytest.mark.parametrize('shape,B,M,nb,seed', [((96, 128), 2, 20000, 5, 1), ((480, 640), 1, 200000, 15, 2)])
def test_every_gradient_mismatch_is_explained_by_a_near_zero_response(shape, B, M, nb, seed):
    from motionpriorcmax_amd import LossFactory, ops
    from oracle import focus_oracle as O
    H, W = shape
    sp = 4
    ev, num_pos = O.synth_events(B, M, shape, nb, seed=seed, pad_frac=0.02)
    g = torch.Generator().manual_seed(seed)
    lut = torch.randn(B, nb, -(-H // sp), -(-W // sp), 1, 2, generator=g) * 2.0
    t_ref = torch.tensor([0.41])
    cfg = _cfg(shape, nb, 'l1')
    orc = O.FocusLossOracle(**cfg)
    lo = lut.clone().requires_grad_(True)
    fo, iwo, rawo = orc.event_path(ev, lo, t_ref, num_pos)
    fo.backward()
    go = lo.grad[..., 0, :]                                             # [B, nb, hq, wq, 2]
    dev = torch.device('cuda:0')
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    lt = lut.to(dev).requires_grad_(True)
    f, blur, raw = ops.EventFocusFn.apply(lt, ev.to(dev), t_ref.to(dev), L._cfg, num_pos)
    f.backward()
    gg = lt.grad.cpu()[..., 0, :]

    # pixels whose Sobel response is zero up to rounding, inside the support of the image (oracle side)
    with torch.no_grad():
        blur_o = O.gaussian_blur3(rawo.detach())                        # [B, 2, H, W]
        dx, dy = O.sobel(blur_o)
        scale = max(dx.abs().max().item(), dy.abs().max().item())
        support = F.max_pool2d(blur_o.abs(), 5, stride=1, padding=2) > 0
        near_zero = ((dx.abs() < 1e-5 * scale) | (dy.abs() < 1e-5 * scale)) & support
        # ... and the adjoint-image pixels such a flip can change: Sobel^T then blur^T (+ the folded reflect ring)
        affected = F.max_pool2d(near_zero.float(), 9, stride=1, padding=4) > 0          # [B, 2, H, W]
        # events with a tap on an affected pixel -> their LUT cell is "explained"
        warped = O.warp_events(ev, lut, sp)[:, 0]                      # [B, M, 2]
        y0 = torch.floor(warped[..., 0] + 1e-6).long().clamp(0, H - 1)
        x0 = torch.floor(warped[..., 1] + 1e-6).long().clamp(0, W - 1)
        pol = (torch.arange(M)[None, :] >= num_pos).long().expand(B, M)
        bi = torch.arange(B)[:, None].expand(B, M)
        hit = affected[bi, pol, y0, x0] & (ev[..., 5] != 0)
        it = ev[..., 4].long()
        iy = torch.div(ev[..., 0], sp, rounding_mode='floor').long()
        ix = torch.div(ev[..., 1], sp, rounding_mode='floor').long()
        explained = torch.zeros(go.shape[:-1], dtype=torch.bool)
        explained[bi[hit], it[hit], iy[hit], ix[hit]] = True

    gmax = go.abs().max().item()
    err = (gg - go).abs().amax(-1)
    tight = err <= 2e-5 * gmax + 1e-4 * go.abs().amax(-1)
    unexplained = (~tight) & (~explained)
    n_cells = explained.numel()
    assert unexplained.sum().item() == 0, (f'{unexplained.sum().item()} of {n_cells} LUT cells differ from the oracle with no '
                                           f'near-zero response near their events; worst {err[unexplained].max().item() / gmax:.2e} of max')
    # the allowance is small, and it is used: the test must not pass because everything is "explained"
    frac_explained = explained.float().mean().item()
    frac_mismatch = (~tight).float().mean().item()
    assert frac_explained < 0.25, frac_explained
    assert frac_mismatch <= frac_explained and frac_mismatch < 0.01
    # with the sign() out of the way ('l2') the whole gradient agrees tightly
    cfg2 = _cfg(shape, nb, 'l2')
    lo2 = lut.clone().requires_grad_(True)
    f2o, _, _ = O.FocusLossOracle(**cfg2).event_path(ev, lo2, t_ref, num_pos)
    f2o.backward()
    lt2 = lut.to(dev).requires_grad_(True)
    f2, _, _ = ops.EventFocusFn.apply(lt2, ev.to(dev), t_ref.to(dev), LossFactory.get_loss_calculator('FOCUS', cfg2)._cfg, num_pos)
    f2.backward()
    g2o, g2g = lo2.grad, lt2.grad.cpu()
    assert ((g2g - g2o).abs() <= 2e-5 * g2o.abs().max() + 1e-4 * g2o.abs()).all()
    print(f'explained cells {100 * frac_explained:.2f} %, mismatching cells {100 * frac_mismatch:.3f} % of {n_cells}')


@pytest.mark.parametrize('shape,B,M,nb,K', [((192, 256), 2, 60000, 5, 8), ((480, 640), 1, 100000, 15, 32)])
def test_end_to_end_gradient_mismatches_are_explained(shape, B, M, nb, K):
    """The same accounting for d loss / d trajectories through the KNN LUT (dsec.yaml switches, smoothness on): a
    trajectory point may differ from the oracle only if an explained LUT cell lies within the reach of its neighbourhood.
    Second case: the DSEC sensor with K = 32 and 15 bins (one sample; the oracle's brute-force KNN takes a minute or two)."""
    from motionpriorcmax_amd import LossFactory, ops
    from oracle import focus_oracle as O
    sp = 4
    H, W = shape
    cfg = dict(_cfg(shape, nb, 'l1'), smooth_weight=0.003, num_knn=K)
    ev, num_pos = O.synth_events(B, M, shape, nb, seed=5, pad_frac=0.02)
    g = torch.Generator().manual_seed(5)
    coeff = torch.randn(B, 1, 2, H, W, generator=g) * 3.0
    times = torch.cat((torch.tensor([0.37]), O.bin_mid_times(nb)))
    traj = O.trajectories_at(coeff, times, O.tile_mask(shape, 4), 1, 'polynomial')
    to = traj.clone().requires_grad_(True)
    lo, _, _ = O.FocusLossOracle(**cfg).calc(to, times, {'events': ev, 'num_pos_events': num_pos})
    lo.backward()
    dev = torch.device('cuda:0')
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    tg = traj.to(dev).requires_grad_(True)
    lg, _, misc = L.calc(tg, times.to(dev), {'events': ev.to(dev), 'num_pos_events': num_pos})
    lg.backward()
    assert abs(lg.item() - lo.item()) <= 1e-5 * abs(lo.item())
    go, gg = to.grad, tg.grad.cpu()                                    # [B, 1 + nb, n, 2]
    with torch.no_grad():
        lut = ops.KnnLutFn.apply(traj.to(dev), L._cfg)[0].cpu()       # the LUT the events were warped with
        raw = misc['iwes'].cpu().reshape(B, 2, H, W)                   # blurred IWE (detached output of calc)
        dx, dy = O.sobel(raw)
        scale = max(dx.abs().max().item(), dy.abs().max().item())
        support = F.max_pool2d(raw.abs(), 5, stride=1, padding=2) > 0
        near_zero = ((dx.abs() < 1e-5 * scale) | (dy.abs() < 1e-5 * scale)) & support
        affected = F.max_pool2d(near_zero.float(), 9, stride=1, padding=4) > 0
        warped = O.warp_events(ev, lut, sp)[:, 0]
        y0 = torch.floor(warped[..., 0] + 1e-6).long().clamp(0, H - 1)
        x0 = torch.floor(warped[..., 1] + 1e-6).long().clamp(0, W - 1)
        pol = (torch.arange(M)[None, :] >= num_pos).long().expand(B, M)
        bi = torch.arange(B)[:, None].expand(B, M)
        hit = affected[bi, pol, y0, x0] & (ev[..., 5] != 0)
        it = ev[..., 4].long()
        iy = torch.div(ev[..., 0], sp, rounding_mode='floor').long()
        ix = torch.div(ev[..., 1], sp, rounding_mode='floor').long()
        hq, wq = H // sp, W // sp
        cell = torch.zeros(B, nb, hq, wq)
        cell[bi[hit], it[hit], iy[hit], ix[hit]] = 1.0
        # a LUT cell averages K points around it (~2 cells away at K = 8, ~3.3 at K = 32, one point per cell; 6 is generous)
        near = F.max_pool2d(cell, 13, stride=1, padding=6) > 0        # [B, nb, hq, wq]
        # trajectory point -> its cell at the bin's time (row 1 + t of the tensor)
        pos = traj[:, 1:]                                              # [B, nb, n, 2]
        cy = torch.div(pos[..., 0], sp, rounding_mode='floor').long().clamp(0, hq - 1)
        cx = torch.div(pos[..., 1], sp, rounding_mode='floor').long().clamp(0, wq - 1)
        bb = torch.arange(B)[:, None, None].expand_as(cy)
        tt = torch.arange(nb)[None, :, None].expand_as(cy)
        expl_bins = near[bb, tt, cy, cx]                               # [B, nb, n]
        explained = torch.cat((expl_bins.any(1, keepdim=True), expl_bins), 1)          # row 0 (t_ref) sums all bins
    gmax = go.abs().max().item()
    err = (gg - go).abs().amax(-1)
    tight = err <= 5e-5 * gmax + 2e-4 * go.abs().amax(-1)
    unexplained = (~tight) & (~explained)
    assert unexplained.sum().item() == 0, (f'{unexplained.sum().item()} trajectory points differ with no near-zero response in reach; '
                                           f'worst {err[unexplained].max().item() / gmax:.2e} of max')
    # (the excuse is generous here -- a third to a half of the points have SOME near-zero response in reach -- so the
    # guard is on the mismatches themselves: at most 1 % of the points may need it)
    assert (~tight).float().mean().item() < 0.01
    print(f'explained points {100 * explained.float().mean().item():.2f} %, mismatching {100 * (~tight).float().mean().item():.3f} %')
