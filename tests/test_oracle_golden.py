"""The CPU oracle (oracle/focus_oracle.py) against golden vectors produced by the unmodified
reference (oracle/gen_golden.py).  This is what pins the oracle; no GPU needed."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_golden
from oracle import focus_oracle as O


def _run(g):
    cfg = g['cfg']
    L = O.FocusLossOracle(**cfg)
    T = cfg['num_tref']
    ev = torch.from_numpy(g['events'])
    times = torch.from_numpy(g['times'])
    cg = torch.from_numpy(g['coeff_grid']).requires_grad_(True)
    mask = O.tile_mask(cfg['image_shape'], int(g['patch']))
    traj = O.trajectories_at(cg, times, mask, int(g['num_basis']), str(g['basis_type']))
    traj.retain_grad()
    want_next = cfg['smooth_weight'] > 0 and cfg['smooth_type'] == 'on_flow_to_next'
    lut, nxt, idx = O.interpolate_flow(traj[:, :T], traj[:, T:], cfg['image_shape'],
                                       cfg['lut_superpixel_size'], cfg['num_knn'], cfg['dist_norm'],
                                       cfg['interpolation_scheme'], want_next, return_idx=True)
    lut.retain_grad()
    focus, iwes, raw = L.event_path(ev, lut, times[:T], int(g['num_pos']))
    smooth = L.smooth_loss(lut, nxt)
    loss = focus + smooth
    loss.backward()
    return dict(traj=traj, lut=lut, nxt=nxt, idx=idx, focus=focus, smooth=smooth, loss=loss,
                iwes=iwes, raw=raw, cg=cg, L=L, ev=ev, times=times)


@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_oracle_matches_reference(name):
    g = load_golden(name)
    r = _run(g)
    np.testing.assert_allclose(r['traj'].detach().numpy(), g['trajectories'], rtol=0, atol=1e-5)
    np.testing.assert_allclose(r['lut'].detach().numpy(), g['flow_lut'], rtol=0, atol=1e-5)
    if 'flow_next' in g:
        np.testing.assert_allclose(r['nxt'].detach().numpy(), g['flow_next'], rtol=0, atol=1e-5)
    if 'ind_k_sorted' in g:
        got = np.sort(r['idx'].numpy(), -1)
        assert (got == g['ind_k_sorted']).all()
    iw = r['iwes'].detach().numpy().reshape(g['iwes'].shape)
    np.testing.assert_allclose(iw, g['iwes'], rtol=0, atol=1e-5 * max(1.0, np.abs(g['iwes']).max()))
    assert abs(r['focus'].item() - g['focus_loss']) <= 1e-6 * abs(g['focus_loss'])
    assert abs(float(r['smooth']) - g['smooth_loss']) <= 1e-6 * abs(g['smooth_loss']) + 1e-12
    assert abs(r['loss'].item() - g['loss']) <= 1e-6 * abs(g['loss'])
    # gradients: identical inputs, same op order -> tight
    gl = r['lut'].grad.numpy()
    denom = np.linalg.norm(g['grad_flow_lut']) + 1e-30
    assert np.linalg.norm(gl - g['grad_flow_lut']) / denom < 1e-5
    gt = r['traj'].grad.numpy()
    denom = np.linalg.norm(g['grad_trajectories']) + 1e-30
    assert np.linalg.norm(gt - g['grad_trajectories']) / denom < 1e-5
    assert abs(r['cg'].grad.abs().sum().item() - g['grad_coeff_grid_abs_sum']) <= \
        1e-4 * g['grad_coeff_grid_abs_sum']
    # the whole coefficient-grid gradient (trajectory_net.py:101-119,142-161): non-zero at the tile centres only
    from oracle import focus_oracle as O
    mask = O.tile_mask(tuple(g['cfg']['image_shape']), int(g['patch']))
    gc = r['cg'].grad.numpy()
    want = g['grad_coeff_grid_at_tiles']
    assert np.linalg.norm(gc[..., mask.numpy()] - want) / (np.linalg.norm(want) + 1e-30) < 1e-5
    assert np.abs(gc[..., ~mask.numpy()]).max() == 0.0 == float(g['grad_coeff_grid_off_tiles_abs_max'])


@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_oracle_calc_entry_point(name):
    g = load_golden(name)
    cfg = g['cfg']
    L = O.FocusLossOracle(**cfg)
    batch = {'events': torch.from_numpy(g['events'])}
    if cfg['polarity_aware_batching']:
        batch['num_pos_events'] = int(g['num_pos'])
    loss, log, misc = L.calc(torch.from_numpy(g['trajectories']), torch.from_numpy(g['times']), batch)
    assert abs(loss.item() - g['loss']) <= 1e-6 * abs(g['loss'])
    assert misc['iwes'].shape == g['iwes'].shape
    assert set(log) == {'focus_loss', 'smoothness_loss'}


def test_variance_objective_config1():
    g = load_golden('g2_config1')
    v = O.contrast_value(torch.from_numpy(g['iwes']).reshape(-1, 2, 128, 128), 'variance')
    assert abs((1 / v).item() - g['variance_focus_loss']) <= 1e-6 * g['variance_focus_loss']


def test_imager_raw_events_path():
    """logging.py:76-79: create_iwe(raw events, weight=1.0, sigma=1)."""
    g = load_golden('g1_allflags')
    ev = torch.from_numpy(g['events'])[:1]
    raw = O.bilinear_vote(ev[..., :2], 1.0, (48, 64))
    img = O.gaussian_blur3(raw[:, None])[:, 0]
    np.testing.assert_allclose(img.numpy(), g['imager_iwe_raw_events'], atol=1e-5)


def test_primitives():
    g = load_golden('g7_primitives')
    img, fld = torch.from_numpy(g['img']), torch.from_numpy(g['field'])
    assert abs((1 / O.contrast_value(img, 'gradient_magnitude', 'l1')).item() - g['gm_l1']) < 1e-6 * g['gm_l1']
    assert abs((1 / O.contrast_value(img, 'gradient_magnitude', 'l2')).item() - g['gm_l2']) < 1e-6 * g['gm_l2']
    assert abs((1 / O.contrast_value(img, 'variance')).item() - g['var']) < 1e-6 * g['var']
    assert abs((1 / O.contrast_value(img[:, 0], 'gradient_magnitude', 'l1')).item() - g['gm_l1_3dim']) < 1e-6 * g['gm_l1_3dim']
    assert abs(O.smoothness(fld).item() - g['smooth']) < 1e-6 * g['smooth']


def test_bezier_basis():
    g = load_golden('g6_bezier10')
    fl = O.bezier_flow(torch.from_numpy(g['params']), g['timestamps'], 10)
    np.testing.assert_allclose(fl.numpy(), g['flows'], atol=1e-5)
    np.testing.assert_allclose(fl[-1].numpy(), g['flow_t1'], atol=1e-5)


@pytest.mark.parametrize('norm', ['l1', 'l2'])
def test_hand_derived_contrast_gradient_matches_autograd(norm):
    """the closed-form d(1/val)/d(raw IWE) used by the HIP backward."""
    g = torch.Generator().manual_seed(0)
    raw = (torch.rand(2, 2, 17, 23, generator=g) * 3).requires_grad_(True)
    val = O.contrast_value(O.gaussian_blur3(raw), 'gradient_magnitude', norm)
    (1 / val).backward()
    v2, gr = O.contrast_grad_image(raw.detach(), norm)
    assert abs(v2.item() - val.item()) < 1e-6 * val.item()
    np.testing.assert_allclose(gr.numpy(), raw.grad.numpy(), atol=2e-7, rtol=1e-4)


def test_per_event_basis_definition_is_consistent_with_the_lut_path():
    """PARITY UNPINNED (no reference code for the per-event continuous-time basis warp): the oracle's definition reduces to the
    reference's pinned LUT event path where the two must agree -- zero coefficients = a zero LUT; a coefficient grid that is
    constant in space with a constant-in-time basis difference... i.e. events all at ONE timestamp and a spatially constant grid =
    the LUT holding that constant flow in every cell."""
    import torch
    from oracle import focus_oracle as O
    H, W, sp, nb, k = 64, 96, 4, 5, 3
    cfg = dict(image_shape=(H, W), num_tref=1, num_bins=nb, num_knn=4, smooth_weight=0.0, lut_superpixel_size=sp, focus_loss_norm='l1',
               dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True, polarity_aware_batching=True, interpolation_scheme='mean',
               smooth_type='on_flow_to_tref')
    L = O.FocusLossOracle(**cfg)
    ev, npos = O.synth_events(2, 3000, (H, W), nb, seed=1)
    batch = {'events': ev, 'num_pos_events': npos}
    tref = torch.tensor([0.41])
    l0, _, m0 = L.calc_per_event_basis(torch.zeros(2, 1, 2 * k, H, W), 0.41, batch, k)
    f0, iw0, _ = L.event_path(ev, torch.zeros(2, nb, H // sp, W // sp, 1, 2), tref, npos)
    assert abs(float(l0) - float(f0)) <= 1e-6 * abs(float(f0)) and torch.allclose(m0['iwes'].reshape(iw0.shape), iw0)
    # one timestamp, spatially constant coefficients: flow = sum_k c_k (tref^k - t^k) for every event = a constant LUT
    ev1 = ev.clone(); ev1[..., 2] = 0.8
    c = torch.tensor([1.5, -0.7, 0.3, -2.0, 0.9, 0.4])                      # (y: k, x: k)
    grid = c.view(1, 1, 2 * k, 1, 1).expand(2, 1, 2 * k, H, W).contiguous()
    kk = torch.arange(1, k + 1, dtype=torch.float32)
    phi = 0.41 ** kk - 0.8 ** kk
    flow = torch.stack(((c[:k] * phi).sum(), (c[k:] * phi).sum()))
    lut = flow.view(1, 1, 1, 1, 1, 2).expand(2, nb, H // sp, W // sp, 1, 2).contiguous()
    l1, _, m1 = L.calc_per_event_basis(grid, 0.41, {'events': ev1, 'num_pos_events': npos}, k)
    f1, iw1, _ = L.event_path(ev1, lut, tref, npos)
    assert abs(float(l1) - float(f1)) <= 1e-5 * abs(float(f1))
    assert (m1['iwes'].reshape(iw1.shape) - iw1).abs().max() <= 1e-4 * iw1.abs().max()
