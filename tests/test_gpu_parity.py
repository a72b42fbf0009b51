"""HIP path vs the CPU oracle and the reference's golden vectors (needs an MI355X).

Tolerances (fp32): IWE atol 1e-5*max, LUT atol 1e-5, loss rel 1e-5 (the
BASELINE target; stage-level 2e-6), gradients by relative L2 (they are discontinuous at pixel
cell boundaries and at sign(), so a max-abs bound is not meaningful)."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_golden

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def _rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30)


def _loss_obj(cfg, **kw):
    from motionpriorcmax_amd import LossFactory
    c = dict(cfg)
    c.update(kw)
    return LossFactory.get_loss_calculator('FOCUS', c)


def test_native_library_is_loaded():
    from motionpriorcmax_amd import _lib
    assert _lib.lib().mpc_version() == 107
    import os
    with open('/proc/self/maps') as f:                       # (MPC_AB_LIB: the bounds-checked build of tools/bounds_run.sh sits beside the product one)
        assert os.path.basename(_lib.LIB_PATH) in f.read()
    assert os.path.basename(_lib.LIB_PATH).startswith('libmpcmax')


@pytest.mark.parametrize('atomic', [False, True])
@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_golden_full_calc(name, atomic):
    g = load_golden(name)
    cfg = g['cfg']
    L = _loss_obj(cfg, debug_atomic_path=atomic)
    dev = _dev()
    traj = torch.from_numpy(g['trajectories']).to(dev).requires_grad_(True)
    times = torch.from_numpy(g['times']).to(dev)
    batch = {'events': torch.from_numpy(g['events']).to(dev)}
    if cfg['polarity_aware_batching']:
        batch['num_pos_events'] = int(g['num_pos'])
    loss, log, misc = L.calc(traj, times, batch)
    assert loss.dim() == 0 and loss.requires_grad
    assert abs(loss.item() - g['loss']) <= 1e-5 * abs(g['loss'])
    assert abs(log['focus_loss'].item() - g['focus_loss']) <= 1e-5 * abs(g['focus_loss'])
    assert abs(log['smoothness_loss'].item() - g['smooth_loss']) <= 1e-5 * abs(g['smooth_loss']) + 1e-12
    iw = misc['iwes'].cpu().numpy()
    assert iw.shape == g['iwes'].shape
    np.testing.assert_allclose(iw, g['iwes'], rtol=0, atol=1e-5 * max(1.0, np.abs(g['iwes']).max()))
    loss.backward()
    assert _rel_l2(traj.grad.cpu().numpy(), g['grad_trajectories']) < 1e-3


@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_golden_stages(name):
    """Stage-wise with bit-identical stage inputs: LUT from trajectories; event path and
    smoothness from the golden LUT."""
    from motionpriorcmax_amd import ops
    g = load_golden(name)
    cfg = g['cfg']
    L = _loss_obj(cfg)
    pc = L._cfg
    dev = _dev()
    T = cfg['num_tref']
    # A5
    traj = torch.from_numpy(g['trajectories']).to(dev)
    lut, nxt = ops.KnnLutFn.apply(traj, pc)
    np.testing.assert_allclose(lut.cpu().numpy(), g['flow_lut'], rtol=0, atol=1e-5)
    if 'flow_next' in g:
        np.testing.assert_allclose(nxt.cpu().numpy(), g['flow_next'], rtol=0, atol=1e-5)
    if 'ind_k_sorted' in g:
        idx = ops.knn_indices(pc, traj).cpu().numpy()
        assert (np.sort(idx, -1) == g['ind_k_sorted']).all()
    # A6-A9 + A10 from the golden LUT
    glut = torch.from_numpy(g['flow_lut']).to(dev).requires_grad_(True)
    ev = torch.from_numpy(g['events']).to(dev)
    times = torch.from_numpy(g['times']).to(dev)
    focus, blur, raw = ops.EventFocusFn.apply(glut, ev, times[:T], pc, int(g['num_pos']))
    assert abs(focus.item() - g['focus_loss']) <= 2e-6 * abs(g['focus_loss'])
    np.testing.assert_allclose(blur.cpu().numpy().reshape(g['iwes'].shape), g['iwes'], rtol=0,
                               atol=1e-5 * max(1.0, np.abs(g['iwes']).max()))
    total = focus
    if cfg['smooth_weight'] > 0:
        if cfg['smooth_type'] == 'on_flow_to_tref':
            field = glut.reshape(-1, *pc.lut_grid, 2 * T)
        else:
            gnext = torch.from_numpy(g['flow_next']).to(dev).requires_grad_(True)
            field = gnext.reshape(-1, *pc.lut_grid, 2)
        smooth = ops.LutSmoothFn.apply(field, pc, cfg['smooth_weight'])
        assert abs(smooth.item() - g['smooth_loss']) <= 2e-6 * abs(g['smooth_loss'])
        total = focus + smooth
    total.backward()
    assert _rel_l2(glut.grad.cpu().numpy(), g['grad_flow_lut']) < 1e-4
    if 'grad_flow_next' in g:
        assert _rel_l2(gnext.grad.cpu().numpy(), g['grad_flow_next']) < 1e-5


def test_imager_raw_events_golden():
    g = load_golden('g1_allflags')
    L = _loss_obj(g['cfg'])
    ev = torch.from_numpy(g['events'])[:1].to(_dev())
    img = L.imager.create_iwe(ev, method='bilinear_vote', sigma=1)
    np.testing.assert_allclose(img.cpu().numpy(), g['imager_iwe_raw_events'], atol=1e-5)


def test_variance_objective_config1():
    g = load_golden('g2_config1')
    L = _loss_obj(g['cfg'], loss_type='variance')
    dev = _dev()
    traj = torch.from_numpy(g['trajectories']).to(dev)
    batch = {'events': torch.from_numpy(g['events']).to(dev), 'num_pos_events': int(g['num_pos'])}
    _, log, _ = L.calc(traj, torch.from_numpy(g['times']).to(dev), batch)
    assert abs(log['focus_loss'].item() - g['variance_focus_loss']) <= 1e-5 * g['variance_focus_loss']


def _oracle_case(shape, B, M, nb, K, seed, **cfgkw):
    from oracle import focus_oracle as O
    cfg = dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=K, smooth_weight=0.003,
               lut_superpixel_size=4, focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True,
               mask_image_border=True, polarity_aware_batching=True, interpolation_scheme='mean',
               smooth_type='on_flow_to_tref')
    cfg.update(cfgkw)
    ev, num_pos = O.synth_events(B, M, shape, nb, seed=seed, pad_frac=0.02)
    g = torch.Generator().manual_seed(seed + 100)
    coeff = torch.randn(B, 1, 2, *shape, generator=g) * 3.0
    T = cfg['num_tref']
    t_ref = torch.tensor([0.41]) if T == 1 else torch.linspace(0, 1, T)
    times = torch.cat((t_ref, O.bin_mid_times(nb)))
    traj = O.trajectories_at(coeff, times, O.tile_mask(shape, 4), 1, 'polynomial')
    return cfg, ev, num_pos, traj, times


@pytest.mark.parametrize('kw', [
    dict(),
    dict(focus_loss_norm='l2', interpolation_scheme='iwd', dist_norm='l1'),
    dict(smooth_type='on_flow_to_next', smooth_weight=0.06),
    dict(loss_type='variance'),
    dict(num_tref=2, scale_iwe_by_dt=False, polarity_aware_batching=False),
], ids=['dsec', 'l2-iwd-l1', 'next', 'variance', 'tref2'])
def test_vs_oracle_seeded(kw):
    """96x128, B=3, 20k events: full calc forward + backward against the CPU oracle."""
    from oracle import focus_oracle as O
    cfg, ev, num_pos, traj, times = _oracle_case((96, 128), 3, 20000, 7, 8, seed=11, **kw)
    Lo = O.FocusLossOracle(**cfg)
    tr_o = traj.clone().requires_grad_(True)
    batch = {'events': ev, 'num_pos_events': num_pos}
    lo, logo, misco = Lo.calc(tr_o, times, batch)
    lo.backward()
    dev = _dev()
    L = _loss_obj(cfg)
    tr_g = traj.to(dev).requires_grad_(True)
    lg, logg, miscg = L.calc(tr_g, times.to(dev), {'events': ev.to(dev), 'num_pos_events': num_pos})
    lg.backward()
    assert abs(lg.item() - lo.item()) <= 1e-5 * abs(lo.item())
    assert abs(logg['smoothness_loss'].item() - logo['smoothness_loss'].item()) <= 1e-5 * abs(logo['smoothness_loss'].item()) + 1e-12
    io = misco['iwes'].numpy()
    np.testing.assert_allclose(miscg['iwes'].cpu().numpy(), io, rtol=0, atol=1e-5 * max(1.0, np.abs(io).max()))
    assert _rel_l2(tr_g.grad.cpu().numpy(), tr_o.grad.numpy()) < 1e-2


@pytest.mark.parametrize('as_samples', [True, False])
def test_num_tref_3_both_paths_vs_oracle(as_samples):
    """num_tref = 3 through `calc`: the T reference times as T samples of the num_tref == 1 kernels (the default since round 6) and the
    general kernels (trefs_as_samples=False), each against the CPU oracle; which kernels ran is asserted."""
    from motionpriorcmax_amd import ops
    from oracle import focus_oracle as O
    cfg, ev, num_pos, traj, times = _oracle_case((96, 128), 2, 20000, 7, 8, seed=13, num_tref=3, scale_iwe_by_dt=False,
                                                 polarity_aware_batching=False, smooth_weight=0.05)
    Lo = O.FocusLossOracle(**cfg)
    tr_o = traj.clone().requires_grad_(True)
    lo, logo, misco = Lo.calc(tr_o, times, {'events': ev})
    lo.backward()
    dev = _dev()
    L = _loss_obj(cfg, trefs_as_samples=as_samples)
    tr_g = traj.to(dev).requires_grad_(True)
    with ops.KernelTimer() as kt:
        lg, logg, miscg = L.calc(tr_g, times.to(dev), {'events': ev.to(dev)})
        lg.backward()
    ran = set(kt.summary())
    assert ('k_knn_strip' in ran and 'k_knn_bwd_tile' in ran and 'k_ev_bin' in ran) == as_samples, ran
    assert abs(lg.item() - lo.item()) <= 1e-5 * abs(lo.item())
    assert abs(logg['smoothness_loss'].item() - logo['smoothness_loss'].item()) <= 1e-5 * abs(logo['smoothness_loss'].item()) + 1e-12
    io = misco['iwes'].numpy()
    assert miscg['iwes'].shape == misco['iwes'].shape
    np.testing.assert_allclose(miscg['iwes'].cpu().numpy(), io, rtol=0, atol=1e-5 * max(1.0, np.abs(io).max()))
    assert _rel_l2(tr_g.grad.cpu().numpy(), tr_o.grad.numpy()) < 1e-2


def test_tiled_and_atomic_paths_agree():
    from motionpriorcmax_amd import ops
    from oracle import focus_oracle as O
    cfg, ev, num_pos, traj, times = _oracle_case((96, 128), 2, 30000, 5, 8, seed=5)
    dev = _dev()
    lut, _ = O.interpolate_flow(traj[:, :1], traj[:, 1:], (96, 128), 4, 8)
    outs = []
    for atomic in (False, True):
        L = _loss_obj(cfg, debug_atomic_path=atomic)
        lt = lut.to(dev).requires_grad_(True)
        f, blur, raw = ops.EventFocusFn.apply(lt, ev.to(dev), times[:1].to(dev), L._cfg, num_pos)
        f.backward()
        outs.append((f.item(), raw.cpu().numpy(), lt.grad.cpu().numpy()))
    assert abs(outs[0][0] - outs[1][0]) <= 2e-6 * abs(outs[1][0])
    np.testing.assert_allclose(outs[0][1], outs[1][1], atol=1e-5 * max(1.0, outs[1][1].max()))
    assert _rel_l2(outs[0][2], outs[1][2]) < 1e-5


def test_all_events_in_one_bucket():
    """All events inside one image strip and one time bin: one bucket of the LDS-tiled path receives everything (buckets
    hold whatever can reach them; there is no spill path), the result stays exact and bitwise reproducible."""
    from motionpriorcmax_amd import ops
    from oracle import focus_oracle as O
    dev = _dev()
    B, M, nb = 1, 40000, 15
    ev, num_pos = O.synth_events(B, M, (480, 640), nb, seed=2)
    ev[..., 0] = ev[..., 0] * (20.0 / 479.0) + 3.0          # rows 3..23: one 30-row strip, one LUT strip
    ev[..., 2] = 0.5 + 0.01 * ev[..., 2]
    ev[..., 4] = 7
    g = torch.Generator().manual_seed(4)
    lut = torch.randn(B, nb, 120, 160, 1, 2, generator=g) * 1.5
    cfg = dict(image_shape=(480, 640), num_tref=1, num_bins=nb, num_knn=32, smooth_weight=0.0,
               lut_superpixel_size=4, focus_loss_norm='l2', dist_norm='l2', scale_iwe_by_dt=True,
               mask_image_border=True, polarity_aware_batching=True, interpolation_scheme='mean',
               smooth_type='on_flow_to_tref')
    lo = lut.clone().requires_grad_(True)
    fo, iwo, rawo = O.FocusLossOracle(**cfg).event_path(ev, lo, torch.tensor([0.41]), num_pos)
    fo.backward()
    L = _loss_obj(cfg)
    lt = lut.to(dev).requires_grad_(True)
    f, blur, raw = ops.EventFocusFn.apply(lt, ev.to(dev), torch.tensor([0.41], device=dev), L._cfg, num_pos)
    f.backward()
    assert abs(f.item() - fo.item()) <= 2e-6 * abs(fo.item())
    np.testing.assert_allclose(raw.cpu().numpy(), rawo.detach().numpy(), atol=1e-5 * rawo.max().item())
    assert _rel_l2(lt.grad.cpu().numpy(), lo.grad.numpy()) < 1e-4
    lt2 = lut.to(dev).requires_grad_(True)
    f2, _, raw2 = ops.EventFocusFn.apply(lt2, ev.to(dev), torch.tensor([0.41], device=dev), L._cfg, num_pos)
    f2.backward()
    assert torch.equal(raw2, raw) and torch.equal(lt2.grad, lt.grad) and f2.item() == f.item()


@pytest.mark.parametrize('shape,sp,patch', [((50, 70), 4, 4), ((33, 47), 2, 3), ((64, 96), 8, 4), ((45, 63), 3, 3)])
def test_odd_image_sizes_vs_oracle(shape, sp, patch):
    """Image sizes that are not multiples of the tile / superpixel / strip sizes, n != Q."""
    from oracle import focus_oracle as O
    nb, K, B, M = 4, 5, 2, 6000
    cfg = dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=K, smooth_weight=0.01,
               lut_superpixel_size=sp, focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True,
               mask_image_border=True, polarity_aware_batching=True, interpolation_scheme='mean',
               smooth_type='on_flow_to_tref')
    ev, num_pos = O.synth_events(B, M, shape, nb, seed=17, pad_frac=0.05, num_pos=M // 3)
    g = torch.Generator().manual_seed(18)
    coeff = torch.randn(B, 1, 2, *shape, generator=g) * 4.0
    times = torch.cat((torch.tensor([0.3]), O.bin_mid_times(nb)))
    traj = O.trajectories_at(coeff, times, O.tile_mask(shape, patch), 1, 'polynomial')
    to = traj.clone().requires_grad_(True)
    lo, logo, misco = O.FocusLossOracle(**cfg).calc(to, times, {'events': ev, 'num_pos_events': num_pos})
    lo.backward()
    dev = _dev()
    tg = traj.to(dev).requires_grad_(True)
    lg, logg, miscg = _loss_obj(cfg).calc(tg, times.to(dev), {'events': ev.to(dev), 'num_pos_events': num_pos})
    lg.backward()
    assert abs(lg.item() - lo.item()) <= 1e-5 * abs(lo.item())
    io = misco['iwes'].numpy()
    np.testing.assert_allclose(miscg['iwes'].cpu().numpy(), io, rtol=0, atol=1e-5 * max(1.0, np.abs(io).max()))
    assert _rel_l2(tg.grad.cpu().numpy(), to.grad.numpy()) < 1e-2


def test_empty_and_all_padding_windows():
    """No events at all (M = 0) and windows made only of padding rows: the IWE is zero, the
    contrast value 0 and the focus loss +inf, exactly as in the reference's arithmetic (1 / 0)."""
    from oracle import focus_oracle as O
    shape, nb = (48, 64), 5
    cfg = dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=4, smooth_weight=0.003,
               lut_superpixel_size=4, focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True,
               mask_image_border=True, polarity_aware_batching=True, interpolation_scheme='mean',
               smooth_type='on_flow_to_tref')
    g = torch.Generator().manual_seed(3)
    coeff = torch.randn(2, 1, 2, *shape, generator=g)
    times = torch.cat((torch.tensor([0.5]), O.bin_mid_times(nb)))
    traj = O.trajectories_at(coeff, times, O.tile_mask(shape, 4), 1, 'polynomial')
    dev = _dev()
    L = _loss_obj(cfg)
    for ev, num_pos in ((torch.zeros(2, 0, 6), 0), (torch.zeros(2, 300, 6), 120)):
        loss, log, misc = L.calc(traj.to(dev), times.to(dev), {'events': ev.to(dev), 'num_pos_events': num_pos})
        assert torch.isinf(loss).item() and loss.item() > 0
        assert misc['iwes'].shape == (2, 1, 2, 48, 64) and float(misc['iwes'].abs().max()) == 0.0
        so = O.FocusLossOracle(**cfg).smooth_loss(O.interpolate_flow(traj[:, :1], traj[:, 1:], shape, 4, 4)[0], None)
        assert abs(log['smoothness_loss'].item() - so.item()) <= 1e-5 * so.item()


def test_single_sample_single_event_and_far_out_of_bounds_flow():
    """B*T = 1 with one valid event, and a LUT that throws every event far outside the image
    (all taps masked): finite gradients, zero image."""
    from motionpriorcmax_amd import ops
    from oracle import focus_oracle as O
    dev = _dev()
    shape, nb = (48, 64), 5
    cfg = dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=4, smooth_weight=0.0,
               lut_superpixel_size=4, focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=False,
               mask_image_border=False, polarity_aware_batching=False, interpolation_scheme='mean',
               smooth_type='on_flow_to_tref')
    L = _loss_obj(cfg)
    ev = torch.zeros(1, 8, 6)
    ev[0, 3] = torch.tensor([20.25, 30.75, 0.4, 1.0, 2.0, 1.0])
    lut = torch.zeros(1, nb, 12, 16, 1, 2)
    lut[0, 2, 5, 7, 0] = torch.tensor([1.5, -2.25])
    fo, iwo, rawo = O.FocusLossOracle(**cfg).event_path(ev, lut, torch.tensor([0.5]), -1)
    lt = lut.to(dev).requires_grad_(True)
    f, blur, raw = ops.EventFocusFn.apply(lt, ev.to(dev), torch.tensor([0.5], device=dev), L._cfg, -1)
    f.backward()
    assert abs(f.item() - fo.item()) <= 2e-6 * abs(fo.item())
    np.testing.assert_allclose(raw.cpu().numpy().reshape(rawo.shape), rawo.numpy(), atol=1e-6)
    assert torch.isfinite(lt.grad).all() and (lt.grad != 0).sum().item() <= 2
    far = torch.full((1, nb, 12, 16, 1, 2), 500.0, device=dev)
    ev2, _ = O.synth_events(1, 500, shape, nb, seed=1)
    _, _, raw2 = ops.EventFocusFn.apply(far, ev2.to(dev), torch.tensor([0.5], device=dev), L._cfg, -1)
    assert float(raw2.abs().max()) == 0.0


@pytest.mark.parametrize('env', [{'MPC_EV_EXACT_ABOVE_MB': '0'}, {'MPC_EV_STRIPS': '40'}])
def test_alternative_event_kernels_agree_with_goldens(env):
    """The backward event buckets sized by the counting pass (normally only above MPC_EV_EXACT_ABOVE_MB) and a forced count of
    image strips in the forward event kernels -- switches read once per process -- must pass the same parity tests as the
    default: the event tests of this file are re-run in a subprocess with the switch set."""
    import os
    import subprocess
    import sys
    if os.environ.get('MPC_ALT_KNN_CHILD'):
        pytest.skip('already inside the child run')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sel = ('golden_full_calc or golden_stages or vs_oracle_seeded or tiled_and_atomic or all_events_in_one_bucket or odd_image_sizes '
           'or empty_and_all_padding or single_sample_single_event or full_size_mass or full_size_event_path or bitwise_reproducible')
    r = subprocess.run([sys.executable, '-m', 'pytest', 'tests/test_gpu_parity.py', '-x', '-q', '-m', 'gpu', '-k', sel],
                       cwd=root, env=dict(os.environ, MPC_ALT_KNN_CHILD='1', **env), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0 and ' passed' in r.stdout, (r.stdout[-1500:], r.stderr[-500:])


def _knn_vs_bruteforce(traj, shape, sp, K, dist_norm='l2', scheme='mean'):
    from motionpriorcmax_amd import ops
    from oracle import focus_oracle as O
    cfg = dict(image_shape=shape, num_tref=1, num_bins=traj.shape[1] - 1, num_knn=K, smooth_weight=0.0,
               lut_superpixel_size=sp, focus_loss_norm='l1', dist_norm=dist_norm, scale_iwe_by_dt=True,
               mask_image_border=True, polarity_aware_batching=True, interpolation_scheme=scheme,
               smooth_type='on_flow_to_tref')
    L = _loss_obj(cfg)
    lut, _ = ops.KnnLutFn.apply(traj.to(_dev()), L._cfg)
    ref, _ = O.interpolate_flow(traj[:, :1], traj[:, 1:], shape, sp, K, dist_norm, scheme)
    return lut.cpu().numpy(), ref.numpy()


def test_knn_large_k_and_dense_points():
    """K = 64 (the search square has to grow) and 4 trajectories per LUT cell (patch 2, sp 4)."""
    from oracle import focus_oracle as O
    g = torch.Generator().manual_seed(31)
    times = torch.cat((torch.tensor([0.4]), O.bin_mid_times(3)))
    coeff = torch.randn(1, 1, 2, 64, 96, generator=g) * 5.0
    traj = O.trajectories_at(coeff, times, O.tile_mask((64, 96), 4), 1, 'polynomial')
    got, ref = _knn_vs_bruteforce(traj, (64, 96), 4, 64)
    np.testing.assert_allclose(got, ref, atol=2e-5)
    traj2 = O.trajectories_at(coeff, times, O.tile_mask((64, 96), 2), 1, 'polynomial')
    got, ref = _knn_vs_bruteforce(traj2, (64, 96), 4, 16, dist_norm='l1', scheme='iwd')
    np.testing.assert_allclose(got, ref, atol=2e-5)


def test_knn_l1_distance_at_dsec_density_on_the_strip_path():
    """dist_norm 'l1' at the DSEC density (480 x 640, one trajectory per 4 x 4 tile, K = 32): a square that holds K points in its diamond
    holds ~2 K candidate slots -- since round 6 the strip kernel's L1 instantiations hold 128 per query and serve the configuration (the
    general tile kernel before).  LUT against the brute-force oracle, gradient against autograd through it, and the launch that ran."""
    from motionpriorcmax_amd import ops
    from oracle import focus_oracle as O
    shape, sp, K = (480, 640), 4, 32
    g = torch.Generator().manual_seed(77)
    times = torch.cat((torch.tensor([0.41]), O.bin_mid_times(2)))
    coeff = torch.randn(1, 1, 2, shape[0], shape[1], generator=g) * 6.0
    traj = O.trajectories_at(coeff, times, O.tile_mask(shape, 4), 1, 'polynomial')
    cfg = dict(image_shape=shape, num_tref=1, num_bins=2, num_knn=K, smooth_weight=0.0, lut_superpixel_size=sp,
               focus_loss_norm='l1', dist_norm='l1', scale_iwe_by_dt=True, mask_image_border=True,
               polarity_aware_batching=True, interpolation_scheme='mean', smooth_type='on_flow_to_tref')
    L = _loss_obj(cfg)
    t = traj.to(_dev()).requires_grad_(True)
    with ops.KernelTimer() as kt:
        lut, _ = ops.KnnLutFn.apply(t, L._cfg)
    ran = set(kt.summary())
    assert 'k_knn_strip' in ran and 'k_knn_query' not in ran, ran
    tr = traj.clone().requires_grad_(True)
    ref, _ = O.interpolate_flow(tr[:, :1], tr[:, 1:], shape, sp, K, 'l1', 'mean')
    np.testing.assert_allclose(lut.detach().cpu().numpy(), ref.detach().numpy(), atol=2e-5)
    gl = torch.randn(lut.shape, generator=g)
    lut.backward(gl.to(_dev()))
    ref.backward(gl)
    assert _rel_l2(t.grad.cpu().numpy(), tr.grad.numpy()) < 1e-5


def test_knn_degenerate_point_sets():
    """Every trajectory collapsed onto a handful of spots (heavy distance ties, empty cells everywhere
    else: the whole-grid search path), and trajectories far outside the image."""
    from oracle import focus_oracle as O
    g = torch.Generator().manual_seed(32)
    n, nb = 192, 2
    base = O.tile_mask((48, 64), 4).nonzero().float()
    traj = base[None, None].repeat(1, 1 + nb, 1, 1).clone()
    # bin 0: three clusters with distinct flows; bin 1: everything 500 px outside the image
    centres = torch.tensor([[5.0, 7.0], [40.0, 50.0], [20.0, 30.0]])
    traj[0, 1] = centres[torch.arange(n) % 3] + torch.randn(n, 2, generator=g) * 1e-3
    traj[0, 2] = base + 500.0
    traj[0, 0] = base + torch.randn(n, 2, generator=g)
    got, ref = _knn_vs_bruteforce(traj, (48, 64), 4, 8)
    np.testing.assert_allclose(got, ref, atol=5e-4)      # flows are O(500): 1e-6 relative


@pytest.mark.parametrize('ring', [12, 40])
def test_knn_many_keys_in_the_kth_bin(ring):
    """A thin annulus of `ring` points around one query with the K-th neighbour inside it: more keys in the
    histogram bin of the K-th smallest than the packed 8-entry list holds.  12 -> the 16-entry re-collection
    path, 40 -> selection by repeated minimum.  Checked against brute force over the whole LUT, forward and
    (through the saved K-th keys) backward."""
    from motionpriorcmax_amd import ops
    from oracle import focus_oracle as O
    shape, sp, K = (64, 96), 4, 26
    g = torch.Generator().manual_seed(50 + ring)
    base = O.tile_mask(shape, 4).nonzero().float()                      # 384 lattice points
    q0 = torch.tensor([8 * sp + sp / 2 - 0.5, 12 * sp + sp / 2 - 0.5])  # centre of LUT cell (8, 12)
    near = q0 + (torch.rand(20, 2, generator=g) - 0.5)                  # 20 points within 0.7 px
    ang = torch.rand(ring, generator=g) * 6.2831853
    rad = 9.0 + (torch.rand(ring, generator=g) - 0.5) * 0.02            # d^2 in 81 +- 0.2: one bin
    annulus = q0 + torch.stack((rad * torch.sin(ang), rad * torch.cos(ang)), -1)
    far = base[((base - q0).norm(dim=1) > 14.0)]                        # lattice elsewhere
    pts = torch.cat((near, annulus, far))
    n = pts.shape[0]
    traj = torch.zeros(1, 2, n, 2)
    traj[0, 1] = pts
    traj[0, 0] = pts + torch.randn(n, 2, generator=g) * 3.0
    got, ref = _knn_vs_bruteforce(traj, shape, sp, K)
    np.testing.assert_allclose(got, ref, atol=2e-5)
    # backward through the same neighbour sets
    L = _loss_obj(dict(image_shape=shape, num_tref=1, num_bins=1, num_knn=K, smooth_weight=0.0, lut_superpixel_size=sp,
                       focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True,
                       polarity_aware_batching=True, interpolation_scheme='mean', smooth_type='on_flow_to_tref'))
    t = traj.to(_dev()).requires_grad_(True)
    lut, _ = ops.KnnLutFn.apply(t, L._cfg)
    gl = torch.randn(lut.shape, generator=g).to(_dev())
    lut.backward(gl)
    tr = traj.clone().requires_grad_(True)
    lref, _ = O.interpolate_flow(tr[:, :1], tr[:, 1:], shape, sp, K, 'l2', 'mean')
    lref.backward(gl.cpu())
    assert _rel_l2(t.grad.cpu().numpy(), tr.grad.numpy()) < 1e-5


def test_cpu_tensors_fail_loudly():
    g = load_golden('g3_squeeze_k1')
    L = _loss_obj(g['cfg'])
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        L.calc(torch.from_numpy(g['trajectories']), torch.from_numpy(g['times']),
               {'events': torch.from_numpy(g['events'])})


# ---- full-size properties (BASELINE sizes: 480x640, 200k events) ----------------------------
def _full_size_inputs(B=2, M=200000, nb=15, seed=3):
    from oracle import focus_oracle as O
    ev, num_pos = O.synth_events(B, M, (480, 640), nb, seed=seed, pad_frac=0.02)
    g = torch.Generator().manual_seed(seed)
    lut = torch.randn(B, nb, 120, 160, 1, 2, generator=g) * 2.0
    return ev, num_pos, lut


def test_full_size_mass_conservation_and_shift():
    """sum(iwe_raw) == sum of w * (in-bounds tap weights); shifting every event by an integer
    number of pixels shifts the raw IWE."""
    from motionpriorcmax_amd import ops
    dev = _dev()
    ev, num_pos, lut = _full_size_inputs()
    L = _loss_obj(dict(image_shape=(480, 640), num_tref=1, num_bins=15, num_knn=32, smooth_weight=0.0,
                       lut_superpixel_size=4, focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True,
                       mask_image_border=True, polarity_aware_batching=True, interpolation_scheme='mean',
                       smooth_type='on_flow_to_tref'))
    t_ref = torch.tensor([0.41], device=dev)
    evd, lutd = ev.to(dev), lut.to(dev)
    _, _, raw = ops.EventFocusFn.apply(lutd, evd, t_ref, L._cfg, num_pos)
    # expected mass, computed independently with torch ops on the device
    b = torch.arange(ev.shape[0], device=dev)[:, None]
    it = evd[..., 4].long()
    iy = torch.div(evd[..., 0], 4, rounding_mode='floor').long()
    ix = torch.div(evd[..., 1], 4, rounding_mode='floor').long()
    pos = lutd[b, it, iy, ix, 0] + evd[..., :2]
    w = evd[..., 5] * (1 - (evd[..., 2] - 0.41).abs().clamp(0, 1))
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
    # integer shift (away from the border so that nothing is clipped differently)
    ev2 = ev.clone()
    inner = (ev[..., 0] > 40) & (ev[..., 0] < 400) & (ev[..., 1] > 40) & (ev[..., 1] < 560)
    ev2[..., 5] = ev2[..., 5] * inner
    ev3 = ev2.clone()
    ev3[..., 0] += 8
    ev3[..., 1] += 12        # multiples of sp keep the LUT cell offsets aligned
    lut0 = torch.zeros_like(lutd)
    _, _, r2 = ops.EventFocusFn.apply(lut0, ev2.to(dev), t_ref, L._cfg, num_pos)
    _, _, r3 = ops.EventFocusFn.apply(lut0, ev3.to(dev), t_ref, L._cfg, num_pos)
    np.testing.assert_allclose(r3[..., 8:, 12:].cpu().numpy(), r2[..., :-8, :-12].cpu().numpy(), atol=1e-4)


@pytest.mark.parametrize('norm', ['l1', 'l2'])
def test_full_size_event_path_vs_oracle(norm):
    """480x640, 200k events: the CPU oracle's event path (LUT given) finishes in seconds, so the
    HIP forward and the hand-derived backward are compared with it directly at BASELINE size."""
    from motionpriorcmax_amd import ops
    from oracle import focus_oracle as O
    dev = _dev()
    ev, num_pos, lut = _full_size_inputs(B=1)
    cfg = dict(image_shape=(480, 640), num_tref=1, num_bins=15, num_knn=32, smooth_weight=0.0,
               lut_superpixel_size=4, focus_loss_norm=norm, dist_norm='l2', scale_iwe_by_dt=True,
               mask_image_border=True, polarity_aware_batching=True, interpolation_scheme='mean',
               smooth_type='on_flow_to_tref')
    lo = lut.clone().requires_grad_(True)
    fo, iwo, rawo = O.FocusLossOracle(**cfg).event_path(ev, lo, torch.tensor([0.41]), num_pos)
    fo.backward()
    L = _loss_obj(cfg)
    lt = lut.to(dev).requires_grad_(True)
    f, blur, raw = ops.EventFocusFn.apply(lt, ev.to(dev), torch.tensor([0.41], device=dev), L._cfg, num_pos)
    f.backward()
    assert abs(f.item() - fo.item()) <= 2e-6 * abs(fo.item())
    np.testing.assert_allclose(raw.cpu().numpy(), rawo.detach().numpy(), atol=1e-5 * rawo.max().item())
    np.testing.assert_allclose(blur.cpu().numpy(), iwo.detach().numpy(), atol=1e-5 * iwo.max().item())
    # a few sign() flips of near-zero Sobel responses are legitimate for 'l1' 
    assert _rel_l2(lt.grad.cpu().numpy(), lo.grad.numpy()) < (2e-3 if norm == 'l1' else 1e-4)


def test_full_size_knn_against_bruteforce_sample():
    """480x640 LUT (19 200 cells, K=32): the HIP LUT equals a brute-force K-nearest mean on a
    random sample of cells (torch on the device as the checker)."""
    from motionpriorcmax_amd import ops
    from oracle import focus_oracle as O
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    coeff = torch.randn(1, 1, 2, 480, 640, generator=g) * 3.0
    times = torch.cat((torch.tensor([0.41]), O.bin_mid_times(15)))
    traj = O.trajectories_at(coeff, times, O.tile_mask((480, 640), 4), 1, 'polynomial').to(dev)
    L = _loss_obj(dict(image_shape=(480, 640), num_tref=1, num_bins=15, num_knn=32, smooth_weight=0.003,
                       lut_superpixel_size=4, focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True,
                       mask_image_border=True, polarity_aware_batching=True, interpolation_scheme='mean',
                       smooth_type='on_flow_to_tref'))
    lut, _ = ops.KnnLutFn.apply(traj, L._cfg)
    grid, hq, wq = O.lut_grid_points((480, 640), 4)
    sel = torch.randperm(hq * wq, generator=g)[:512]
    q = grid[sel].to(dev)
    for t in (0, 7, 14):
        pts = traj[0, 1 + t]
        d = ((q[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
        idx = torch.sort(d, dim=1, stable=True).indices[:, :32]
        f = (traj[0, 0] - pts)[idx].mean(1)
        got = lut[0, t].reshape(-1, 2)[sel.to(dev)]
        assert (got - f).abs().max().item() < 1e-5


def test_hd_sensor_1280x720_large_lut_grid():
    """1280x720 (57 600 LUT cells: beyond the single-workgroup LDS counting sort, so the points are bucketed
    by the global-memory sort): the LUT equals a brute-force K-nearest mean on a sample of cells, the result
    is bitwise reproducible from run to run, the KNN backward matches autograd through the brute-force
    neighbour sets, and the full loss runs with a finite gradient."""
    from motionpriorcmax_amd import ops
    from oracle import focus_oracle as O
    dev = _dev()
    shape = (720, 1280)
    g = torch.Generator().manual_seed(33)
    coeff = torch.randn(1, 1, 2, *shape, generator=g) * 3.0
    nb = 5
    times = torch.cat((torch.tensor([0.41]), O.bin_mid_times(nb)))
    traj = O.trajectories_at(coeff, times, O.tile_mask(shape, 4), 1, 'polynomial').to(dev)
    cfg = dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=32, smooth_weight=0.003,
               lut_superpixel_size=4, focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True,
               mask_image_border=True, polarity_aware_batching=True, interpolation_scheme='mean',
               smooth_type='on_flow_to_tref')
    L = _loss_obj(cfg)
    tr = traj.clone().requires_grad_(True)
    lut, _ = ops.KnnLutFn.apply(tr, L._cfg)
    lut2, _ = ops.KnnLutFn.apply(traj, L._cfg)
    assert torch.equal(lut.detach(), lut2)
    grid, hq, wq = O.lut_grid_points(shape, 4)
    assert hq * wq == 57600
    sel = torch.randperm(hq * wq, generator=g)[:384]
    q = grid[sel].to(dev)
    gsel = torch.randn(nb, 384, 2, generator=g).to(dev)
    tref = traj.clone().requires_grad_(True)
    ref_terms = []
    for t in range(nb):
        pts = tref[0, 1 + t]
        d = ((q[:, None, :] - pts.detach()[None, :, :]) ** 2).sum(-1)
        idx = torch.sort(d, dim=1, stable=True).indices[:, :32]
        f = (tref[0, 0] - pts)[idx].mean(1)
        got = lut[0, t].reshape(-1, 2)[sel.to(dev)]
        assert (got.detach() - f.detach()).abs().max().item() < 1e-5
        ref_terms.append((f * gsel[t]).sum())
    torch.stack(ref_terms).sum().backward()
    gl = torch.zeros_like(lut)
    glv = gl[0].reshape(nb, -1, 2)
    glv[:, sel.to(dev)] = gsel
    lut.backward(gl)
    assert _rel_l2(tr.grad.cpu().numpy(), tref.grad.cpu().numpy()) < 1e-5
    # whole loss on this sensor size
    ev, num_pos = O.synth_events(1, 100000, shape, nb, seed=9)
    t2 = traj.clone().requires_grad_(True)
    loss, _, misc = L.calc(t2, times.to(dev), {'events': ev.to(dev), 'num_pos_events': num_pos})
    loss.backward()
    assert torch.isfinite(loss).item() and torch.isfinite(t2.grad).all().item() and float(t2.grad.abs().sum()) > 0
    assert misc['iwes'].shape == (1, 1, 2, 720, 1280)
    assert abs(float(misc['iwes'].sum()) - 0.0) > 1.0


def test_loss_and_gradient_are_bitwise_reproducible():
    """Run to run, on the same inputs: LUT, IWEs, loss and the gradient are identical bit for bit (integer
    LDS accumulation in the event path, index-ordered buckets in the KNN, no floating-point atomics)."""
    from oracle import focus_oracle as O
    dev = _dev()
    B, M, nb = 2, 60000, 15
    ev, num_pos, _ = _full_size_inputs(B=B, M=M, nb=nb, seed=13)
    coeff = torch.randn(B, 1, 6, 480, 640, generator=torch.Generator().manual_seed(14))
    times = torch.cat((torch.tensor([0.41]), O.bin_mid_times(nb)))
    traj = O.trajectories_at(coeff, times, O.tile_mask((480, 640), 4), 3, 'polynomial')
    L = _loss_obj(dict(image_shape=(480, 640), num_tref=1, num_bins=nb, num_knn=32, smooth_weight=0.003,
                       lut_superpixel_size=4, focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True,
                       mask_image_border=True, polarity_aware_batching=True, interpolation_scheme='mean',
                       smooth_type='on_flow_to_tref'))
    outs = []
    for _ in range(3):
        t = traj.to(dev).requires_grad_(True)
        loss, _, misc = L.calc(t, times.to(dev), {'events': ev.to(dev), 'num_pos_events': num_pos})
        loss.backward()
        outs.append((loss.detach().clone(), misc['iwes'].clone(), t.grad.clone()))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])


def test_contrast_maximisation_recovers_a_known_flow():
    """Beyond parity: optimising a constant flow with the loss' own gradient (Adam, from zero) collapses
    the events of points moving at (6, -9) px per window back onto the points: the recovered flow is the
    true one.  Exercises the whole hand-derived backward end to end."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from toy_flow_recovery import run
    c, v = run(verbose=False)
    assert (c - v).abs().max().item() < 0.3, (c.tolist(), v.tolist())
