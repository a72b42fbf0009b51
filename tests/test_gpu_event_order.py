"""SURVEY.md 8f-1, layout half: events ordered by (time bin, LUT strip) inside each polarity block
(mpc_event_bucket_order) and the step that uses the offsets table (no per-event record for the backward).
The reference's loss does not depend on the row order inside a polarity block (focus.py:182-230 sums over events), and
here the accumulators are integers, so everything below is checked bit for bit."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(shape, nb, **over):
    c = dict(image_shape=shape, num_tref=1, num_bins=nb, num_knn=8, smooth_weight=0.003, lut_superpixel_size=4,
             focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True, mask_image_border=True,
             polarity_aware_batching=True, interpolation_scheme='mean', smooth_type='on_flow_to_tref')
    c.update(over)
    return c


def _case(shape, B, M, nb, seed, mag=3.0):
    from oracle import focus_oracle as O
    ev, num_pos = O.synth_events(B, M, shape, nb, seed=seed, pad_frac=0.03)
    g = torch.Generator().manual_seed(seed)
    coeff = torch.randn(B, 1, 2, *shape, generator=g) * mag
    times = torch.cat((torch.tensor([0.3]), O.bin_mid_times(nb)))
    traj = O.trajectories_at(coeff, times, O.tile_mask(shape, 4), 1, 'polynomial')
    return ev, num_pos, traj, times


def _step(L, traj, times, batch):
    t = traj.clone().requires_grad_(True)
    loss, log, misc = L.calc(t, times, batch)
    loss.backward()
    return loss.detach().clone(), t.grad.clone(), misc['iwes'].clone()


@pytest.mark.parametrize('split', [True, False])
def test_layout_of_the_ordered_tensor(split):
    from motionpriorcmax_amd import LossFactory, _lib as C, ops
    shape, B, M, nb = (96, 128), 3, 9000, 5
    ev, num_pos, _, _ = _case(shape, B, M, nb, 3)
    ev[1, 5:40, 5] = 0.0                                      # padding-like rows in the middle of a block
    ev[0, :, 0] = ev[0, :, 0] * 1.2 - 8                       # rows outside the image: LUT row clamps
    L = LossFactory.get_loss_calculator('FOCUS', _cfg(shape, nb, polarity_aware_batching=split))
    dev = torch.device('cuda:0')
    out = L.order_events({'events': ev.to(dev), 'num_pos_events': num_pos})
    o, offs = out['events'].cpu().numpy(), out['event_offsets'].cpu().numpy()
    sh = ops.make_shape(L._cfg, B, M, num_pos if split else M, 1)
    S = int(C.lib().mpc_event_lut_strips(ctypes.byref(sh)))
    hq = -(-shape[0] // 4)
    Rr = -(-hq // S)
    assert offs.shape == (B, 2, nb * S + 1)
    e = ev.numpy()
    Mp = num_pos if split else M
    for b in range(B):
        for pol, (r0, r1) in enumerate(((0, Mp), (Mp, M))):
            # same multiset of rows in the block
            src = e[b, r0:r1]; dst = o[b, r0:r1]
            assert sorted(map(bytes, src)) == sorted(map(bytes, dst))
            of = offs[b, pol]
            assert of[0] == r0 and np.all(np.diff(of) >= 0) and of[-1] <= r1
            key = np.where(dst[:, 5] == 0, nb * S,
                           np.clip(dst[:, 4].astype(np.int64), 0, nb - 1) * S
                           + np.clip(np.floor(dst[:, 0] / np.float32(4)).astype(np.int64), 0, hq - 1) // Rr)
            want = np.searchsorted(key, np.arange(nb * S + 1)) + r0      # key must be non-decreasing
            assert np.all(np.diff(key) >= 0)
            np.testing.assert_array_equal(of, want)


@pytest.mark.parametrize('over', [
    {}, {'polarity_aware_batching': False}, {'scale_iwe_by_dt': False, 'mask_image_border': False},
    {'focus_loss_norm': 'l2', 'dist_norm': 'l1', 'interpolation_scheme': 'iwd'}, {'smooth_type': 'on_flow_to_next'},
    {'smooth_weight': 0.0}, {'lut_superpixel_size': 8},
])
@pytest.mark.parametrize('fused', [True, False])
def test_same_loss_and_gradient_bit_for_bit(over, fused, monkeypatch):
    from motionpriorcmax_amd import LossFactory, ops
    monkeypatch.setattr(ops, 'FUSED_CALLS', fused)
    shape, B, M, nb = (96, 128), 3, 15000, 5
    ev, num_pos, traj, times = _case(shape, B, M, nb, 7, mag=6.0)    # mag 6: some events leave the image
    dev = torch.device('cuda:0')
    L = LossFactory.get_loss_calculator('FOCUS', _cfg(shape, nb, **over))
    batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
    ordered = L.order_events(batch)
    plain = {'events': ordered['events'], 'num_pos_events': num_pos}
    traj, times = traj.to(dev), times.to(dev)
    l0, g0, i0 = _step(L, traj, times, batch)
    l1, g1, i1 = _step(L, traj, times, plain)          # ordered rows, record path
    l2, g2, i2 = _step(L, traj, times, ordered)        # ordered rows + offsets: no records
    assert g0.abs().max() > 0
    for l, g, i in ((l1, g1, i1), (l2, g2, i2)):
        assert torch.equal(l, l0) and torch.equal(i, i0)
        assert torch.equal(g, g0), (g - g0).abs().max().item()


def test_ordered_step_against_the_oracle():
    from motionpriorcmax_amd import LossFactory
    from oracle import focus_oracle as O
    shape, B, M, nb = (64, 96), 2, 6000, 4
    cfg = _cfg(shape, nb)
    ev, num_pos, traj, times = _case(shape, B, M, nb, 21)
    to = traj.clone().requires_grad_(True)
    lo, _, mo = O.FocusLossOracle(**cfg).calc(to, times, {'events': ev, 'num_pos_events': num_pos})
    lo.backward()
    dev = torch.device('cuda:0')
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    ordered = L.order_events({'events': ev.to(dev), 'num_pos_events': num_pos})
    lg, gg, ig = _step(L, traj.to(dev), times.to(dev), ordered)
    assert abs(lg.item() - lo.item()) <= 1e-5 * abs(lo.item())
    np.testing.assert_allclose(ig.cpu().numpy(), mo['iwes'].numpy(), rtol=0, atol=1e-5 * mo['iwes'].abs().max().item())
    assert (gg.cpu() - to.grad).norm() / to.grad.norm() < 1e-2


def test_unit_weight_rows_and_empty_blocks():
    """All events of one polarity (an empty block), and a sample made of padding only."""
    from motionpriorcmax_amd import LossFactory
    shape, B, M, nb = (96, 128), 2, 5000, 5
    ev, _, traj, times = _case(shape, B, M, nb, 5)
    ev[1, :, 5] = 0.0
    dev = torch.device('cuda:0')
    L = LossFactory.get_loss_calculator('FOCUS', _cfg(shape, nb))
    for num_pos in (0, M):
        batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
        l0, g0, _ = _step(L, traj.to(dev), times.to(dev), batch)
        l1, g1, _ = _step(L, traj.to(dev), times.to(dev), L.order_events(batch))
        assert torch.equal(l0, l1) and torch.equal(g0, g1)


def test_no_events_at_all():
    """M = 0: the table is all zeros and the step runs (found by tools/fuzz_parity.py)."""
    from motionpriorcmax_amd import LossFactory
    shape, B, nb = (96, 128), 2, 5
    _, _, traj, times = _case(shape, B, 100, nb, 3)
    dev = torch.device('cuda:0')
    L = LossFactory.get_loss_calculator('FOCUS', _cfg(shape, nb))
    batch = {'events': torch.zeros(B, 0, 6, device=dev), 'num_pos_events': 0}
    ob = L.order_events(batch)
    assert ob['events'].shape == (B, 0, 6) and int(ob['event_offsets'].abs().sum()) == 0
    l0, g0, _ = _step(L, traj.to(dev), times.to(dev), batch)
    l1, g1, _ = _step(L, traj.to(dev), times.to(dev), ob)
    assert (torch.equal(l0, l1) or (torch.isnan(l0) and torch.isnan(l1))) and torch.equal(torch.nan_to_num(g0), torch.nan_to_num(g1))


def test_offsets_are_checked():
    from motionpriorcmax_amd import LossFactory
    shape, B, M, nb = (96, 128), 2, 4000, 5
    ev, num_pos, traj, times = _case(shape, B, M, nb, 9)
    dev = torch.device('cuda:0')
    L = LossFactory.get_loss_calculator('FOCUS', _cfg(shape, nb))
    ordered = L.order_events({'events': ev.to(dev), 'num_pos_events': num_pos})
    bad = dict(ordered); bad['event_offsets'] = ordered['event_offsets'][:, :, :-1].contiguous()
    with pytest.raises(ValueError, match='event_offsets'):
        L.calc(traj.to(dev), times.to(dev), bad)
    bad['event_offsets'] = ordered['event_offsets'].to(torch.int64)
    with pytest.raises(ValueError, match='event_offsets'):
        L.calc(traj.to(dev), times.to(dev), bad)
    # a table that does not belong to the tensor (wild values): wrong results, but no fault.  (Seeded: one table in twenty makes
    # the forward drop every event -- an infinite loss, which is what the reference returns for an empty image too; the unseeded
    # draw of rounds 2-3 made this test fail at that rate.)
    gen = torch.Generator(device=dev).manual_seed(0)
    wild = dict(ordered); wild['event_offsets'] = torch.randint(-10**6, 10**6, ordered['event_offsets'].shape, dtype=torch.int32, device=dev, generator=gen)
    l, g, _ = _step(L, traj.to(dev), times.to(dev), wild)
    assert torch.isfinite(g).all()
    for seed in (193, 250):                      # two of the tables that empty the image: the call still returns
        gen = torch.Generator(device=dev).manual_seed(seed)
        wild['event_offsets'] = torch.randint(-10**6, 10**6, ordered['event_offsets'].shape, dtype=torch.int32, device=dev, generator=gen)
        l, g, _ = _step(L, traj.to(dev), times.to(dev), wild)
        assert g.shape == traj.shape
    # num_tref > 1 has no bucketed layout
    L2 = LossFactory.get_loss_calculator('FOCUS', _cfg(shape, nb, num_tref=3, scale_iwe_by_dt=False, polarity_aware_batching=False))
    with pytest.raises(ValueError, match='bucketed'):
        L2.order_events({'events': ev.to(dev)})


def test_c3_full_batch_ordered():
    """BASELINE configs[1] at full size: the ordered step equals the time-ordered one bit for bit."""
    from motionpriorcmax_amd import LossFactory
    import bench
    dev = torch.device('cuda:0')
    wl = bench.WORKLOADS['C3']
    ev, num_pos, traj, times = bench.synth_inputs(wl, seed=1)
    L = LossFactory.get_loss_calculator('FOCUS', bench.loss_config(wl))
    batch = {'events': ev.to(dev), 'num_pos_events': num_pos}
    traj, times = traj.to(dev), times.to(dev)
    l0, g0, i0 = _step(L, traj, times, batch)
    l1, g1, i1 = _step(L, traj, times, L.order_events(batch))
    assert torch.equal(l0, l1) and torch.equal(i0, i1) and torch.equal(g0, g1)
