"""Event ingest (next row 8f-1): HIP path vs the reference's collate golden and the numpy oracle."""
import numpy as np
import pytest
import torch

from test_ingest_oracle import load_ingest

pytestmark = pytest.mark.gpu


def _run(x, y, t, p, counts, H, W, nb, voxel=False):
    from motionpriorcmax_amd.utils import ingest_events
    dev = torch.device('cuda:0')
    return ingest_events(torch.from_numpy(x).to(dev), torch.from_numpy(y).to(dev), torch.from_numpy(t).to(dev),
                         torch.from_numpy(p).to(dev), torch.from_numpy(counts), (H, W), nb, want_voxel_input=voxel)


def test_ingest_golden_collate():
    g = load_ingest()
    out = _run(g['x'], g['y'], g['t'], g['p'], g['counts'], int(g['H']), int(g['W']), int(g['nb']))
    assert out['num_pos_events'] == int(g['num_pos_events'])
    np.testing.assert_array_equal(out['events'].cpu().numpy(), g['events'])


def test_ingest_full_size_ragged_vs_oracle_and_feeds_the_loss():
    """DSEC-size windows of different length (one empty), bit-exact against the oracle; the result is
    accepted by FocusLoss.calc and the voxel rows by the voxel-grid builder."""
    from motionpriorcmax_amd import LossFactory
    from motionpriorcmax_amd.utils import voxel_grids
    from oracle import focus_oracle as O
    from oracle import ingest_oracle as I
    H, W, nb = 480, 640, 15
    ns = [200000, 120000, 0, 64]
    N = max(ns)
    raws = [I.synth_raw(n, H, W, seed=70 + b) if n else None for b, n in enumerate(ns)]
    pad = lambda k, dt: np.stack([np.concatenate((r[k], np.zeros(N - len(r[k]), dt))) if r else np.zeros(N, dt) for r in raws])
    x, y, t, p = pad(0, 'float32'), pad(1, 'float32'), pad(2, 'int64'), pad(3, 'float32')
    out = _run(x, y, t, p, np.array(ns, dtype=np.int32), H, W, nb, voxel=True)
    empty = (np.zeros((0, 5), np.float32), np.zeros((0, 5), np.float32))
    ref, num_pos = I.collate([I.sample_events(*r, H, W, nb) if r else empty for r in raws])
    assert out['num_pos_events'] == num_pos
    np.testing.assert_array_equal(out['events'].cpu().numpy(), ref)
    np.testing.assert_array_equal(out['xytp'][0].cpu().numpy(), I.voxel_input(*raws[0]))
    # downstream: the loss and the voxel grid take it as is
    cfg = dict(image_shape=(H, W), num_tref=1, num_bins=nb, num_knn=32, smooth_weight=0.003,
               lut_superpixel_size=4, focus_loss_norm='l1', dist_norm='l2', scale_iwe_by_dt=True,
               mask_image_border=True, polarity_aware_batching=True, interpolation_scheme='mean',
               smooth_type='on_flow_to_tref')
    g = torch.Generator().manual_seed(1)
    times = torch.cat((torch.tensor([0.41]), O.bin_mid_times(nb)))
    traj = O.trajectories_at(torch.randn(len(ns), 1, 2, H, W, generator=g), times, O.tile_mask((H, W), 4), 1, 'polynomial')
    loss, _, misc = LossFactory.get_loss_calculator('FOCUS', cfg).calc(
        traj.cuda(), times.cuda(), {'events': out['events'], 'num_pos_events': out['num_pos_events']})
    assert torch.isfinite(loss).item() and misc['iwes'].shape == (len(ns), 1, 2, H, W)
    # the bucketed layout straight from ingest (layout half): same loss and gradient bit for bit
    from motionpriorcmax_amd.utils import ingest_events
    L = LossFactory.get_loss_calculator('FOCUS', cfg)
    dev = torch.device('cuda:0')
    ordered = ingest_events(*(torch.from_numpy(a).to(dev) for a in (x, y, t, p)), torch.tensor(ns, dtype=torch.int32), (H, W), nb, order_for=L)
    assert ordered['num_pos_events'] == num_pos and 'event_offsets' in ordered
    # ... written in bucket order by ingest itself (mpc_ingest_scatter_ordered): the same table, and bucket by bucket the same
    # rows, as ordering the time-ordered tensor afterwards (mpc_event_bucket_order); padding rows stay zero
    two = L.order_events({'events': out['events'], 'num_pos_events': num_pos})
    assert torch.equal(ordered['event_offsets'], two['event_offsets'])
    offs = ordered['event_offsets'].cpu().numpy()
    ea, eb = ordered['events'].cpu().numpy(), two['events'].cpu().numpy()
    M = ea.shape[1]
    for b in range(len(ns)):
        for pol in range(2):
            o = offs[b, pol]
            end = num_pos if pol == 0 else M
            assert (np.diff(o) >= 0).all() and o[0] == (0 if pol == 0 else num_pos) and o[-1] <= end
            assert not ea[b, o[-1]:end].any() and not eb[b, o[-1]:end].any()
            for k in np.nonzero(np.diff(o))[0]:
                ra, rb = ea[b, o[k]:o[k + 1]], eb[b, o[k]:o[k + 1]]
                ia = np.lexsort(ra.T[::-1]); ib = np.lexsort(rb.T[::-1])
                assert np.array_equal(ra[ia], rb[ib]), (b, pol, k)
    res = []
    for batch in ({'events': out['events'], 'num_pos_events': num_pos}, ordered):
        tg = traj.cuda().requires_grad_(True)
        l, _, m = L.calc(tg, times.cuda(), batch)
        l.backward()
        res.append((l.detach(), tg.grad, m['iwes']))
    assert all(torch.equal(a, b) for a, b in zip(*res))
    vox = voxel_grids(out['xytp'][:2], torch.tensor(ns[:2], dtype=torch.int32), (nb, H, W), 'mean_std')
    assert torch.isfinite(vox).all()


def test_ingest_single_event_window_matches_numpy():
    """A window of ONE event normalises its time as 0/0: the reference's numpy arithmetic gives t = NaN and
    bin = num_bins (NaN sorts last); the kernel reproduces that instead of inventing a value."""
    from oracle import ingest_oracle as I
    H, W, nb = 40, 56, 5
    ns = [1, 7]
    raws = [I.synth_raw(n, H, W, seed=90 + b, spill=0.0) for b, n in enumerate(ns)]
    N = max(ns)
    pad = lambda k, dt: np.stack([np.concatenate((r[k], np.zeros(N - len(r[k]), dt))) for r in raws])
    out = _run(pad(0, 'float32'), pad(1, 'float32'), pad(2, 'int64'), pad(3, 'float32'), np.array(ns, dtype=np.int32), H, W, nb)
    with np.errstate(invalid='ignore'):
        ref, num_pos = I.collate([I.sample_events(*r, H, W, nb) for r in raws])
    assert out['num_pos_events'] == num_pos
    got = out['events'].cpu().numpy()
    assert np.array_equal(got, ref, equal_nan=True)
    assert np.isnan(got[0, :, 2]).sum() == 1 and (got[0][np.isnan(got[0, :, 2])][:, 4] == nb).all()
