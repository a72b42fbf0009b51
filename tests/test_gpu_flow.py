"""Dense flow + flow metrics (next row 8f-3): HIP path vs the reference's golden vectors and the CPU oracle."""
import numpy as np
import pytest
import torch

from test_flow_oracle import DF_CASES, FE_CASES, FE_KEYS, load_flow_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('tag', DF_CASES)
def test_dense_flow_golden(tag):
    from motionpriorcmax_amd.utils import dense_flow_from_traj
    g = load_flow_golden()
    dense, patch = dense_flow_from_traj(torch.from_numpy(g[f'df_{tag}_traj_flow']).to(DEV),
                                        torch.from_numpy(g[f'df_{tag}_pix']).to(DEV), int(g[f'df_{tag}_ps']),
                                        tuple(int(v) for v in g[f'df_{tag}_shape']))
    np.testing.assert_array_equal(patch.cpu().numpy(), g[f'df_{tag}_patch'])
    np.testing.assert_allclose(dense.cpu().numpy(), g[f'df_{tag}_dense'], rtol=1e-5, atol=1e-5)   # fp32


def test_dense_flow_dsec_size_vs_oracle_and_properties():
    """480 x 640, patch 4 (dsec.yaml), B=2: against the CPU restatement; a constant field stays constant
    (the taps are renormalised at the borders) and the operator is linear."""
    from motionpriorcmax_amd.utils import dense_flow_from_traj, get_optical_flow_tile_mask
    from oracle import flow_oracle as F
    H, W, ps, B = 480, 640, 4, 2
    pix = torch.nonzero(get_optical_flow_tile_mask((H, W), ps))
    g = torch.Generator().manual_seed(11)
    tf = torch.randn(B, pix.shape[0], 2, generator=g) * 5
    dense, patch = dense_flow_from_traj(tf.to(DEV), pix.to(DEV), ps, (H, W))
    ref_dense, ref_patch = F.dense_flow_from_traj(tf.numpy(), pix.numpy(), ps, (H, W))
    np.testing.assert_array_equal(patch.cpu().numpy(), ref_patch)
    np.testing.assert_allclose(dense.cpu().numpy(), ref_dense, rtol=1e-5, atol=2e-5)
    const, _ = dense_flow_from_traj(torch.full_like(tf, 2.5).to(DEV), pix.to(DEV), ps, (H, W))
    assert float((const - 2.5).abs().max()) < 1e-5
    tf2 = torch.randn(B, pix.shape[0], 2, generator=g)
    d2, _ = dense_flow_from_traj(tf2.to(DEV), pix.to(DEV), ps, (H, W))
    d12, _ = dense_flow_from_traj((tf + 3 * tf2).to(DEV), pix.to(DEV), ps, (H, W))
    assert float((d12 - (dense + 3 * d2)).abs().max()) < 2e-4


def test_dense_flow_general_scales_and_sparse_positions():
    """Non-integer up- and down-scaling paths of the filter; a sparse trajectory list leaves zeros elsewhere."""
    from motionpriorcmax_amd import _lib as C
    from motionpriorcmax_amd.utils import dense_flow_from_traj
    from oracle import flow_oracle as F
    g = torch.Generator().manual_seed(5)
    H, W, ps = 37, 53, 5            # patch grid 7 x 10 -> 37 x 53: scale 0.189.., borders clipped
    pix = torch.tensor([[0, 0], [12, 31], [34, 49], [20, 7]])
    tf = torch.randn(1, 4, 3, generator=g)
    dense, patch = dense_flow_from_traj(tf.to(DEV), pix.to(DEV), ps, (H, W))
    rd, rp = F.dense_flow_from_traj(tf.numpy(), pix.numpy(), ps, (H, W))
    np.testing.assert_array_equal(patch.cpu().numpy(), rp)
    assert int((patch != 0).sum()) == 12
    np.testing.assert_allclose(dense.cpu().numpy(), rd, rtol=1e-5, atol=1e-5)
    # patch 1 with H, W equal to the grid: identity resize
    tf = torch.randn(1, 6 * 8, 2, generator=g)
    pix = torch.nonzero(torch.ones(6, 8, dtype=torch.bool))
    dense, patch = dense_flow_from_traj(tf.to(DEV), pix.to(DEV), 1, (6, 8))
    np.testing.assert_allclose(dense.cpu().numpy(), patch.cpu().numpy(), rtol=0, atol=1e-6)


@pytest.mark.parametrize('tag', FE_CASES)
def test_flow_error_golden(tag):
    from motionpriorcmax_amd.utils import calculate_flow_error
    g = load_flow_golden()
    t = lambda k: None if k not in g else torch.from_numpy(g[k]).to(DEV)
    err = calculate_flow_error(t(f'fe_{tag}_gt'), t(f'fe_{tag}_pred'), t(f'fe_{tag}_mask'), t(f'fe_{tag}_scale'))
    got = np.array([float(err[k]) for k in FE_KEYS])
    np.testing.assert_allclose(got, g[f'fe_{tag}_err'], rtol=2e-5)                 # fp32 sums, acosf


@pytest.mark.parametrize('shape', [(8, 480, 640), (3, 37, 53)])
def test_flow_error_vs_oracle(shape):
    """DSEC size (vector path) and an odd size (scalar path), with mask and time scale; identical flows give 0."""
    from motionpriorcmax_amd.utils import calculate_flow_error, ErrorCalculatorFactory
    from oracle import flow_oracle as F
    B, H, W = shape
    gt, pr, em, ts = F.synth_flow_case(B, H, W, seed=21)
    ref = F.calculate_flow_error(gt, pr, em, ts)
    err = calculate_flow_error(gt.to(DEV), pr.to(DEV), em.to(DEV), ts.to(DEV))
    for k in FE_KEYS:
        assert abs(float(err[k]) - float(ref[k])) <= 2e-5 * abs(float(ref[k])), k
    same = calculate_flow_error(gt.to(DEV), gt.to(DEV), em.to(DEV))
    assert float(same['EPE']) == 0.0 and float(same['1PE']) == 0.0
    assert float(same['AE']) < 0.05            # acos near 1 amplifies fp32 rounding of the cosine (same in the reference)
    run = ErrorCalculatorFactory.get_error_calculator('DSEC').run(
        {'flow': pr.to(DEV)}, {'forward_flow': gt.to(DEV), 'flow_valid': em.to(DEV)})
    ref2 = F.calculate_flow_error(gt, pr, em)
    assert abs(float(run['EPE']) - float(ref2['EPE'])) <= 2e-5 * float(ref2['EPE'])
    with pytest.raises(ValueError):
        ErrorCalculatorFactory.get_error_calculator('KITTI')


def test_flow_error_inf_and_empty_mask():
    from motionpriorcmax_amd.utils import calculate_flow_error
    from oracle import flow_oracle as F
    gt, pr, _, _ = F.synth_flow_case(2, 16, 16, 3, False, False)
    none = calculate_flow_error(gt.to(DEV), pr.to(DEV), torch.zeros(2, 1, 16, 16, dtype=torch.bool, device=DEV))
    assert all(float(none[k]) == 0.0 for k in FE_KEYS)          # 0 / 1e-5
    gt[0, 0, 2, 2] = float('inf')
    assert torch.isnan(calculate_flow_error(gt.to(DEV), pr.to(DEV))['EPE'])    # as the reference (inf * 0)
