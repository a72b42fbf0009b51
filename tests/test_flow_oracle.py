"""Dense-flow / flow-metrics oracle against the reference's golden vectors (CPU)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import flow_oracle as F

DF_CASES = ['a', 'b', 'c']
FE_CASES = ['a', 'b', 'c', 'd']
FE_KEYS = ('EPE', '1PE', '2PE', '3PE', 'AE')


def load_flow_golden():
    z = np.load(os.path.join(GOLDEN, 'g10_flow.npz'))
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize('tag', DF_CASES)
def test_dense_flow_oracle_matches_reference(tag):
    g = load_flow_golden()
    dense, patch = F.dense_flow_from_traj(g[f'df_{tag}_traj_flow'], g[f'df_{tag}_pix'], int(g[f'df_{tag}_ps']),
                                          tuple(int(v) for v in g[f'df_{tag}_shape']))
    np.testing.assert_array_equal(patch, g[f'df_{tag}_patch'])
    # fp32 tolerance: torch evaluates the filter weights with a different rounding order
    np.testing.assert_allclose(dense, g[f'df_{tag}_dense'], rtol=1e-5, atol=1e-5)


def test_resize_restatement_matches_torch_interpolate_down_and_odd():
    """The stand-in used for the goldens IS torch's interpolate; check the restatement beyond integer upscaling."""
    x = torch.randn(1, 2, 30, 40, generator=torch.Generator().manual_seed(0))
    for size in ((15, 17), (61, 95), (30, 40)):
        ref = torch.nn.functional.interpolate(x, size=size, mode='bicubic', align_corners=False, antialias=True)
        np.testing.assert_allclose(F.resize_bicubic_aa(x.numpy(), size), ref.numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('tag', FE_CASES)
def test_flow_error_oracle_matches_reference(tag):
    g = load_flow_golden()
    err = F.calculate_flow_error(g[f'fe_{tag}_gt'], g[f'fe_{tag}_pred'], g.get(f'fe_{tag}_mask'), g.get(f'fe_{tag}_scale'))
    got = np.array([float(err[k]) for k in FE_KEYS])
    np.testing.assert_allclose(got, g[f'fe_{tag}_err'], rtol=1e-6)


def test_flow_error_inf_ground_truth_is_nan_as_in_reference():
    """flow.py:48-49 multiplies by the mask, so an infinite ground-truth value poisons the sums."""
    gt, pr, _, _ = F.synth_flow_case(1, 8, 8, 3, False, False)
    gt[0, 0, 2, 2] = float('inf')
    assert torch.isnan(F.calculate_flow_error(gt, pr)['EPE'])
