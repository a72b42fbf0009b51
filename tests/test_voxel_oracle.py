"""Voxel-grid oracle against the reference's golden vectors (CPU)."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import voxel_oracle as V

VOX_CASES = ['g8_voxel_meanstd', 'g8_voxel_max', 'g8_voxel_raw', 'g8_voxel_q05_meanstd', 'g8_voxel_q10_raw', 'g8_voxel_q02_max']


def load_vox(name):
    import os
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    norm = str(z['norm'])
    return {k: z[k] for k in z.files}, (None if norm == 'None' else norm)


@pytest.mark.parametrize('name', VOX_CASES)
def test_voxel_oracle_matches_reference(name):
    g, norm = load_vox(name)
    out = V.voxel_grid(*(torch.from_numpy(g[k]) for k in ('x', 'y', 't', 'p')), tuple(int(v) for v in g['shape']), norm,
                       float(g.get('quantile', 0.0)))
    np.testing.assert_allclose(out.numpy(), g['grid'], rtol=0, atol=1e-6)
