"""Randomised differential tests (fixed seeds): the HIP path against the CPU oracles over random small shapes,
flags, batch raggedness and event counts -- the loss (tools/fuzz_parity.py), the next-row operators
(tools/fuzz_aux.py) and the KNN LUT against a brute-force search on the device up to the DSEC grid (tools/fuzz_knn.py).  Run without the caching allocator, so that an out-of-bounds access faults instead of
landing in cached memory (this is how the one in the KNN bucket sort was found)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('tool,n,seed', [('fuzz_parity.py', 40, 21), ('fuzz_aux.py', 25, 22), ('fuzz_knn.py', 60, 23)])
def test_fuzz_against_oracle(tool, n, seed):
    env = dict(os.environ, PYTORCH_NO_CUDA_MEMORY_CACHING='1')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', tool), str(n), str(seed)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and f'{n} cases, 0 bad' in r.stdout, (r.stdout[-1500:], r.stderr[-800:])
