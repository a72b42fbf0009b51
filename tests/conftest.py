import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped automatically where no GPU is visible, so `pytest tests/` is
    # safe in the CPU container even without `-m "not gpu"`.
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _bounds_checked_build(request):
    """With the bounds-checked debug build of the library loaded (-DMPC_BOUNDS, csrc/bounds.h; tools/bounds_run.sh), every GPU test
    ends with mpc_bounds_check(): an out-of-range index into dynamic LDS or a workspace sub-buffer fails the test with the source
    line that produced it.  The product build returns -1 there and nothing happens."""
    yield
    if 'gpu' not in request.keywords or not torch.cuda.is_available():
        return
    from motionpriorcmax_amd import _lib as C
    n = C.lib().mpc_bounds_check()
    if n > 0:
        pytest.fail(C.lib().mpc_last_error_string().decode())


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    cfg = {}
    if 'cfg_keys' in d:
        for k, v in zip(d['cfg_keys'], d['cfg_vals']):
            k, v = str(k), str(v)
            if v in ('True', 'False'):
                cfg[k] = (v == 'True')
            elif k == 'image_shape':
                cfg[k] = tuple(int(s) for s in v.strip('()').split(','))
            else:
                try:
                    cfg[k] = int(v)
                except ValueError:
                    try:
                        cfg[k] = float(v)
                    except ValueError:
                        cfg[k] = v
    d['cfg'] = cfg
    return d


GOLDEN_CASES = ['g1_allflags', 'g2_config1', 'g3_squeeze_k1', 'g3b_tref3', 'g4_iwd_l1_next',
                'g5a_dct3_l2', 'g5b_poly3']
