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
