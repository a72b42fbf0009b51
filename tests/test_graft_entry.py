"""The driver's build hook must work in the CPU container (it is the round's "does it build" check)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_build_hook_runs_and_versions_agree():
    import __graft_entry__ as g
    g.build()
    from motionpriorcmax_amd import _lib
    header = open(os.path.join(ROOT, 'include', 'mpcmax.h')).read()
    want = int(re.search(r'#define MPC_VERSION (\d+)', header).group(1))
    assert _lib.lib().mpc_version() == want
    # the binding's own check names the same number
    src = open(os.path.join(ROOT, 'motionpriorcmax_amd', '_lib.py')).read()
    assert f'mpc_version() != {want}' in src
