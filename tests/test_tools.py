"""Host-side checks of the measurement tools (no GPU): the instruction-mix floor of tools/valu_floor.py is computed from the committed SQ
profile and the code object inside libmpcmax.so (llvm-objdump), so it can be reproduced -- and kept working -- in the build container."""
import json
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def _latest_sq():
    prof = os.path.join(ROOT, 'profiles')
    c = sorted(f for f in os.listdir(prof) if re.match(r'r\d+_sq_C3\.json', f))
    return os.path.join(prof, c[-1])


@pytest.mark.skipif(not os.path.exists('/opt/rocm/lib/llvm/bin/llvm-objdump'), reason='no llvm-objdump')
def test_valu_floor_reproduces_from_the_committed_profile():
    import valu_floor as vf
    sq = json.load(open(_latest_sq()))
    for kname in ('k_knn_strip', 'k_knn_bwd_tile'):
        assert kname in sq and sq[kname].get('valu_mix'), 'the committed SQ profile carries the instruction-class pass'
        r = vf.floor_of(kname, 'false, false', sq[kname])
        # the floor lies between the 2-cycle ideal and the measured time; every class got a cost; the counts add up to SQ_INSTS_VALU
        assert r['two_cycle_ideal_us'] < r['floor_us'] < r['kernel_us_in_profile']
        assert 2.0 < r['mean_issue_cycles'] < 3.0
        assert abs(sum(r['dynamic_by_class'].values()) - r['dynamic_valu_wave_instructions']) <= 1e-3 * r['dynamic_valu_wave_instructions']
        assert r['static_valu_instructions'] > 500 and r['mix_source'].startswith('SQ class counters')


def test_valu_floor_classifies_the_mnemonics_of_the_hot_loops():
    import valu_floor as vf
    want = {'v_fma_f32': ('FMA_F32', 'fma'), 'v_pk_fma_f32': ('FMA_F32', 'pkfma'), 'v_sub_f32_e32': ('ADD_F32', 'addf'), 'v_mul_f32_e32': ('MUL_F32', 'mul'),
            'v_cvt_pk_u8_f32': ('CVT', 'cvt'), 'v_bcnt_u32_b32': ('INT32', 'bcnt'), 'v_and_b32_e32': ('INT32', 'and_'), 'v_add_u32_e32': ('INT32', 'addu'),
            'v_rcp_f32_e32': ('TRANS_F32', 'trans'), 'v_cndmask_b32_e64': ('OTHER', 'cndmask'), 'v_cmp_lt_i32_e32': ('OTHER', 'cmp'),
            'v_readlane_b32': ('OTHER', 'lane'), 'v_lshl_add_u64': ('INT64', 'vop3'), 'v_max_f32_e32': ('OTHER_F32', 'minf'), 'v_mov_b32_e32': ('OTHER', 'mov')}
    for mn, cls in want.items():
        assert vf.classify(mn) == cls, (mn, vf.classify(mn))
        assert cls[1] in vf.COST
