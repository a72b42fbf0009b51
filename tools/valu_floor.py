#!/usr/bin/env python3
"""Instruction-mix floor of the two kernels of the path that are bound by vector-instruction issue.

A SIMD of gfx950 issues one wave64 vector instruction per 2 cycles AT BEST; most instructions cost more (tools/ubench/op_cycles.hip,
measured at 8 wavefronts per SIMD with independent operands: profiles/r03_ubench_op_cycles.txt -- 1.6-1.9 cycles for plain fp32 /
integer adds, 2.9 for VOP3 forms, shifts, bit counts, conversions and compares, 5.3 for reciprocals, 5-11 where a select depends
on a compare).  "0.54 of the 2-cycle peak" therefore says little: this tool prices the kernel's OWN instruction mix.

    floor_us = sum over instruction classes (dynamic count x measured issue cost) / (SIMDs x clock)

* dynamic counts per COARSE class (fp32 add / mul / fma / transcendental, int32, int64, conversions, everything else) come from the
  committed SQ counters of the step (profiles/rNN_sq_<workload>.json: `valu_mix`, third PMC pass of tools/sq_profile.sh);
* inside a coarse class the kernel's instructions are priced by the STATIC mix of that class in its disassembly
  (llvm-objdump of the code object inside libmpcmax.so): e.g. the int32 class of k_knn_strip is 45 % plain adds / ands (1.6-1.9
  cycles), 30 % shifts / bit counts / compares (2.9), 10 % selects ...  The unrolled search loops ARE the kernel (the dynamic to
  static ratio of the hot blocks is the same for every class to first order), so this is the mix the wavefronts execute.
* cost of a mnemonic: its class in the table below (ubench name); unlisted mnemonics cost the VOP3 default (2.87).

    python tools/valu_floor.py [--sq profiles/r06_sq_C3.json] [--kernels k_knn_strip,k_knn_bwd_tile] [--json out.json]
"""
import argparse
import collections
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'motionpriorcmax_amd', 'libmpcmax.so')
LLVM = '/opt/rocm/lib/llvm/bin'
SIMDS, CLOCK_HZ = 1024, 2.4e9

# measured issue cost (SIMD cycles per wave64 instruction at 8 wavefronts per SIMD), profiles/r03_ubench_op_cycles.txt
COST = {'fma': 2.63, 'mul': 1.87, 'addf': 1.87, 'minf': 2.87, 'pkfma': 2.87, 'pkadd': 2.87, 'addu': 1.88, 'subu': 1.64, 'and_': 1.64,
        'xor_': 1.64, 'bcnt': 2.87, 'lshl': 2.87, 'vop3': 2.87, 'cmp': 2.87, 'cvt': 2.87, 'mov': 1.64, 'dpp': 2.87, 'trans': 5.34,
        'mul24': 2.87, 'mullo': 2.87, 'cndmask': 2.87, 'cmpcnd_pair': 5.12, 'lane': 2.87, 'f64': 5.34}
# (v_cndmask alone in a loop measures 11.5 cycles -- eight selects reading one VCC serialise -- but beside its compare the pair costs
# 5.1: a select is priced as the rest of a pair, 5.12 - 2.87, rounded up to the VOP3 default)


def classify(mn):
    """mnemonic -> (coarse SQ class, ubench cost class)"""
    m = mn
    for suf in ('_e32', '_e64', '_sdwa', '_dpp', '_e64_dpp'):
        if m.endswith(suf):
            m = m[:-len(suf)]
    dpp = mn.endswith('_dpp')
    if m.startswith('v_pk_fma_f32'):
        return 'FMA_F32', 'pkfma'
    if m.startswith('v_pk_add_f32') or m.startswith('v_pk_mul_f32'):
        return ('ADD_F32' if 'add' in m else 'MUL_F32'), 'pkadd'
    if re.match(r'v_(fma|fmac|mad|mac|fmaak|fmamk)_f32', m):
        return 'FMA_F32', 'fma'
    if re.match(r'v_(add|sub|subrev)_f32', m):
        return 'ADD_F32', ('dpp' if dpp else 'addf')
    if re.match(r'v_mul_f32', m) or m.startswith('v_mul_legacy_f32'):
        return 'MUL_F32', 'mul'
    if re.match(r'v_(rcp|rsq|sqrt|exp|log|sin|cos)_', m):
        return 'TRANS_F32', 'trans'
    if re.match(r'v_cvt_', m):
        return 'CVT', 'cvt'
    if re.search(r'_(f64|u64|i64|b64)$', m) or m.startswith('v_lshl_add_u64') or m.startswith('v_mov_b64'):
        return 'INT64', 'f64' if 'f64' in m else 'vop3'
    if re.match(r'v_cmp', m):
        return 'OTHER', 'cmp'
    if re.match(r'v_cndmask', m):
        return 'OTHER', 'cndmask'
    if re.match(r'v_(readlane|writelane|readfirstlane|permlane|bpermute|swap)', m):
        return 'OTHER', 'lane'
    if re.match(r'v_(mov|not)_b32', m):
        return 'OTHER', ('dpp' if dpp else 'mov')
    if re.match(r'v_(min|max|med3|div_scale|div_fmas|div_fixup|ldexp|frexp|fract|floor|ceil|trunc|rndne)', m) and 'f32' in m:
        return 'OTHER_F32', 'minf'
    if re.match(r'v_(add|addc)_(u32|co_u32|i32)', m):
        return 'INT32', 'addu'
    if re.match(r'v_(sub|subrev|subb)_(u32|co_u32|i32)', m):
        return 'INT32', 'subu'
    if re.match(r'v_(and|or)_b32', m):
        return 'INT32', 'and_'
    if re.match(r'v_xor_b32', m):
        return 'INT32', 'xor_'
    if re.match(r'v_bcnt', m):
        return 'INT32', 'bcnt'
    if re.match(r'v_(lshlrev|lshrrev|ashrrev)_b?i?32', m) or re.match(r'v_(lshl|lshr|ashr)', m):
        return 'INT32', 'lshl'
    if re.match(r'v_mul_(u32_u24|i32_i24|hi)', m) or re.match(r'v_mad_(u32_u24|i32_i24)', m):
        return 'INT32', 'mul24'
    if re.match(r'v_mul_lo_u32', m) or re.match(r'v_mad_u64_u32', m):
        return 'INT32', 'mullo'
    return 'INT32', 'vop3'          # v_add3, v_lshl_add, v_lshl_or, v_and_or, v_or3, v_bfe, v_bfi, v_perm, v_min/max_i32, v_sad ...


def disassemble(kernel_prefix):
    """{mangled-name: [mnemonics]} of the kernels whose demangled name starts with `kernel_prefix`"""
    out = {}
    with tempfile.TemporaryDirectory() as td:
        lib = shutil.copy(LIB, os.path.join(td, 'lib.so'))
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', lib], cwd=td, capture_output=True, text=True)
        for f in sorted(os.listdir(td)):
            if 'gfx950' not in f:
                continue
            txt = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', os.path.join(td, f)], capture_output=True, text=True).stdout
            cur = None
            for line in txt.splitlines():
                m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
                if m:
                    sym = m.group(1)
                    dem = subprocess.run(['c++filt', sym], capture_output=True, text=True).stdout.strip().replace('void ', '')
                    cur = dem if dem.startswith(kernel_prefix) else None
                    if cur is not None:
                        out[cur] = []
                    continue
                if cur is None:
                    continue
                t = line.split()
                if t and re.match(r'^(v_|s_|ds_|global_|buffer_|flat_|scratch_)', t[0]):
                    out[cur].append(t[0])
    return out


def floor_of(kname, variant_hint, sq):
    """floor of one kernel: `sq` = its entry in the SQ json (valu_insts, kernel_us, valu_mix)"""
    ks = disassemble(kname + '<')
    if not ks:
        raise SystemExit(f'no kernel {kname} in {LIB}')
    # the instantiation the benchmark runs: <..., false, false, ...> (no L1, no flow_to_next, no iwd) unless a hint says otherwise
    pick = [k for k in ks if variant_hint in k] or sorted(ks)
    name = pick[0]
    valu = [m for m in ks[name] if m.startswith('v_')]
    static = collections.Counter()
    sub = collections.defaultdict(collections.Counter)
    for m in valu:
        coarse, fine = classify(m)
        static[coarse] += 1
        sub[coarse][fine] += 1
    mean_cost = {c: sum(COST[f] * n for f, n in sub[c].items()) / static[c] for c in static}
    total = float(sq['valu_insts'])
    mix = sq.get('valu_mix')
    if mix:
        dyn = {c: float(mix.get(c, 0.0)) for c in ('ADD_F32', 'MUL_F32', 'FMA_F32', 'TRANS_F32', 'INT32', 'INT64', 'CVT')}
        rest = max(total - sum(dyn.values()), 0.0)
        # what the class counters do not cover (compares, selects, moves, lane ops, min / max ...): by their static shares
        others = {c: static[c] for c in static if c not in dyn}
        so = sum(others.values()) or 1
        for c, n in others.items():
            dyn[c] = rest * n / so
        source = 'SQ class counters x static mix inside a class'
    else:
        st = sum(static.values())
        dyn = {c: total * n / st for c, n in static.items()}
        source = 'static mix of the disassembly scaled to SQ_INSTS_VALU (no class counters in the profile)'
    cyc = sum(dyn[c] * mean_cost.get(c, COST['vop3']) for c in dyn)
    floor_us = cyc / (SIMDS * CLOCK_HZ) * 1e6
    ideal_us = total * 2.0 / (SIMDS * CLOCK_HZ) * 1e6
    return {'kernel': name.split('(')[0], 'static_valu_instructions': len(valu), 'dynamic_valu_wave_instructions': total,
            'dynamic_by_class': {c: round(v) for c, v in sorted(dyn.items())},
            'mean_issue_cycles_by_class': {c: round(v, 2) for c, v in sorted(mean_cost.items())},
            'static_fine_mix': {c: dict(sub[c].most_common()) for c in sorted(sub)},
            'mean_issue_cycles': round(cyc / total, 3), 'floor_us': round(floor_us, 1), 'two_cycle_ideal_us': round(ideal_us, 1),
            'kernel_us_in_profile': round(float(sq['kernel_us']), 1), 'achieved_over_floor': round(floor_us / float(sq['kernel_us']), 3),
            'mix_source': source, 'cost_table': 'profiles/r03_ubench_op_cycles.txt', 'assumes': f'{SIMDS} SIMDs at {CLOCK_HZ / 1e9:.1f} GHz'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sq', default='')
    ap.add_argument('--kernels', default='k_knn_strip,k_knn_bwd_tile')
    ap.add_argument('--variant', default='false, false', help='substring of the template arguments of the instantiation to price')
    ap.add_argument('--json', default='')
    args = ap.parse_args()
    sqf = args.sq
    if not sqf:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, 'profiles')) if re.match(r'r\d+_sq_C3\.json', f))
        sqf = os.path.join(ROOT, 'profiles', cands[-1])
    sq = json.load(open(sqf))
    out = {'sq_profile': os.path.relpath(sqf, ROOT), 'library_of_profile': sq.get('_library')}
    for k in args.kernels.split(','):
        if k not in sq:
            print(f'{k}: not in {sqf}', file=sys.stderr)
            continue
        r = floor_of(k, args.variant, sq[k])
        out[k] = r
        print(f"{r['kernel']}: {r['dynamic_valu_wave_instructions'] / 1e6:.1f} M wave-instructions per launch, mean issue cost {r['mean_issue_cycles']} cycles"
              f" -> floor {r['floor_us']} us (2-cycle ideal {r['two_cycle_ideal_us']} us); measured {r['kernel_us_in_profile']} us = {r['achieved_over_floor']} of the floor's rate")
        for c in r['dynamic_by_class']:
            print(f"    {c:10s} {r['dynamic_by_class'][c] / 1e6:8.2f} M  x {r['mean_issue_cycles_by_class'].get(c, COST['vop3']):5.2f} cycles   {r['static_fine_mix'].get(c, {})}")
    if args.json:
        json.dump(out, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
