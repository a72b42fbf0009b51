#!/usr/bin/env python3
"""<tag>_sq.csv + <tag>_kstats.csv (tools/sq_profile.sh) -> the per-kernel JSON bench.py reads for `roofline.knn`.

    python tools/sq_to_json.py <sq.csv> <kstats.csv> <out.json>"""
import csv
import hashlib
import json
import os
import sys


def main():
    ks = {}
    for r in csv.DictReader(open(sys.argv[2])):
        name = r['Name'].split('(')[0].replace('void ', '').split('<')[0]
        ks[name] = float(r['AverageNs']) / 1e3
    out = {}
    for r in csv.DictReader(open(sys.argv[1])):
        name = r['kernel'].split('<')[0]
        if name not in ks or not r.get('SQ_INSTS_VALU'):
            continue
        out[name] = {'valu_insts': float(r['SQ_INSTS_VALU']), 'waves': float(r['SQ_WAVES']), 'kernel_us': ks[name],
                     'valu_per_wave': float(r['valu_per_wave'] or 0), 'wait_any_frac': float(r['wait_any_frac'] or 0),
                     'wait_inst_frac': float(r['wait_inst_frac'] or 0), 'lds_per_wave': float(r['lds_per_wave'] or 0)}
        # (round 6) vector instructions by class (third PMC pass of tools/sq_profile.sh): what tools/valu_floor.py prices
        mix = {c[len('SQ_INSTS_VALU_'):]: float(r[c]) for c in r if c.startswith('SQ_INSTS_VALU_') and r[c] not in ('', None)}
        if mix:
            out[name]['valu_mix'] = mix
        # share of the kernel's cycles in which a CU's LDS was busy (SQ_BUSY_CYCLES is summed over the 32 shader engines of 8 CUs
        # each, SQ_LDS_IDX_ACTIVE over the CUs), and the share of those cycles that were bank-conflict cycles
        busy, act = float(r.get('SQ_BUSY_CYCLES') or 0), float(r.get('SQ_LDS_IDX_ACTIVE') or 0)
        if busy > 0:
            out[name]['lds_active_frac'] = round(act / busy / 8.0, 3)
            out[name]['lds_conflict_frac'] = round(float(r.get('SQ_LDS_BANK_CONFLICT') or 0) / act, 3) if act > 0 else 0.0
    # which library the counters belong to: bench.py prints it beside the figures it takes from this file
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'motionpriorcmax_amd', 'libmpcmax.so')
    if os.path.exists(lib):
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from motionpriorcmax_amd import build as _build
        out['_library'] = 'libmpcmax.so of sources ' + _build.source_hash()       # (the file's own hash changes with every build)
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    print(json.dumps(out, indent=1)[:1500])


if __name__ == '__main__':
    main()
