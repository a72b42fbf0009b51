#!/usr/bin/env python3
"""Per-kernel means of the SQ counters of one or more rocprofv3 --pmc passes, as one CSV (profiles/*_sq_*.csv).

    python tools/sq_summary.py <out.csv> <pmc_dir> [<pmc_dir> ...]

Every pass is `rocprofv3 --pmc <up to 8 SQ counters> --kernel-trace --output-format csv -d <pmc_dir> -- python3 <probe>`;
counters of different passes are merged by kernel name.  The first half of the launches of a kernel (warm-up) is
dropped.  Derived columns (SQ_* cycle counters are in quad-cycles, MI355X_MICROARCH.md):
  valu_per_wave      SQ_INSTS_VALU / SQ_WAVES                  wave-instructions per wavefront
  valu_busy_frac     SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES-normalised: 4 * SQ_ACTIVE_INST_VALU / (SQ_BUSY_CU_CYCLES * 4 SIMDs) when present
  wait_any_frac      SQ_WAIT_ANY / SQ_WAVE_CYCLES              share of wave lifetime parked on s_waitcnt / barriers
  wait_inst_frac     SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES         share of wave lifetime stalled at issue
  active_frac        SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES"""
import collections
import csv
import glob
import sys


def main():
    out, dirs = sys.argv[1], sys.argv[2:]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                name = r['Kernel_Name'].split('(')[0].replace('void ', '')
                agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
    counters = sorted({c for k in agg for c in agg[k]})
    derived = ['valu_per_wave', 'salu_per_wave', 'lds_per_wave', 'wait_any_frac', 'wait_inst_frac', 'active_frac', 'valu_active_frac']
    with open(out, 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(['kernel', 'launches'] + counters + derived)
        for k in sorted(agg):
            m = {}
            n = 0
            for c, v in agg[k].items():
                v = v[len(v) // 2:]
                m[c] = sum(v) / len(v)
                n = max(n, len(v))

            def ratio(a, b):
                return round(m[a] / m[b], 4) if a in m and b in m and m[b] else ''
            row = [k, n] + [round(m[c], 1) if c in m else '' for c in counters]
            row += [ratio('SQ_INSTS_VALU', 'SQ_WAVES'), ratio('SQ_INSTS_SALU', 'SQ_WAVES'), ratio('SQ_INSTS_LDS', 'SQ_WAVES'),
                    ratio('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES'), ratio('SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES'),
                    ratio('SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES'), ratio('SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES')]
            w.writerow(row)
    print(open(out).read())


if __name__ == '__main__':
    main()
