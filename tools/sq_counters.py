#!/usr/bin/env python3
"""Per-kernel mean of every counter found in rocprofv3 --pmc output directories (diagnostics).

    python tools/sq_counters.py <kernel-substring> <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys


def main():
    pat = sys.argv[1]
    for d in sys.argv[2:]:
        for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if pat in r['Kernel_Name']:
                    agg[r['Counter_Name']].append(float(r['Counter_Value']))
            for k, v in sorted(agg.items()):
                v = v[len(v) // 2:]
                print(f'{k:28s} {sum(v) / len(v):16.0f}   ({len(v)} launches)')


if __name__ == '__main__':
    main()
