#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM-side bytes per launch.

    python tools/summarize_pmc.py <fetch_dir> <write_dir> <out.json> [<calib_fetch_dir> <calib_write_dir>] [--steps N]

The counters are in KiB (MI355X_MICROARCH.md, HBM section).  On gfx950 FETCH_SIZE tallies a 128-byte request at 64 bytes:
tools/ubench/fetch_calib.bin reads 1 GiB (beyond the 256 MiB Infinity Cache) with 4-, 8- and 16-byte accesses per lane and
the calibration passes give bytes / counter = 2.00 for all three; it also writes and then reads a 34 MB buffer (the size of
the path's intermediate images: producer -> consumer through the Infinity Cache), where the counter still tallies the
requests but at their true size when they are served on-die.  So for EVERY kernel, with the same labels:
    fetch_bytes_raw          the counter as it stands        -> with write_bytes: hbm_bytes_lower
    fetch_bytes_x2           the counter x the 1 GiB factor  -> with write_bytes: hbm_bytes_upper
A kernel that streams from HBM sits at the upper figure, one whose input was just written by its producer nearer the lower
one; nothing else is claimed per kernel.  WRITE_SIZE reads the bytes exactly.  FETCH_SIZE and WRITE_SIZE come from separate
passes (they do not fit one).  `launches_per_step` = launches seen / steps of the profiled command (--steps, default 10 + the
8 set-up and 5 warm-up steps and the instrumented and check passes of bench.py make this approximate: it is rounded)."""
import collections
import csv
import glob
import json
import sys

def per_kernel(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                name = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
                agg[name].append(float(r['Counter_Value']))
    return agg


def mean_tail(v):
    v = v[len(v) // 2:]          # skip the warm-up launches
    return 1024.0 * sum(v) / max(len(v), 1)


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith('--')]
    fetch = per_kernel(argv[0], 'FETCH_SIZE')
    write = per_kernel(argv[1], 'WRITE_SIZE')
    factor = 2.0
    calib = None
    if len(argv) > 4:
        cf, cw = per_kernel(argv[3], 'FETCH_SIZE'), per_kernel(argv[4], 'WRITE_SIZE')
        GiB = float(1 << 30)
        calib = {}
        # the three copies share one demangled base name; rocprofv3 lists them in launch order: float, float2, float4
        cc = cf.get('k_calib_copy', [])
        if len(cc) >= 3:
            per = len(cc) // 3 * 3
            for width, off in ((4, 0), (8, 1), (16, 2)):
                v = [cc[i] for i in range(off, per, 3)]
                calib[f'read_1GiB_{width}B_per_lane_bytes_over_counter'] = GiB / (1024.0 * v[-1])
            factor = calib['read_1GiB_16B_per_lane_bytes_over_counter']
        cg = cf.get('k_calib_gather8', [])
        if cg:
            calib['gather_8B_counter_bytes_over_payload'] = 1024.0 * cg[-1] / (GiB / 4)
        cs_ = cf.get('k_calib_small_read', [])
        if cs_:
            calib['read_34MB_just_written_bytes_over_counter'] = 34.0e6 / (1024.0 * cs_[-1]) if cs_[-1] > 0 else None
        ww = cw.get('k_calib_copy', [])
        if len(ww) >= 3:
            calib['write_bytes_over_counter_16B'] = GiB / (1024.0 * ww[len(ww) // 3 * 3 - 1])
    steps = 10
    for a in sys.argv[1:]:
        if a.startswith('--steps='):
            steps = int(a.split('=')[1])
    out = {'_calibration': calib, '_fetch_factor_upper': factor}
    for k in sorted(set(fetch) | set(write)):
        fb = mean_tail(fetch.get(k, [0.0]))
        wb = mean_tail(write.get(k, [0.0]))
        n = len(fetch.get(k, [])) or len(write.get(k, []))
        out[k] = {'fetch_bytes_raw': fb, 'fetch_bytes_x2': fb * factor, 'write_bytes': wb,
                  'hbm_bytes_lower': fb + wb, 'hbm_bytes_upper': fb * factor + wb,
                  'launches_seen': n}
    json.dump(out, open(argv[2], 'w'), indent=1)
    for k, v in out.items():
        if not k.startswith('_') and v['hbm_bytes_upper'] > 1e5:
            print(f"{k[:36]:36s} fetch raw {v['fetch_bytes_raw'] / 1e6:8.1f} MB (x{factor:.2f} upper)  write {v['write_bytes'] / 1e6:8.1f} MB")
    print('calibration:', calib)


if __name__ == '__main__':
    main()
