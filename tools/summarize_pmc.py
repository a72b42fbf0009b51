#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.

    python tools/summarize_pmc.py <fetch_dir> <write_dir> <out.json> [<calib_fetch_dir> <calib_write_dir>]

The counters are in KiB (MI355X_MICROARCH.md, HBM section).  On gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced streaming read; other access widths are uncalibrated, so the factor is MEASURED: tools/ubench/fetch_calib.bin
reads 1 GiB with 4-, 8- and 16-byte accesses per lane and gathers 8-byte words at random; the optional calibration
passes give bytes / counter per access width.  Every kernel is listed with its RAW counter bytes and with a corrected
figure that uses the factor of its dominant read width (table KERNEL_WIDTH below, from the kernels' sources); both are
kept, the corrected one is what bench.py reports as `roofline.traffic`.  FETCH_SIZE and WRITE_SIZE come from separate
passes (they do not fit one)."""
import collections
import csv
import glob
import json
import sys

# dominant global READ access of every kernel of the path: 16 = 16 bytes per lane streaming, 8 / 4 = narrower streaming,
# 'g' = gathers (a cache line per lane access)
KERNEL_WIDTH = {
    'k_iwe_accum': 16, 'k_lut_accum': 16,            # float4 records, streamed (k_lut_accum also gathers the adjoint image)
    'k_ev_bin': 8, 'k_knn_bwd_combine': 8, 'k_lut_smooth': 8, 'k_knn_bucket': 8,
    'k_contrast_fused': 4, 'k_finalize': 8, 'k_zero_words': 16,
    'k_knn_strip': 'g', 'k_knn_bwd_tile': 'g', 'k_knn_fallback': 'g', 'k_knn_query': 'g', 'k_knn_bwd_points': 'g',
}


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
    fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
    write = per_kernel(sys.argv[2], 'WRITE_SIZE')
    factors = {16: 2.0, 8: None, 4: None, 'g': None}
    calib = None
    if len(sys.argv) > 5:
        cf, cw = per_kernel(sys.argv[4], 'FETCH_SIZE'), per_kernel(sys.argv[5], 'WRITE_SIZE')
        GiB = float(1 << 30)
        calib = {}
        for width, key in ((4, 'k_calib_copy'), ):
            pass
        # the three copies share one demangled base name; rocprofv3 lists them in launch order: float, float2, float4
        cc = cf.get('k_calib_copy', [])
        if len(cc) >= 3:
            per = len(cc) // 3 * 3
            w4 = [cc[i] for i in range(0, per, 3)]; w8 = [cc[i] for i in range(1, per, 3)]; w16 = [cc[i] for i in range(2, per, 3)]
            for width, v in ((4, w4), (8, w8), (16, w16)):
                calib[f'read_{width}B_per_lane_bytes_over_counter'] = GiB / (1024.0 * v[-1])
                factors[width] = GiB / (1024.0 * v[-1])
        cg = cf.get('k_calib_gather8', [])
        if cg:
            payload = GiB / 4                      # 256 MiB of 8-byte words; + 128 MiB of indices, streamed
            calib['gather_8B_counter_bytes_over_payload'] = 1024.0 * cg[-1] / payload
            # a gather's counter is requests x 64 B as well; whether its requests are 64 or 128 bytes is not resolved by this
            # calibration, so the doubled figure is an UPPER bound for gather-dominated kernels (flagged in the output)
            factors['g'] = 2.0
        ww = cw.get('k_calib_copy', [])
        if len(ww) >= 3:
            calib['write_bytes_over_counter_16B'] = GiB / (1024.0 * ww[len(ww) // 3 * 3 - 1])
    out = {'_calibration': calib, '_factors_used': {str(k): v for k, v in factors.items()}}
    for k in sorted(set(fetch) | set(write)):
        fb = mean_tail(fetch.get(k, [0.0]))
        wb = mean_tail(write.get(k, [0.0]))
        width = KERNEL_WIDTH.get(k)
        fac = factors.get(width)
        corrected = fb * fac if fac else fb
        out[k] = {'fetch_bytes_raw': fb, 'read_width': width, 'fetch_factor': fac if fac else 1.0,
                  'fetch_factor_is_upper_bound': width == 'g' or width is None,
                  'fetch_bytes_corrected': corrected, 'write_bytes': wb,
                  'hbm_bytes_per_launch': corrected + wb, 'hbm_bytes_per_launch_raw': fb + wb,
                  'launches_seen': len(fetch.get(k, []))}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
    for k, v in out.items():
        if not k.startswith('_') and v['hbm_bytes_per_launch'] > 1e5:
            print(f"{k[:36]:36s} fetch raw {v['fetch_bytes_raw'] / 1e6:8.1f} MB x{v['fetch_factor']:.2f}  write {v['write_bytes'] / 1e6:8.1f} MB")
    print('calibration:', calib)


if __name__ == '__main__':
    main()
