#!/usr/bin/env python3
"""VGPRs, scratch bytes, LDS and occupancy-relevant figures of every kernel in libmpcmax.so, read from the code object's
metadata notes (no GPU needed):  python tools/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'motionpriorcmax_amd', 'libmpcmax.so')
LLVM = '/opt/rocm/lib/llvm/bin'


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ''
    with tempfile.TemporaryDirectory() as td:
        # every source file is one bundle inside the fat binary; llvm-objdump extracts them NEXT TO ITS INPUT: work on a copy
        import shutil
        lib = shutil.copy(LIB, os.path.join(td, 'lib.so'))
        subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '--offloading', lib], cwd=td, capture_output=True, text=True)
        files = [f for f in os.listdir(td) if 'gfx950' in f]
        rows = []
        for f in files:
            notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', os.path.join(td, f)], capture_output=True, text=True).stdout
            for blk in notes.split('- .agpr_count:')[1:]:
                def g(k):
                    m = re.search(r'\.' + k + r':\s+(\S+)', blk)
                    return m.group(1) if m else '?'
                name = g('name')
                dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
                rows.append((dem.split('(')[0].replace('void ', ''), g('vgpr_count'), g('vgpr_spill_count'), g('private_segment_fixed_size'),
                             g('group_segment_fixed_size'), g('sgpr_count')))
        for r in sorted(set(rows)):
            if flt in r[0]:
                print('%-70s vgpr %4s spill %3s scratch %5s lds %6s sgpr %3s' % (r[0][:70], *r[1:]))


if __name__ == '__main__':
    main()
