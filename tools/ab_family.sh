#!/bin/bash
# A/B of compile-time variants on one input family of tools/realistic_probe.py: tools/ab_family.sh <family> "<flags A>" "<flags B>" ...
fam=$1; shift
for flags in "$@"; do
  MPC_EXTRA_HIPCC_FLAGS="$flags" python motionpriorcmax_amd/build.py > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== flags: [$flags]"
  python tools/realistic_probe.py --families $fam --steps 10 2>&1 | grep "^$fam" | cut -c1-600
done
python motionpriorcmax_amd/build.py > /dev/null 2>&1
