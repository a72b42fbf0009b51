#!/bin/bash
# A/B of compile-time variants on several families: ab_multi.sh "<fam,fam>" "<flags A>" "<flags B>" ...
fams=$1; shift
for flags in "$@"; do
  MPC_EXTRA_HIPCC_FLAGS="$flags" python motionpriorcmax_amd/build.py > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== flags: [$flags]"
  python tools/realistic_probe.py --families $fams --steps 10 2>&1 | grep -v amdgpu | grep -v "^{" | cut -c1-230
done
python motionpriorcmax_amd/build.py > /dev/null 2>&1
