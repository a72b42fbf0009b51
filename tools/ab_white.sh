#!/bin/bash
# A/B of compile-time variants on the white-noise C3 step: tools/ab_white.sh "<flags A>" "<flags B>" ...   (run on the GPU box)
for flags in "$@"; do
  MPC_EXTRA_HIPCC_FLAGS="$flags" python motionpriorcmax_amd/build.py > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== flags: [$flags]"
  python tools/realistic_probe.py --families white --steps 20 2>&1 | grep "^white" | cut -c1-600
done
python motionpriorcmax_amd/build.py > /dev/null 2>&1
