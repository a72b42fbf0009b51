#!/bin/bash
for flags in "$@"; do
  MPC_EXTRA_HIPCC_FLAGS="$flags" python motionpriorcmax_amd/build.py > /dev/null 2>&1 || { echo "build failed: $flags"; continue; }
  echo "== flags: [$flags]"
  for wl in C2 C4; do python tools/realistic_probe.py --workload $wl --steps 50 --families white 2>&1 | grep "^white" | cut -c1-330; done
done
python motionpriorcmax_amd/build.py > /dev/null 2>&1
