#!/bin/bash
# tuning sweep on the GPU box: prefetch depth of the marching contrast kernel
# (every variant overwrites the in-tree libmpcmax.so: the default build is restored when the script ends, however it ends)
trap "python -m motionpriorcmax_amd.build > /dev/null 2>&1" EXIT
for v in 4 6 8 2; do
  echo "== CM_PF=$v"
  MPC_EXTRA_HIPCC_FLAGS="-DCM_PF=$v" python -m motionpriorcmax_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  python tools/ordered_probe.py C3 2>&1 | tail -1 | grep -o "'mpc_contrast_fwd': [0-9.]*"
  python tools/ordered_probe.py C2 2>&1 | tail -1 | grep -o "'mpc_contrast_fwd': [0-9.]*"
done
