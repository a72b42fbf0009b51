#!/bin/bash
# tuning sweep on the GPU box: rows per band of the marching contrast kernel
# (every variant overwrites the in-tree libmpcmax.so: the default build is restored when the script ends, however it ends)
trap "python -m motionpriorcmax_amd.build > /dev/null 2>&1" EXIT
for v in 16 24 48 60 32; do
  echo "== MPC_CT_H=$v"
  MPC_EXTRA_HIPCC_FLAGS="-DMPC_CT_H=$v" python -m motionpriorcmax_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  python tools/ordered_probe.py C3 2>&1 | tail -1 | grep -o "'mpc_contrast_fwd': [0-9.]*"
  python tools/ordered_probe.py C2 2>&1 | tail -1 | grep -o "'mpc_contrast_fwd': [0-9.]*"
done
