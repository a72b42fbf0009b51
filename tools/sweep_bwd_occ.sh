#!/bin/bash
# tuning sweep on the GPU box: register budget of k_knn_bwd_tile
# (every variant overwrites the in-tree libmpcmax.so: the default build is restored when the script ends, however it ends)
trap "python -m motionpriorcmax_amd.build > /dev/null 2>&1" EXIT
for v in "-DKNN_BW_OCC=7" "-DKNN_BW_OCC=6" "-DKNN_BW_OCC=8"; do
  echo "== $v"
  MPC_EXTRA_HIPCC_FLAGS="$v" python -m motionpriorcmax_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  python tools/ordered_probe.py C3 2>&1 | tail -1 | grep -o "'mpc_knn_lut_bwd': [0-9.]*"
  python tools/ordered_probe.py C2 2>&1 | tail -1 | grep -o "'mpc_knn_lut_bwd': [0-9.]*"
done
