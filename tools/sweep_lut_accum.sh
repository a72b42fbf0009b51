#!/bin/bash
# tuning sweep of k_lut_accum (threads per workgroup, records in flight) on the GPU box: rebuilds the library per variant
# (every variant overwrites the in-tree libmpcmax.so: the default build is restored when the script ends, however it ends)
trap "python -m motionpriorcmax_amd.build > /dev/null 2>&1" EXIT
for v in "-DEV_LUT_THREADS=1024 -DEV_LUT_INFLIGHT=2" "-DEV_LUT_THREADS=1024 -DEV_LUT_INFLIGHT=3" "-DEV_LUT_INFLIGHT_ORD=3" "-DEV_LUT_INFLIGHT_ORD=1"; do
  echo "== $v"
  MPC_EXTRA_HIPCC_FLAGS="$v" python -m motionpriorcmax_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  python tools/ordered_probe.py C3 2>&1 | grep -v "no table" | tail -2
done
