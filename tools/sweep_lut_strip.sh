#!/bin/bash
# tuning sweep on the GPU box: rows of a LUT strip (= backward bucket; MPC_EV_CSTRIP_KB) x threads of the ordered k_lut_accum
# (every variant overwrites the in-tree libmpcmax.so: the default build is restored when the script ends, however it ends)
trap "python -m motionpriorcmax_amd.build > /dev/null 2>&1" EXIT
for v in "-DEV_LUT_THREADS_ORD=1024" "-DEV_LUT_THREADS_ORD=512" "-DEV_LUT_THREADS_ORD=512 -DEV_LUT_INFLIGHT_ORD=4" "-DEV_LUT_THREADS_ORD=256 -DEV_LUT_INFLIGHT_ORD=4"; do
  MPC_EXTRA_HIPCC_FLAGS="$v" python -m motionpriorcmax_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  for kb in 48 24 16 12; do
    echo "== $v  MPC_EV_CSTRIP_KB=$kb"
    MPC_EV_CSTRIP_KB=$kb python bench.py --also "" --no-cpu-baseline --no-hip-graph 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_us']
print(d['ms_per_step'], {n:k[n]['us_per_launch'] for n in ('k_ev_bin','k_iwe_accum','k_lut_accum')})"
  done
done
