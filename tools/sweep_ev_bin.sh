#!/bin/bash
# tuning sweep on the GPU box: events per thread of k_ev_bin on the ingest's bucket-ordered layout (bench.py per-kernel times)
# (every variant overwrites the in-tree libmpcmax.so: the default build is restored when the script ends, however it ends)
trap "python -m motionpriorcmax_amd.build > /dev/null 2>&1" EXIT
for v in "-DEV_PER_THREAD=1" "-DEV_PER_THREAD=2" "-DEV_PER_THREAD=4" "-DEV_PER_THREAD=4 -DEV_STAGE=2304"; do
  echo "== $v"
  MPC_EXTRA_HIPCC_FLAGS="$v" python -m motionpriorcmax_amd.build > /dev/null 2>&1 || echo BUILD FAILED
  python bench.py --also "" --no-cpu-baseline --no-hip-graph 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_us']
print(d['ms_per_step'], {n:k[n]['us_per_launch'] for n in ('k_ev_bin','k_iwe_accum','k_lut_accum')}, d['other_event_layout']['ms_per_step'])"
done
