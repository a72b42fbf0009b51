#!/bin/bash
# kernel-trace stats of the KNN probe (C3 shape) for several values of MPC_KNN_BWD_G: tools/kb_sweep.sh "<G list>" [B] [workload]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/kb_sweep; rm -rf $O; mkdir -p $O
B=${2:-14}; WL=${3:-C3}
for G in $1; do
  export MPC_KNN_BWD_G=$G
  if [ "$G" = "old" ]; then unset MPC_KNN_BWD_G; export MPC_KNN_BWD_SCATTER=0; fi
  if [ "$G" = "auto" ]; then unset MPC_KNN_BWD_G; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$G -- python3 tools/knn_probe.py $B $WL > $O/log_$G.txt 2>&1
  echo "== G=$G"; python3 - <<PY
import csv,glob
f=glob.glob('$O/ks_$G/*/*kernel_stats.csv')
rows=list(csv.DictReader(open(f[0]))) if f else []
for r in rows:
    if 'knn' in r['Name']: print('  %-60s calls %4s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
  unset MPC_KNN_BWD_SCATTER
done
