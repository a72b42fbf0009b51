#!/bin/bash
# Evidence run of a round (R=r03 below) on the GPU box: bench line, kernel-trace stats and SQ counters per workload, PMC traffic with the
# FETCH_SIZE calibration.  Everything lands in gpurun_out/${R}_*; the summaries are copied into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
R=${R:-r04}
for wl in C3 C2 C4; do
  tools/sq_profile.sh ${R}_$wl bench.py --workload $wl --also "" --no-cpu-baseline --no-hip-graph --no-realistic --no-nondefault --steps 10 > /dev/null 2>&1
  python3 tools/sq_to_json.py $O/${R}_${wl}_sq.csv $O/${R}_${wl}_kstats.csv $O/${R}_sq_$wl.json > /dev/null
done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_f -- tools/ubench/fetch_calib.bin > $O/${R}_fetch_calib.txt 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_w -- tools/ubench/fetch_calib.bin > /dev/null 2>&1
for wl in C3 C2 C4; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf_$wl -- python3 bench.py --workload $wl --also "" --no-cpu-baseline --no-hip-graph --no-realistic --no-nondefault --steps 10 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw_$wl -- python3 bench.py --workload $wl --also "" --no-cpu-baseline --no-hip-graph --no-realistic --no-nondefault --steps 10 > /dev/null 2>&1
  python3 tools/summarize_pmc.py $O/pf_$wl $O/pw_$wl $O/traffic_$wl.json $O/cal_f $O/cal_w | tail -12
  rm -rf $O/pf_$wl $O/pw_$wl
done
rm -rf $O/cal_f $O/cal_w
# the bench line LAST, with this run's summaries in profiles/ of the box's copy of the tree: its roofline block quotes them (kernel
# stats, SQ counters of the library that is loaded, PMC traffic)
for wl in C3 C2 C4; do
  cp $O/${R}_${wl}_kstats.csv profiles/${R}_rocprofv3_kernel_stats_${wl}.csv; cp $O/${R}_${wl}_sq.csv profiles/${R}_sq_${wl}.csv
  cp $O/${R}_sq_${wl}.json profiles/${R}_sq_${wl}.json; cp $O/traffic_${wl}.json profiles/traffic_${wl}.json
done
python bench.py > $O/${R}_bench_C3.json 2> $O/${R}_bench_err.log; tail -c 400 $O/${R}_bench_C3.json
