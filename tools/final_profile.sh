cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_C3.json 2> gpurun_out/bench_err.log; tail -c 600 gpurun_out/bench_C3.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --also "" --no-cpu-baseline --steps 40 > /dev/null 2>&1
cp gpurun_out/ks/*/*kernel_stats.csv gpurun_out/kernel_stats_C3.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pf -- python3 bench.py --also "" --no-cpu-baseline --steps 10 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pw -- python3 bench.py --also "" --no-cpu-baseline --steps 10 > /dev/null 2>&1
python tools/summarize_pmc.py gpurun_out/pf gpurun_out/pw gpurun_out/traffic_C3.json
