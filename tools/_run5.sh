python tools/overlap_pair_probe.py C3 > gpurun_out/r05_overlap_pairs.txt 2>&1; cat gpurun_out/r05_overlap_pairs.txt | grep -v amdgpu
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench5.json 2> gpurun_out/r05_bench5.err; tail -c 400 gpurun_out/r05_bench5.err; head -c 1500 gpurun_out/r05_bench5.json
bash tools/bounds_run.sh gpurun_out/r05_bounds_run.txt > /dev/null 2>&1; cat gpurun_out/r05_bounds_run.txt
