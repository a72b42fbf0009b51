python -m pytest tests -m gpu -x -q > gpurun_out/r05_t6.txt 2>&1; tail -4 gpurun_out/r05_t6.txt
python tools/overlap_pair_probe.py C3 2>&1 | grep -v amdgpu > gpurun_out/r05_overlap_pairs.txt; cat gpurun_out/r05_overlap_pairs.txt
F=white,unet,translate40,translate60,diverge-30,diverge+45,diverge-45
echo "== KS_FAR_ROT=1 (default)" > gpurun_out/r05_ab6.txt
python tools/realistic_probe.py --families $F --steps 12 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> gpurun_out/r05_ab6.txt
MPC_EXTRA_HIPCC_FLAGS=-DKS_FAR_ROT=0 python motionpriorcmax_amd/build.py >/dev/null 2>&1
echo "== KS_FAR_ROT=0" >> gpurun_out/r05_ab6.txt
python tools/realistic_probe.py --families $F --steps 12 2>&1 | grep -v amdgpu.ids | cut -c1-400 >> gpurun_out/r05_ab6.txt
python motionpriorcmax_amd/build.py >/dev/null 2>&1
cat gpurun_out/r05_ab6.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench6.json 2> gpurun_out/r05_bench6.err; tail -c 300 gpurun_out/r05_bench6.err; head -c 1200 gpurun_out/r05_bench6.json
