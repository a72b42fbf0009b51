#!/bin/bash
# The bounds-checked debug build of the library (csrc/bounds.h, -DMPC_BOUNDS) under the golden / parity tests and the three
# differential fuzzers, once per round on the GPU box; the log goes to profiles/ (VERDICT r04 item 8).  The checked library is built
# BESIDE the product one (build_ab/libmpcmax_bounds.so) and loaded through MPC_AB_LIB (motionpriorcmax_amd/_lib.py): an interrupted
# run leaves the product library as it was.
#   bash tools/bounds_run.sh [log]        e.g.  gpurun -- 'bash tools/bounds_run.sh gpurun_out/r05_bounds_run.txt'
LOG=${1:-gpurun_out/bounds_run.txt}
mkdir -p "$(dirname "$LOG")"
BLIB=build_ab/libmpcmax_bounds.so
mkdir -p build_ab
MPC_EXTRA_HIPCC_FLAGS=-DMPC_BOUNDS python -c "from motionpriorcmax_amd import build as b; b.build_library(force=True, out='$BLIB')" > /dev/null 2>&1 || { echo "bounds build failed" | tee "$LOG"; exit 1; }
export MPC_AB_LIB=$BLIB
{
  echo "== $BLIB built with -DMPC_BOUNDS (MPC_AB_LIB); mpc_bounds_check() after every test (tests/conftest.py) and every fuzz batch"
  python -c "from motionpriorcmax_amd import _lib as C; print('mpc_bounds_check() on a fresh library:', C.lib().mpc_bounds_check(), '(0 = bounds build, clean; -1 = product build)')"
  python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_realistic.py tests/test_gpu_event_order.py tests/test_gpu_voxel.py tests/test_gpu_ingest.py tests/test_gpu_errors.py tests/test_gpu_weights.py tests/test_gpu_per_event.py -m gpu -q -x 2>&1 | tail -6
  for f in fuzz_parity fuzz_knn fuzz_aux; do
    echo "== tools/$f.py"
    timeout 900 python tools/$f.py 150 7 2>&1 | tail -3
  done
} 2>&1 | grep -v amdgpu.ids | tee "$LOG"
