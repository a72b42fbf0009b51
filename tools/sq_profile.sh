#!/bin/bash
# SQ-counter passes of one probe program on the GPU box: tools/sq_profile.sh <tag> <probe.py> [probe args...]
# -> gpurun_out/<tag>_sq.csv (per-kernel means, tools/sq_summary.py) and gpurun_out/<tag>_kstats.csv (kernel-trace stats)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$tag; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- python3 "$@" > $O/ks.log 2>&1
cp $O/ks/*/*kernel_stats.csv gpurun_out/${tag}_kstats.csv
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p1 -- python3 "$@" > $O/p1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/p2 -- python3 "$@" > $O/p2.log 2>&1
# (round 6) the vector instructions by class, for the instruction-mix floor of tools/valu_floor.py
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT --kernel-trace --output-format csv -d $O/p3 -- python3 "$@" > $O/p3.log 2>&1
python3 tools/sq_summary.py gpurun_out/${tag}_sq.csv $O/p1 $O/p2 $O/p3 > /dev/null
rm -rf $O/ks $O/p1 $O/p2 $O/p3
