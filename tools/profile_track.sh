#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats + PMC passes of the tracking correlator kernel alone
# (tools/bench_track_kernel.py: device-resident block, states and accumulators).  tools/profile_track.sh <tag> [channels]
set -u
TAG=${1:-r02_track}
CH=${2:-212992}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD=${TRACK_CMD:-"python $REPO/tools/bench_track_kernel.py $CH"}   # TRACK_CMD: another driver, e.g. tools/bench_track_loop_kernel.py 20 212992
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o pmc -- $CMD > $OUT/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc_sq3 -o pmc -- $CMD > $OUT/pmc_sq3.log 2>&1
tail -2 $OUT/trace.log
