#!/bin/bash
# Run on the GPU box (gpurun): kernel-trace stats of the default bench command + separate PMC passes.
# Outputs land in gpurun_out/prof_<tag>/ ; summaries are copied to profiles/ by tools/summarize_profile.py.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXTRA=${BENCH_ARGS:-}   # e.g. BENCH_ARGS="--n-ms 10" for the non-coherent configuration
# the trace runs the driver's own step counts (bench.py's defaults, 50 + 5) unless TRACE_ARGS says otherwise: its MEAN launch is
# what bench.py prices roofline.frac from
TRACE_ARGS=${TRACE_ARGS:-"--steps 50 --warmup 5"}
BENCH="python $REPO/bench.py $TRACE_ARGS --no-cpu-baseline --no-tracking --no-pcie --no-native $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1
SMALL="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-tracking --no-pcie --no-native $EXTRA"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $SMALL > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $SMALL > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_sq -o pmc -- $SMALL > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -o pmc -- $SMALL > $OUT/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq3 -o pmc -- $SMALL > $OUT/pmc_sq3.log 2>&1
find $OUT -name "*.csv" | head -40
