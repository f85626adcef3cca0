#!/bin/bash
# Round 6's measurement pass on the GPU box (through gpurun): the default bench line, the kernel traces + counters the line's
# roofline blocks are priced from (headline kernel, ten-block kernel, vector-ALU kernel), the VALU class-rate micro-benchmark.
# tools/r06_collect.sh condenses gpurun_out/ into profiles/ afterwards.
set -x
T=${1:-r06}
mkdir -p gpurun_out
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
tail -c 400 gpurun_out/${T}_bench.err
cp bench_detail.json gpurun_out/${T}_bench_detail.json
bash tools/profile_bench.sh ${T} > gpurun_out/${T}_prof.log 2>&1
BENCH_ARGS="--n-ms 10" TRACE_ARGS="--steps 10 --warmup 2" bash tools/profile_bench.sh ${T}_10ms > gpurun_out/${T}_prof10.log 2>&1
BENCH_ARGS="--acq-path vector" TRACE_ARGS="--steps 10 --warmup 2" bash tools/profile_bench.sh ${T}_poly > gpurun_out/${T}_profpoly.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates > gpurun_out/${T}_valu_rates_microbench.txt 2>&1
tail -3 gpurun_out/${T}_valu_rates_microbench.txt
head -c 4200 gpurun_out/${T}_bench.json
