#!/bin/bash
# Round 6's measurement pass on the GPU box (through gpurun), in the order that makes the line and its evidence ONE box's:
#   1. kernel traces + counters of the three grid kernels the line prices (headline k_acq_mx<0>, ten-block k_acq_mx<3>,
#      vector-ALU k_acq_poly), the VALU class-rate micro-benchmark;
#   2. their summaries into profiles/ ON THE BOX (tools/r06_collect.sh regenerates the same files at home from gpurun_out/);
#   3. the default bench run, which reads those summaries: roofline.frac = flops / the mean launch of THIS box's trace.
set -x
T=${1:-r06}
mkdir -p gpurun_out
bash tools/profile_bench.sh ${T} > gpurun_out/${T}_prof.log 2>&1
BENCH_ARGS="--n-ms 10" TRACE_ARGS="--steps 10 --warmup 2" bash tools/profile_bench.sh ${T}_10ms > gpurun_out/${T}_prof10.log 2>&1
BENCH_ARGS="--acq-path vector" TRACE_ARGS="--steps 10 --warmup 2" bash tools/profile_bench.sh ${T}_poly > gpurun_out/${T}_profpoly.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates > gpurun_out/${T}_valu_rates_microbench.txt 2>&1
python tools/summarize_profile.py $T 256 1 > /dev/null
python tools/summarize_profile.py ${T}_10ms 256 10 > /dev/null
python tools/summarize_profile.py ${T}_poly 256 1 k_acq_poly > /dev/null
python tools/valu_class_rates.py gpurun_out/${T}_valu_rates_microbench.txt $T > /dev/null
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"
tail -c 400 gpurun_out/${T}_bench.err
cp bench_detail.json gpurun_out/${T}_bench_detail.json
head -c 4200 gpurun_out/${T}_bench.json
