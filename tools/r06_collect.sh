#!/bin/bash
# after tools/r06_measure.sh <tag> ran on the GPU box: condense gpurun_out/ into the tracked files under profiles/
set -e
T=${1:-r06}
cd "$(dirname "$0")/.."
python tools/summarize_profile.py $T 256 1 > /dev/null
python tools/summarize_profile.py ${T}_10ms 256 10 > /dev/null
python tools/summarize_profile.py ${T}_poly 256 1 k_acq_poly > /dev/null
python tools/valu_class_rates.py gpurun_out/${T}_valu_rates_microbench.txt $T > /dev/null
tail -1 gpurun_out/${T}_bench.json > profiles/${T}_bench_1gpu.json
cp gpurun_out/${T}_bench_detail.json profiles/${T}_bench_detail.json
ls profiles | grep "^$T"
