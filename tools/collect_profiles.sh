# after tools/gpu_validate.sh ran on the GPU box under tag $1: condense gpurun_out/ into the tracked files under profiles/
set -e
T=$1
cd "$(dirname "$0")/.."
python tools/summarize_profile.py $T 256 1 > /dev/null
python tools/summarize_profile.py ${T}_10ms 256 10 > /dev/null
for p in "bench:bench_1gpu" "bench_64:bench_1gpu_64_captures" "bench_10ms:bench_1gpu_config4_10ms" "bench_strong:bench_1gpu_strong_signal"; do
  tail -1 gpurun_out/${T}_${p%%:*}.json > profiles/${T}_${p##*:}.json
done
cp gpurun_out/${T}_tracking_latency.json profiles/${T}_tracking_latency.json
ls profiles | grep "^$T"
