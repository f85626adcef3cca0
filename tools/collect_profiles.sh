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
[ -s gpurun_out/${T}_native.json ] && cp gpurun_out/${T}_native.json profiles/${T}_native_grid_32x29x2046.json
[ -s gpurun_out/${T}_pcie_probe.txt ] && grep contexts gpurun_out/${T}_pcie_probe.txt > profiles/${T}_pcie_probe.txt
[ -d gpurun_out/prof_${T}_track ] && python tools/summarize_track_profile.py ${T}_track 212992 > /dev/null
[ -d gpurun_out/prof_${T}_track_loop ] && python tools/summarize_track_profile.py ${T}_track_loop 212992 k_track_loop 20 > /dev/null
[ -s gpurun_out/${T}_track_loop_kernel_us.json ] && cp gpurun_out/${T}_track_loop_kernel_us.json profiles/${T}_track_loop_kernel_us.json
[ -s gpurun_out/prof_${T}_native/trace/trace_kernel_stats.csv ] && cp gpurun_out/prof_${T}_native/trace/trace_kernel_stats.csv profiles/${T}_native_grid_kernel_stats.csv
[ -s gpurun_out/${T}_native_pmc_summary.json ] && cp gpurun_out/${T}_native_pmc_summary.json profiles/${T}_native_grid_pmc_summary.json
[ -s gpurun_out/${T}_gputests.log ] && cp gpurun_out/${T}_gputests.log profiles/${T}_gputests.log
if [ -f gpurun_out/${T}_sweep.txt ]; then
python - "$T" <<'PY'
import json, sys
rows = []
for l in open(f"gpurun_out/{sys.argv[1]}_sweep.txt"):
    if l.startswith("{"):
        d = json.loads(l)
        rows.append({"captures": d["searches"], "blocks_per_search": d["n_ms"], "kernel": d["kernel"],
                     "ms_per_launch": d["ms"], "hyp_per_s": d["hyp_per_s"]})
json.dump({"command": "tools/gpu_validate.sh launch-size sweep (tools/bench_grid_kernel.py, $GPSX_ACQ_ALGO=mx|poly; 32 PRN x 21 Doppler x 16368 phases "
                      "per capture, captures resident in HBM)", "tag": sys.argv[1], "rows": rows},
          open(f"profiles/{sys.argv[1]}_launch_size_sweep.json", "w"), indent=1)
PY
fi
ls profiles | grep "^$T"
