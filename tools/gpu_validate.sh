set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r02l_gputests.log
cat gpurun_out/r02l_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r02l_bench.json 2> gpurun_out/r02l_bench.err
tail -c 300 gpurun_out/r02l_bench.err
python bench.py --searches 64 --no-cpu-baseline --no-tracking > gpurun_out/r02l_bench_64.json 2>> gpurun_out/r02l_bench.err
python bench.py --n-ms 10 --no-cpu-baseline --no-tracking --steps 5 > gpurun_out/r02l_bench_10ms.json 2>> gpurun_out/r02l_bench.err
python bench.py --amp-scale 1.0 --no-cpu-baseline --no-tracking --steps 10 > gpurun_out/r02l_bench_strong.json 2>> gpurun_out/r02l_bench.err
bash tools/profile_bench.sh r02l > gpurun_out/r02l_prof.log 2>&1
BENCH_ARGS="--n-ms 10" bash tools/profile_bench.sh r02l_10ms > gpurun_out/r02l_prof10.log 2>&1
bash tools/profile_track.sh r02l_track > gpurun_out/r02l_proft.log 2>&1
python tools/bench_track_kernel.py > gpurun_out/r02l_trackkernel.json 2>&1
python tools/bench_tracking.py > gpurun_out/r02l_tracking_latency.json 2>/dev/null
bash tools/gpu_sweep.sh > gpurun_out/r02l_sweep.txt 2>&1
python tools/bench_native_grid.py 2>/dev/null | tail -1 > gpurun_out/r02l_native.json
python tools/pcie_probe.py > gpurun_out/r02l_pcie_probe.txt 2>&1
python - <<'PY'
import json
for f in ("r02l_bench", "r02l_bench_64", "r02l_bench_10ms", "r02l_bench_strong"):
    d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d.get("pcie_inclusive", {}).get("value"), (d.get("tracking") or {}).get("value"))
PY
