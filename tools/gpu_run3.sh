set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02c_gputests.log
cat gpurun_out/r02c_gputests.log
python bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err
tail -c 300 gpurun_out/r02c_bench.err
python bench.py --n-ms 10 --no-cpu-baseline --no-tracking --steps 10 > gpurun_out/r02c_bench_10ms.json 2>> gpurun_out/r02c_bench.err
GPSX_ACQ_ALGO=poly python bench.py --n-ms 10 --no-cpu-baseline --no-tracking --steps 5 > gpurun_out/r02c_bench_10ms_poly.json 2>> gpurun_out/r02c_bench.err
bash tools/profile_bench.sh r02c > gpurun_out/r02c_prof.log 2>&1
python - <<'PY'
import json
for f in ("r02c_bench", "r02c_bench_10ms", "r02c_bench_10ms_poly"):
    d = json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("pcie_inclusive", {}).get("value"))
PY
