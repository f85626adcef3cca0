mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r03h_gputests.log
python tools/bench_native_grid.py 2>/dev/null | tail -1 > gpurun_out/r03h_native.json
cat gpurun_out/r03h_native.json
python bench.py --no-cpu-baseline --no-tracking --steps 20 2>/dev/null | tail -1 > gpurun_out/r03h_bench.json
python -c "
import json; d = json.loads(open('gpurun_out/r03h_bench.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('pcie_inclusive', {}).get('value'))"
python bench.py --n-ms 10 --no-cpu-baseline --no-tracking --steps 5 2>/dev/null | tail -1 > gpurun_out/r03h_bench10.json
python -c "
import json; d = json.loads(open('gpurun_out/r03h_bench10.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
