mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "multi_block or non_coherent or matrix_core or overflow or randomised or bench_dist or alternative" 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20
python bench.py --n-ms 10 --no-cpu-baseline --no-tracking --steps 5 2>/dev/null | tail -1 > gpurun_out/r03j_bench10.json
python -c "
import json; d = json.loads(open('gpurun_out/r03j_bench10.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"
python tools/bench_grid_kernel.py 1 10 10 2>/dev/null | tail -1
python bench.py --no-cpu-baseline --no-tracking --steps 20 2>/dev/null | tail -1 > gpurun_out/r03j_bench.json
python -c "
import json; d = json.loads(open('gpurun_out/r03j_bench.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_profiled'))"
