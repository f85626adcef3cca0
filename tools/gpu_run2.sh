set -x
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "matrix_core or alternative_grid" 2>&1 | tail -25 > gpurun_out/r02b_mxtests.log
cat gpurun_out/r02b_mxtests.log
python bench.py --no-cpu-baseline --no-tracking > gpurun_out/r02b_bench.json 2> gpurun_out/r02b_bench.err
tail -c 400 gpurun_out/r02b_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r02b_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d.get('pcie_inclusive'))
"
