set -x
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "byte_phase or native_grid or randomised" 2>&1 | tail -5
for i in 1 2 3; do
  timeout 120 python tools/bench_native_grid.py 2>/dev/null | tail -1 | cut -c100-220
done
