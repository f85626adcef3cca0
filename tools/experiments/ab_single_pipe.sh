set -x
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "fine_grid_pipeline" 2>&1 | tail -15
for i in 1 2 3; do
  echo -n "pipe:   "; python tools/bench_grid_kernel.py 256 1 20 2>/dev/null | tail -1 | cut -c60-200
  echo -n "legacy: "; GPSX_ACQ_SINGLE_LEGACY=1 python tools/bench_grid_kernel.py 256 1 20 2>/dev/null | tail -1 | cut -c60-200
done
