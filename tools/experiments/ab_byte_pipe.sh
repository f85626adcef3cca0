set -x
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "byte_phase or native_grid" 2>&1 | tail -5
for i in 1 2 3; do
  echo -n "pipe:   "; timeout 120 python tools/bench_native_grid.py 2>/dev/null | tail -1 | cut -c100-220
  echo -n "legacy: "; GPSX_ACQ_BYTE_LEGACY=1 timeout 120 python tools/bench_native_grid.py 2>/dev/null | tail -1 | cut -c100-220
done
timeout 300 python tools/experiments/byte_timeline.py
