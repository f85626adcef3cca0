python -m pytest tests -m gpu -x -q -k "alternative or full_cold or golden or randomised or windows" 2>&1 | grep -E "passed|failed|rror|assert" | head
for n in 1 2 3; do python tools/bench_grid_kernel.py $n 1 20 2>/dev/null | tail -1; done
GPSX_ACQ_SPLIT=4 python tools/bench_grid_kernel.py 1 1 20 2>/dev/null | tail -1
