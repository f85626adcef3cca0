python -m pytest tests -m gpu -x -q -k "alternative or full_cold or golden or randomised or windows or bench_size or kernels_do_not" 2>&1 | grep -E "passed|failed|rror|assert" | head
for n in 1 2 3 4 6; do python tools/bench_grid_kernel.py $n 1 20 2>/dev/null | tail -1; done
GPSX_ACQ_SPLIT=2 python tools/bench_grid_kernel.py 1 1 20 2>/dev/null | tail -1
