python -m pytest tests -m gpu -x -q -k "multi_block or non_coherent or matrix_core or overflow or randomised or bench_dist or alternative or full_cold" 2>&1 | grep -E "passed|failed|rror|assert" | head
python tools/bench_grid_kernel.py 256 10 3 2>/dev/null | tail -1
