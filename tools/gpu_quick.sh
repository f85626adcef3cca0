# quick A/B on the GPU box: the matrix-core parity tests, then the kernel timings (256 / 64 captures, 10 blocks)
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "matrix_core or bench_size or alternative or multi_block" 2>&1 | tail -4
python tools/bench_grid_kernel.py 256 1 20 2>/dev/null | tail -1
python tools/bench_grid_kernel.py 64 1 20 2>/dev/null | tail -1
python tools/bench_grid_kernel.py 64 10 5 2>/dev/null | tail -1
