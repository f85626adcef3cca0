mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "byte_phases or randomised or kernels_do_not_write" 2>&1 | tail -3
python tools/bench_native_grid.py 2>/dev/null | tail -1 > gpurun_out/r03g_native.json
cat gpurun_out/r03g_native.json
python tools/bench_native_grid.py --searches 1 2>/dev/null | tail -1
