python -m pytest tests -m gpu -x -q -k "byte_phases or randomised or kernels_do_not or multi_block or matrix_core or alternative" 2>&1 | grep -E "passed|failed|rror|assert" | head
python tools/bench_native_grid.py 2>/dev/null | tail -1 | cut -c1-250
python tools/bench_native_grid.py --searches 1 2>/dev/null | tail -1 | cut -c1-250
python tools/bench_native_grid.py --searches 8 2>/dev/null | tail -1 | cut -c1-250
python bench.py --no-cpu-baseline --no-tracking --no-pcie --steps 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_profiled'))"
