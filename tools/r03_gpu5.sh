set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -k "byte_phases or randomised or graph or track_epl_256 or rejects_prns" 2>&1 | tail -5 > gpurun_out/r03e_tests.log
cat gpurun_out/r03e_tests.log
python tools/bench_native_grid.py 2>/dev/null | tail -1 > gpurun_out/r03e_native.json
cat gpurun_out/r03e_native.json
python tools/bench_tracking_closed_loop.py --channels 256 16384 32768 65536 98304 131072 --ms 1200 --signals 32 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print({k: d[k] for k in ('channels','p50_us','p99_us','max_us','steps_over_1ms','warmup_max_us','real_time','code_and_carrier_lock')})
" | tee gpurun_out/r03e_closed_loop.txt
