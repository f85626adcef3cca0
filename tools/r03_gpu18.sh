python -m pytest tests/test_gpu_steps.py tests/test_c_host_demo.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | head
python tools/bench_tracking_closed_loop.py --channels 256 65536 81920 98304 114688 131072 --ms 1200 --signals 32 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print({k: round(d[k]) if isinstance(d[k], float) else d[k] for k in ('channels','p50_us','p99_us','max_us','steps_over_1ms','real_time','code_and_carrier_lock')})
"
