mkdir -p gpurun_out
for cfg in "14 2000" "14 200" "10 2000" "8 20000"; do set -- $cfg
GPSX_STEP_THREADS=$1 GPSX_STEP_SPIN=$2 python tools/bench_tracking_closed_loop.py --channels 16384 65536 98304 --ms 1200 --signals 32 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$cfg', {k: round(d[k]) if isinstance(d[k], float) else d[k] for k in ('channels','p50_us','p99_us','max_us','steps_over_1ms','real_time')})
"
done | tee gpurun_out/r03f_closed_loop.txt
