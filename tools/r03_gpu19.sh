for pf in 2 4 8 16; do
GPSX_STEP_PREFETCH=$pf python tools/bench_tracking_closed_loop.py --channels 98304 --ms 1000 --signals 32 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('pf=$pf', {k: round(d[k]) if isinstance(d[k], float) else d[k] for k in ('channels','p50_us','p99_us','max_us','steps_over_1ms')})
"
done
