mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_steps.py -q -x -m gpu 2>&1 | tail -3 > gpurun_out/g_tests.log
python tools/bench_tracking_closed_loop.py --channels 256 16384 65536 98304 131072 --signals 32 --ms 2000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['channels'], d['host_workers'], 'p50 %.0f p99 %.0f max %.0f late %d lock %d behind %.1f' % (d['p50_us'], d['p99_us'], d['max_us'], d['steps_over_1ms'], d['code_and_carrier_lock'], d['behind_at_end_ms']), d['slowest_steady_steps_ms'])" > gpurun_out/g_cl.log
python tools/bench_tracking_closed_loop.py --unpaced --channels 65536 131072 --signals 32 --ms 2000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('unpaced', d['channels'], d['host_workers'], 'p50 %.0f p99 %.0f max %.0f late %d' % (d['p50_us'], d['p99_us'], d['max_us'], d['steps_over_1ms']))" >> gpurun_out/g_cl.log
cat gpurun_out/g_tests.log gpurun_out/g_cl.log
