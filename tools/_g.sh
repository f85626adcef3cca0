mkdir -p gpurun_out; rm -f gpurun_out/g_cl.log
timeout 900 python -m pytest tests/test_gpu_steps.py -q -x -m gpu 2>&1 | tail -3 > gpurun_out/g_tests.log
for rep in 1 2 3; do
python tools/bench_tracking_closed_loop.py --channels 16384 65536 98304 131072 --signals 32 --ms 1200 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['channels'], d['host_workers'], 'p50 %.0f p99 %.0f max %.0f late %d behind %.1f thr %s ms' % (d['p50_us'], d['p99_us'], d['max_us'], d['steps_over_1ms'], d['behind_at_end_ms'], d['cpu_quota_throttled_ms_during_run']), d['slowest_steady_steps_ms'])" >> gpurun_out/g_cl.log
done
uptime >> gpurun_out/g_cl.log
cat gpurun_out/g_tests.log gpurun_out/g_cl.log
