mkdir -p gpurun_out; rm -f gpurun_out/g_cl.log
python -m pytest tests/test_gpu_steps.py -q -x -m gpu 2>&1 | tail -3 > gpurun_out/g_tests.log
for cfg in "0 0 0" "4 1 0" "4 0 0" "4 1 1" "4 0 1" "2 1 1" "8 1 1" "0 0 0"; do
  set -- $cfg
  echo "== GPSX_STEP_CHUNKS=$1 HOT=$2 WAIT=$3" >> gpurun_out/g_cl.log
  GPSX_STEP_CHUNKS=$1 GPSX_STEP_HOT=$2 GPSX_CHUNK_WAIT=$3 python tools/bench_tracking_closed_loop.py --channels 65536 131072 --signals 32 --ms 1000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['channels'], 'p50 %.0f p99 %.0f max %.0f late %d lock %d' % (d['p50_us'], d['p99_us'], d['max_us'], d['steps_over_1ms'], d['code_and_carrier_lock']))" >> gpurun_out/g_cl.log
done
cat gpurun_out/g_tests.log gpurun_out/g_cl.log
