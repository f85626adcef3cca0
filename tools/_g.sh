mkdir -p gpurun_out; rm -f gpurun_out/g_ab.log
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "grid or mx or byte or alternative or mag8 or bench" 2>&1 | tail -3 > gpurun_out/g_tests.log
for i in 1 2 3; do
  for lib in "" _b _c; do
    echo -n "lib$lib weak: " >> gpurun_out/g_ab.log
    GPSX_LIB=stm32f4_sdr_gps_amd/lib/libgpsx$lib.so python tools/bench_grid_kernel.py 256 1 20 2>/dev/null | tail -1 | cut -c60-200 >> gpurun_out/g_ab.log
    echo -n "lib$lib strong: " >> gpurun_out/g_ab.log
    GPSX_BENCH_AMP=1.0 GPSX_LIB=stm32f4_sdr_gps_amd/lib/libgpsx$lib.so python tools/bench_grid_kernel.py 256 1 20 2>/dev/null | tail -1 | cut -c60-200 >> gpurun_out/g_ab.log
  done
done
python tools/bench_tracking_closed_loop.py --channels 256 16384 65536 131072 163840 --signals 32 --ms 2000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['channels'], d['host_workers'], 'p50 %.0f p99 %.0f max %.0f late %d lock %d behind %.1f' % (d['p50_us'], d['p99_us'], d['max_us'], d['steps_over_1ms'], d['code_and_carrier_lock'], d['behind_at_end_ms']), d['slowest_steady_steps_ms'])" > gpurun_out/g_cl.log
cat gpurun_out/g_tests.log gpurun_out/g_ab.log gpurun_out/g_cl.log
