mkdir -p gpurun_out; rm -f gpurun_out/g_ab.log
python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "grid or mx or byte or alternative or mag8" 2>&1 | tail -3 > gpurun_out/g_tests.log
for i in 1 2 3; do
  for lib in "" _c; do
    echo -n "lib$lib: " >> gpurun_out/g_ab.log
    GPSX_LIB=stm32f4_sdr_gps_amd/lib/libgpsx$lib.so python tools/bench_grid_kernel.py 256 1 20 2>/dev/null | tail -1 | cut -c60-200 >> gpurun_out/g_ab.log
  done
done
cat gpurun_out/g_tests.log gpurun_out/g_ab.log
