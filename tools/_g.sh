mkdir -p gpurun_out; rm -f gpurun_out/g_ab.log
for i in 1 2 3; do
  for lib in "" _b; do
    echo -n "lib$lib weak: " >> gpurun_out/g_ab.log
    GPSX_LIB=stm32f4_sdr_gps_amd/lib/libgpsx$lib.so timeout 300 python tools/bench_grid_kernel.py 256 1 20 2>/dev/null | tail -1 | cut -c60-200 >> gpurun_out/g_ab.log
  done
done
cat gpurun_out/g_ab.log
cp stm32f4_sdr_gps_amd/lib/libgpsx.so /tmp/libgpsx_main.so
cp stm32f4_sdr_gps_amd/lib/libgpsx_b.so stm32f4_sdr_gps_amd/lib/libgpsx.so
GPSX_ACQ_NO_SPLIT=1 timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "full_cold_start or saturated or bench_size or matrix_core" 2>&1 | tail -15 > gpurun_out/g_tests.log
cp /tmp/libgpsx_main.so stm32f4_sdr_gps_amd/lib/libgpsx.so
cat gpurun_out/g_tests.log
