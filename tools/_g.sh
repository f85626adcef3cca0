mkdir -p gpurun_out; rm -f gpurun_out/g_ab.log
for i in 1 2; do
  for lib in "" _c; do
    echo -n "lib$lib walk: " >> gpurun_out/g_ab.log
    GPSX_LIB=stm32f4_sdr_gps_amd/lib/libgpsx$lib.so python tools/bench_grid_kernel.py 256 10 5 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/g_ab.log
    echo -n "lib$lib byte: " >> gpurun_out/g_ab.log
    GPSX_LIB=stm32f4_sdr_gps_amd/lib/libgpsx$lib.so python tools/bench_native_grid.py 2>/dev/null | tail -1 | cut -c100-260 >> gpurun_out/g_ab.log
    echo -n "lib$lib one: " >> gpurun_out/g_ab.log
    GPSX_LIB=stm32f4_sdr_gps_amd/lib/libgpsx$lib.so python tools/bench_grid_kernel.py 1 1 50 2>/dev/null | tail -1 | cut -c1-200 >> gpurun_out/g_ab.log
  done
done
cat gpurun_out/g_ab.log
