timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -k "byte_phase or native_grid or randomised" 2>&1 | tail -2
for i in 1 2 3; do
  for lib in "" _b; do echo -n "lib$lib: "; GPSX_LIB=stm32f4_sdr_gps_amd/lib/libgpsx$lib.so python tools/bench_native_grid.py 2>/dev/null | tail -1 | cut -c100-200; done
done
