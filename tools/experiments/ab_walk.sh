for i in 1 2 3; do
  for lib in "" _b; do
    echo -n "lib$lib: "
    GPSX_LIB=stm32f4_sdr_gps_amd/lib/libgpsx$lib.so python tools/bench_grid_kernel.py 256 10 5 2>/dev/null | tail -1 | cut -c60-200
  done
done
timeout 900 python -m pytest tests/test_gpu_walk_form.py tests/test_gpu_parity.py -q -x -k "walk or multi_block or 10ms" 2>&1 | tail -3
