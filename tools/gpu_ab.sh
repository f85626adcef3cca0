# same-box A/B: the library as built vs libgpsx_b.so (tools/build_variant.sh), alternating
for i in 1 2 3; do
  python tools/bench_grid_kernel.py ${ARGS:-256 1 20} 2>/dev/null | tail -1 | cut -c60-150
  GPSX_LIB=stm32f4_sdr_gps_amd/lib/libgpsx_b.so python tools/bench_grid_kernel.py ${ARGS:-256 1 20} 2>/dev/null | tail -1 | cut -c60-150
done
