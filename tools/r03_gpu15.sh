for v in "" NO_EPI NO_PASS NONE; do
  L=stm32f4_sdr_gps_amd/lib/libgpsx${v:+_$v}.so
  echo -n "$v: "; GPSX_LIB=$L python tools/bench_native_grid.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_launch'])"
done
