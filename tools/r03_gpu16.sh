python -m pytest tests -m gpu -x -q -k "byte_phases or randomised or kernels_do_not" 2>&1 | grep -E "passed|failed|rror|assert" | head
for v in "" NO_PASS; do
  L=stm32f4_sdr_gps_amd/lib/libgpsx${v:+_$v}.so
  echo -n "v=$v: "; GPSX_LIB=$L python tools/bench_native_grid.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_launch'])"
done
python tools/bench_native_grid.py --searches 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('one capture', d['ms_per_launch'])"
