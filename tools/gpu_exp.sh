for ex in ${EXPS:-0 4}; do
  GPSX_MX_EXPERIMENT=$ex python tools/bench_grid_kernel.py ${ARGS:-64 1 20} 2>/dev/null | tail -1
done
