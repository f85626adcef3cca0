for n in 1 2 4 8 16 32 64 128 256; do
  for a in mx poly; do
    GPSX_ACQ_ALGO=$a python tools/bench_grid_kernel.py $n 1 10 2>/dev/null | tail -1
  done
done
for n in 1 4 16; do
  for a in mx poly; do
    GPSX_ACQ_ALGO=$a python tools/bench_grid_kernel.py $n 10 5 2>/dev/null | tail -1
  done
done
