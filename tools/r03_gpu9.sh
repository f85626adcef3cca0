mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error|assert" | head -20 | tee gpurun_out/r03i_gputests.log
for n in 1 2 4 6 8; do
  python tools/bench_grid_kernel.py $n 1 20 2>/dev/null | tail -1
  GPSX_ACQ_NO_SPLIT=1 python tools/bench_grid_kernel.py $n 1 20 2>/dev/null | tail -1
done | tee gpurun_out/r03i_small_launches.txt
