set -x
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "track" 2>&1 | tail -15 > gpurun_out/r03a_track_tests.log
cat gpurun_out/r03a_track_tests.log
python tools/bench_track_kernel.py 2048 4096 16384 65536 212992 > gpurun_out/r03a_trackkernel.json 2>&1
cat gpurun_out/r03a_trackkernel.json
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r03a_gputests.log
cat gpurun_out/r03a_gputests.log
