set -x
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02a_gputests.log
python bench.py > gpurun_out/r02a_bench.json 2> gpurun_out/r02a_bench.err
tail -c 600 gpurun_out/r02a_bench.err
bash tools/profile_bench.sh r02a > gpurun_out/r02a_prof.log 2>&1
bash tools/profile_track.sh r02a_track > gpurun_out/r02a_proft.log 2>&1
python tools/bench_track_kernel.py > gpurun_out/r02a_trackkernel.json 2>&1
cat gpurun_out/r02a_gputests.log
