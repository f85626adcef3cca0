set -x
mkdir -p gpurun_out
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; nproc
python -m pytest tests -m gpu -x -q -k "byte_phases or randomised or alternative or windows_sets or kernels_do_not_write or full_cold" 2>&1 | tail -5 > gpurun_out/r03d_tests.log
cat gpurun_out/r03d_tests.log
python tools/bench_native_grid.py 2>/dev/null | tail -1 > gpurun_out/r03d_native.json
cat gpurun_out/r03d_native.json
python tools/bench_tracking_closed_loop.py --channels 256 16384 65536 131072 196608 262144 --ms 1200 --signals 32 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print({k: d[k] for k in ('channels','host_workers','p50_us','p99_us','max_us','steps_over_1ms','warmup_max_us','real_time','code_and_carrier_lock')})
" | tee gpurun_out/r03d_closed_loop.txt
