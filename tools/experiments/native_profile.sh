set -x
R=$GRAFT_REPO_ROOT
python tools/bench_native_grid.py 2>/dev/null | tail -1 > gpurun_out/r04b_native.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04b_native/trace -o trace -- python $R/tools/bench_native_grid.py > $R/gpurun_out/r04b_native_prof.log 2>&1 )
PMC_JSON=$R/gpurun_out/r04b_native_pmc_summary.json PMC_CMD="python $R/tools/bench_native_grid.py --steps 3 --warmup 1" bash tools/pmc_quick.sh "k_acq_mx<4>" > gpurun_out/r04b_native_pmc.log 2>&1
tail -8 gpurun_out/r04b_native_pmc.log
cut -c1-260 gpurun_out/r04b_native.json
grep "k_acq_mx" gpurun_out/prof_r04b_native/trace/trace_kernel_stats.csv | head -3
