# On the GPU box: the whole validation + measurement pass of a round under tag $1 (default r04); tools/collect_profiles.sh <tag>
# then condenses gpurun_out/ into profiles/.  `bash tools/gpu_validate.sh ab [args]` instead: a same-box A/B of the library as
# built against stm32f4_sdr_gps_amd/lib/libgpsx_b.so (tools/build_variant.sh), alternating, three times each.
if [ "${1:-}" = "ab" ]; then
  shift
  for i in 1 2 3; do
    for lib in _lab _b; do   # (A = the working tree's lab build: same kernels as the product, reads the knobs; B = tools/build_variant.sh)
      echo -n "lib$lib: "
      GPSX_LIB=stm32f4_sdr_gps_amd/lib/libgpsx$lib.so python tools/bench_grid_kernel.py ${@:-256 1 20} 2>/dev/null | tail -1 | cut -c60-200
    done
  done
  exit 0
fi
T=${1:-r04}
set -x
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror|assert" | head -20 > gpurun_out/${T}_gputests.log
cat gpurun_out/${T}_gputests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -c 300 gpurun_out/${T}_bench.err
python bench.py --searches 64 --no-cpu-baseline --no-tracking > gpurun_out/${T}_bench_64.json 2>> gpurun_out/${T}_bench.err
python bench.py --n-ms 10 --no-cpu-baseline --no-tracking --steps 5 > gpurun_out/${T}_bench_10ms.json 2>> gpurun_out/${T}_bench.err
python bench.py --amp-scale 1.0 --no-cpu-baseline --no-tracking --steps 10 > gpurun_out/${T}_bench_strong.json 2>> gpurun_out/${T}_bench.err
bash tools/profile_bench.sh ${T} > gpurun_out/${T}_prof.log 2>&1
BENCH_ARGS="--n-ms 10" bash tools/profile_bench.sh ${T}_10ms > gpurun_out/${T}_prof10.log 2>&1
bash tools/profile_track.sh ${T}_track > gpurun_out/${T}_proft.log 2>&1
TRACK_CMD="python $GRAFT_REPO_ROOT/tools/bench_track_loop_kernel.py 20 212992" bash tools/profile_track.sh ${T}_track_loop > gpurun_out/${T}_proftl.log 2>&1
python tools/bench_track_kernel.py 2048 16384 65536 212992 688128 > gpurun_out/${T}_track_kernel_us.json 2>&1
python tools/bench_track_loop_kernel.py 20 256 16384 65536 212992 1048576 > gpurun_out/${T}_track_loop_kernel_us.json 2>&1
python tools/bench_tracking.py > gpurun_out/${T}_tracking_latency.json 2>/dev/null
# launch-size sweep: 32 PRN x 21 Doppler x 16368 phases per capture, captures resident in HBM, both grid kernels
( for n in 1 2 4 8 13 16 32 64 128 256; do for a in mx poly; do GPSX_ACQ_ALGO=$a python tools/bench_grid_kernel.py $n 1 10 2>/dev/null | tail -1; done; done
  for n in 1 4 16; do for a in mx poly; do GPSX_ACQ_ALGO=$a python tools/bench_grid_kernel.py $n 10 5 2>/dev/null | tail -1; done; done ) > gpurun_out/${T}_sweep.txt 2>&1
python tools/bench_native_grid.py 2>/dev/null | tail -1 > gpurun_out/${T}_native.json
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${T}_native/trace -o trace -- python $GRAFT_REPO_ROOT/tools/bench_native_grid.py > $GRAFT_REPO_ROOT/gpurun_out/${T}_native_prof.log 2>&1 )
PMC_JSON=$GRAFT_REPO_ROOT/gpurun_out/${T}_native_pmc_summary.json PMC_CMD="python $GRAFT_REPO_ROOT/tools/bench_native_grid.py --steps 3 --warmup 1" bash tools/pmc_quick.sh "k_acq_mx<4>" > gpurun_out/${T}_native_pmc.log 2>&1
python tools/pcie_probe.py > gpurun_out/${T}_pcie_probe.txt 2>&1
python - "$T" <<'PY'
import json, sys
T = sys.argv[1]
for f in ("bench", "bench_64", "bench_10ms", "bench_strong"):
    d = json.loads(open(f"gpurun_out/{T}_{f}.json").read().strip().splitlines()[-1])
    cl = (d.get("tracking") or {}).get("closed_loop") or {}
    print(f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d.get("pcie_inclusive", {}).get("value"), (d.get("tracking") or {}).get("value"), cl.get("value"), (cl.get("device_loop") or {}).get("value"))
PY
