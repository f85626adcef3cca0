#!/bin/bash
# The host layer under sanitizers: builds lib/libgpsx_asan.so (ASan), lib/libgpsx_ubsan.so (UBSan) and lib/libgpsx_tsan.so (TSan) from the
# product sources (csrc/Makefile `san`) and runs the host-side test suites against them with the sanitizer runtime preloaded into
# python.  No GPU needed for the default set; `gpu` as $1 adds the step / closed-loop GPU suites and the worker-pool soak (run
# that through gpurun).  Logs: gpurun_out/san_<kind>.log (+ the sanitizer's own report files san_<kind>.report.*); a summary line
# per kind is printed.  Copy the logs you want judged to profiles/.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
MODE=${1:-cpu}
mkdir -p gpurun_out
RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)")
CPU_TESTS="tests/test_abi_and_host.py tests/test_nav_master.py tests/test_pvt.py tests/test_ephemeris.py tests/test_libm_restatement.py"
GPU_TESTS="tests/test_gpu_steps.py tests/test_gpu_pvt_chain.py tests/test_gpu_track_mux.py"
for KIND in ${SAN_KINDS:-asan ubsan tsan}; do
  make -C stm32f4_sdr_gps_amd/csrc san SAN=$KIND -j4 > gpurun_out/san_${KIND}_build.log 2>&1 || { echo "$KIND: BUILD FAILED"; tail -5 gpurun_out/san_${KIND}_build.log; continue; }
  LOG=gpurun_out/san_$KIND.log
  if [ $KIND = asan ] && [ "$MODE" = gpu ]; then
    # ASan cannot share a process with the HIP runtime here: its HSA interceptors abort on the runtime's first pool allocation
    # (and with allocator_may_return_null the runtime itself segfaults) -- ASan covers the host-only suites (cpu mode)
    echo "asan: skipped in gpu mode (the ASan runtime's HSA interceptors do not survive this HIP runtime); run the cpu mode"; continue
  fi
  rm -f gpurun_out/san_$KIND.report.*
  if [ $KIND = asan ]; then
    PRE="$RT/libclang_rt.asan-x86_64.so"
    export ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=0:log_path=$ROOT/gpurun_out/san_asan.report:protect_shadow_gap=0"
  elif [ $KIND = ubsan ]; then
    PRE="$RT/libclang_rt.ubsan_standalone-x86_64.so"
    export UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0:log_path=$ROOT/gpurun_out/san_ubsan.report"
  else
    PRE="$RT/libclang_rt.tsan-x86_64.so"
    export TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:log_path=$ROOT/gpurun_out/san_tsan.report:ignore_noninstrumented_modules=1"
  fi
  ( echo "== $KIND: $(date -u +%FT%TZ)  library lib/libgpsx_$KIND.so  preload $PRE"
    if [ "$MODE" = gpu ]; then
      # (one test starts a python of its own that dlopens the library WITHOUT the preload: the TSan / ASan runtimes cannot be
      #  loaded late -- "cannot allocate memory in static TLS block" -- so it is left to the UBSan and plain runs)
      SKIP=""; [ $KIND != ubsan ] && SKIP="--deselect tests/test_gpu_steps.py::test_batched_step_under_an_overridden_time_source_calls_it_once_from_the_calling_thread"
      GPSX_LIB_PATH=$ROOT/stm32f4_sdr_gps_amd/lib/libgpsx_$KIND.so LD_PRELOAD=$PRE python -m pytest $CPU_TESTS $GPU_TESTS $SKIP -q -p no:cacheprovider 2>&1 | tail -15
      echo "== $KIND: worker-pool soak (tools/soak_step_pool.py 20000 2048)"
      GPSX_LIB_PATH=$ROOT/stm32f4_sdr_gps_amd/lib/libgpsx_$KIND.so LD_PRELOAD=$PRE timeout 600 python tools/soak_step_pool.py 20000 2048 2>&1 | tail -5
    else
      GPSX_LIB_PATH=$ROOT/stm32f4_sdr_gps_amd/lib/libgpsx_$KIND.so LD_PRELOAD=$PRE python -m pytest $CPU_TESTS -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
    fi ) > $LOG 2>&1
  N=$(ls gpurun_out/san_$KIND.report.* 2>/dev/null | wc -l)
  echo "$KIND: $(grep -E 'passed|failed|error' $LOG | tail -1)  sanitizer report files: $N"
  for f in gpurun_out/san_$KIND.report.*; do [ -f "$f" ] && { echo "--- $f"; head -40 "$f"; }; done
done
