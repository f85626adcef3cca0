# builds stm32f4_sdr_gps_amd/lib/libgpsx_b.so -- a LAB build (-DGPSX_LAB: the only kind that may carry timing ablations and
# instrumented variants, some of which give wrong results) -- from the k_acq_mx.hip of the working tree (-DMX_VARIANT_B is defined
# for an #ifdef'd alternative, if the source carries one) for same-box A/B timings -- boxes differ by +-3 %:
#   git stash; bash tools/build_variant.sh; git stash pop; make -C stm32f4_sdr_gps_amd/csrc      # B = HEAD, A = working tree
#   bash tools/gpu_validate.sh ab   (on the GPU box: alternates A and B through $GPSX_LIB of tools/bench_grid_kernel.py)
# VARIANT_DEFS: the variant's own -D flags (default -DGPSX_MX_ABLATIONS: $GPSX_MX_EXPERIMENT is then read per launch)
set -e
cd "$(dirname "$0")/../stm32f4_sdr_gps_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -ffp-contract=off -fno-fast-math -Wno-unused-function -DGPSX_LAB"
DEFS="${VARIANT_DEFS:--DGPSX_MX_ABLATIONS}"
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -DMX_VARIANT_B $DEFS -c k_acq_mx.hip -o ../build/k_acq_mx_b.o
/opt/rocm/bin/hipcc $F $DEFS -c gpsx_api.hip -o ../build/gpsx_api_b.o
OBJS=$(ls ../build/*.o | grep -v k_acq_mx | grep -v gpsx_api | grep -v _b.o | grep -v _lab.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libgpsx_b.so $OBJS ../build/k_acq_mx_b.o ../build/gpsx_api_b.o -Wl,-rpath,/opt/rocm/lib -ldl
echo built ../lib/libgpsx_b.so
