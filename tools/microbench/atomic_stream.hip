// atomic_stream.hip -- can the walk form's running sums move as fire-and-forget L2 atomics instead of load / add / store?
// Three kernels over the same footprint (every 8-byte record touched once per pass, lanes of a wave on consecutive records):
//   rmw     : v = x[i]; x[i] = v + a          (what k_acq_mx<3> does: a load whose latency the wave has to cover, a store)
//   atomic64: atomicAdd(&x[i], a), no return  (one 64-bit add per record at the L2)
//   atomic32: two 32-bit atomicAdds per record
// prints GB/s of read + write traffic equivalents (2 x footprint per pass).   hipcc --offload-arch=gfx950 -O3 atomic_stream.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__global__ void k_rmw(unsigned long long *x, size_t n, unsigned long long a)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    x[i] = x[i] + a;
}
__global__ void k_atomic64(unsigned long long *x, size_t n, unsigned long long a)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __hip_atomic_fetch_add(&x[i], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_atomic32(unsigned int *x, size_t n, unsigned int a)
{
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n / 2; i += (size_t)gridDim.x * blockDim.x) {
    __hip_atomic_fetch_add(&x[2 * i], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&x[2 * i + 1], a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main()
{
  const size_t mbs[] = {128, 268, 1024};
  for (size_t mb : mbs) {
    const size_t n = mb * (1 << 20) / 8;
    unsigned long long *x;
    if (hipMalloc(&x, n * 8) != hipSuccess) return 1;
    hipMemset(x, 0, n * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int which = 0; which < 3; which++) {
      const int reps = 20;
      for (int r = -2; r < reps; r++) {
        if (r == 0) hipEventRecord(e0);
        if (which == 0) hipLaunchKernelGGL(k_rmw, dim3(2048), dim3(256), 0, 0, x, n, 1ull);
        if (which == 1) hipLaunchKernelGGL(k_atomic64, dim3(2048), dim3(256), 0, 0, x, n, 1ull);
        if (which == 2) hipLaunchKernelGGL(k_atomic32, dim3(2048), dim3(256), 0, 0, (unsigned int *)x, 2 * n, 1u);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      ms /= reps;
      printf("%zu MB  %-8s %.4f ms per pass  %.2f TB/s (read + write equivalents)\n", mb, which == 0 ? "rmw" : which == 1 ? "atomic64" : "atomic32",
             ms, 2.0 * n * 8 / (ms * 1e-3) / 1e12);
    }
    hipFree(x);
  }
  return 0;
}
