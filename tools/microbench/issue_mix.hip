// How do an MFMA stream and a VALU stream share one SIMD on gfx950?  k_acq_mx keeps two waves per SIMD: one issues the FP4
// MFMAs of a pass, the other the clip/square/root/search instructions of the previous pass.  This micro-benchmark times both
// streams (s_memtime, shader cycles), alone and together, for VALU streams of different instruction classes and different
// distances between dependent instructions.
// Build: hipcc --offload-arch=gfx950 -O3 issue_mix.hip -o issue_mix ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned int u32;
typedef unsigned long long u64;
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));

constexpr int kMfmaPerIter = 32;    // 8 accumulators x 4

// VALU stream: kValuPerIter instructions per iteration over ILP independent registers (dependent distance = ILP)
// OP 0: v_mul_f32 (2-cycle class)  1: v_max_i32 (4-cycle class)  2: v_sqrt_f32  3: the epilogue's mix per hypothesis
template <int OP, int ILP, int kValuPerIter>
__device__ __forceinline__ void valu_iter(float (&x)[8], float c)
{
  float2 cp = make_float2(c, c);
  asm volatile("" : "+v"(cp));
#pragma unroll
  for (int i = 0; i < kValuPerIter; i++) {
    float &r = x[i % ILP];
    if (OP == 0)
      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r) : "v"(c));
    else if (OP == 1)
      asm volatile("v_max_i32 %0, %0, %1" : "+v"(r) : "v"(c));
    else if (OP == 2)
      asm volatile("v_sqrt_f32 %0, %0" : "+v"(r));
    else if (OP == 3)
      asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r) : "v"(c));
    else if (OP == 4) {   // packed f32: two values per lane and instruction (register pairs x[0:1], x[2:3], ..)
      float2 &rp = reinterpret_cast<float2 *>(x)[(i % ILP) % 4];
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(rp) : "v"(cp));
    } else if (OP == 5) {
      float2 &rp = reinterpret_cast<float2 *>(x)[(i % ILP) % 4];
      asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(rp) : "v"(cp));
    } else if (OP == 6) {
      float2 &rp = reinterpret_cast<float2 *>(x)[(i % ILP) % 4];
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(rp) : "v"(cp));
    } else if (OP == 7)
      asm volatile("v_mul_f32_e64 %0, %0, |%1| clamp" : "+v"(r) : "v"(c));
    else if (OP == 8)
      asm volatile("v_max3_u32 %0, %0, %1, %1" : "+v"(r) : "v"(c));
    else if (OP == 9)
      asm volatile("v_pk_add_u16 %0, %0, %1 clamp" : "+v"(r) : "v"(c));
    else if (OP == 10)
      asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(r) : "v"(c));
    else if (OP == 11)
      asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(r) : "v"(c));
    else if (OP == 12)
      asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(r) : "v"(c));
    else if (OP == 13)
      asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(r) : "v"(c));
    else if (OP == 14)   // (distance >= 2: the dot instructions need a wait state before a dependent read)
      asm volatile("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(r) : "v"(c));
    else if (OP == 15)
      asm volatile("v_dot2c_f32_f16 %0, %1, %1" : "+v"(r) : "v"(c));
  }
}

template <int OP, int ILP, int VPI, bool WITH_MFMA, bool WITH_VALU>
__global__ __launch_bounds__(512, 1) void k_mix(u32 *out, int iters, float c)
{
  __shared__ u32 pad[30000];   // > 80 KB: one workgroup per CU
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  pad[tid] = tid;
  __syncthreads();
  const int role = wave >> 2;
  u64 t0 = 0, t1 = 0;
  float sink = 0.f;
  if (role == 0) {
    if (WITH_MFMA) {
      v16f acc[8];
      for (int a = 0; a < 8; a++)
        for (int r = 0; r < 16; r++)
          acc[a][r] = 0.f;
      v8i fa = {(int)pad[lane] * 0x11111111, 0x2a2a2a2a, 0x02a0020a, (int)pad[lane + 64], 0, 0, 0, 0};
      v8i fb = {0x2a2a2a2a, (int)pad[lane + 3] * 0x01010101, 0x02a0020a, 0x22222222, 0, 0, 0, 0};
      asm volatile("" : "+v"(fa), "+v"(fb));
      t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
      for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < kMfmaPerIter; k++)
          acc[k & 7] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb, acc[k & 7], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
      asm volatile("s_nop 0" ::: "memory");
      for (int a = 0; a < 8; a++)
        for (int r = 0; r < 16; r++)
          sink += acc[a][r];
      asm volatile("" : "+v"(sink));
      t1 = __builtin_amdgcn_s_memtime();
    }
  } else {
    if (WITH_VALU) {
      float x[8];
      for (int i = 0; i < 8; i++)
        x[i] = 1.0f + 0.001f * (float)(lane + i);
      t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
      for (int it = 0; it < iters; it++)
        valu_iter<OP, ILP, VPI>(x, c);
      for (int i = 0; i < 8; i++)
        sink += x[i];
      asm volatile("" : "+v"(sink));
      t1 = __builtin_amdgcn_s_memtime();
    }
  }
  if (lane == 0) {
    out[(blockIdx.x * 8 + wave) * 2] = (u32)(t1 - t0);
    out[(blockIdx.x * 8 + wave) * 2 + 1] = __float_as_uint(sink);
  }
}

template <int OP, int ILP, int VPI>
static int run(const char *name, int n_wg, int iters)
{
  u32 *d_out;
  CHECK(hipMalloc(&d_out, (size_t)n_wg * 16 * sizeof(u32)));
  std::vector<u32> h(n_wg * 16);
  auto mean = [&](int role, double per) {
    double s = 0;
    int cnt = 0;
    for (int b = 0; b < n_wg; b++)
      for (int w = 4 * role; w < 4 * role + 4; w++) {
        s += h[(b * 8 + w) * 2];
        cnt++;
      }
    return s / cnt / per;
  };
  printf("%-30s dep. distance %d, %d per MFMA:", name, ILP, VPI / kMfmaPerIter);
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL((k_mix<OP, ILP, VPI, false, true>), dim3(n_wg), dim3(512), 0, 0, d_out, iters, 1.0f);
    CHECK(hipDeviceSynchronize());
  }
  CHECK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
  printf("  alone %5.2f cyc/VALU,", mean(1, (double)iters * VPI));
  hipLaunchKernelGGL((k_mix<OP, ILP, VPI, true, false>), dim3(n_wg), dim3(512), 0, 0, d_out, iters, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
  printf(" %5.1f cyc/MFMA |", mean(0, (double)iters * kMfmaPerIter));
  hipLaunchKernelGGL((k_mix<OP, ILP, VPI, true, true>), dim3(n_wg), dim3(512), 0, 0, d_out, iters, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
  const double m = mean(0, (double)iters * kMfmaPerIter), v = mean(1, (double)iters * VPI);
  printf(" together %5.2f cyc/VALU (stream %6.0f cyc/iter), %5.1f cyc/MFMA (stream %6.0f cyc/iter)\n", v, v * VPI, m, m * kMfmaPerIter);
  CHECK(hipFree(d_out));
  return 0;
}

int main()
{
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d; one MFMA wave + one VALU wave per SIMD; an iteration = %d MFMAs | VPI VALU instructions\n", prop.gcnArchName,
         prop.multiProcessorCount, kMfmaPerIter);
  const int n_wg = 256, it = 200;
  run<0, 1, 192>("v_mul_f32 (2-cycle class)", n_wg, it);
  run<0, 2, 192>("v_mul_f32 (2-cycle class)", n_wg, it);
  run<0, 4, 192>("v_mul_f32 (2-cycle class)", n_wg, it);
  run<0, 8, 192>("v_mul_f32 (2-cycle class)", n_wg, it);
  run<0, 8, 384>("v_mul_f32 (2-cycle class)", n_wg, it);
  run<1, 1, 192>("v_max_i32 (4-cycle class)", n_wg, it);
  run<1, 2, 192>("v_max_i32 (4-cycle class)", n_wg, it);
  run<1, 4, 192>("v_max_i32 (4-cycle class)", n_wg, it);
  run<1, 8, 192>("v_max_i32 (4-cycle class)", n_wg, it);
  run<1, 8, 64>("v_max_i32 (4-cycle class)", n_wg, it);
  run<1, 8, 128>("v_max_i32 (4-cycle class)", n_wg, it);
  run<1, 8, 256>("v_max_i32 (4-cycle class)", n_wg, it);
  run<2, 1, 64>("v_sqrt_f32", n_wg, it);
  run<2, 8, 64>("v_sqrt_f32", n_wg, it);
  run<3, 1, 192>("v_lshl_or_b32 (VOP3)", n_wg, it);
  run<3, 8, 192>("v_lshl_or_b32 (VOP3)", n_wg, it);
  run<4, 4, 192>("v_pk_mul_f32 (2 values/lane)", n_wg, it);
  run<5, 4, 192>("v_pk_fma_f32 (2 values/lane)", n_wg, it);
  run<6, 4, 192>("v_pk_add_f32 (2 values/lane)", n_wg, it);
  run<6, 4, 96>("v_pk_add_f32 (2 values/lane)", n_wg, it);
  run<7, 8, 192>("v_mul_f32 |x| clamp (VOP3)", n_wg, it);
  run<8, 8, 192>("v_max3_u32", n_wg, it);
  run<9, 8, 192>("v_pk_add_u16 clamp", n_wg, it);
  run<10, 8, 192>("v_perm_b32", n_wg, it);
  run<11, 8, 192>("v_or3_b32", n_wg, it);
  run<12, 8, 192>("v_cvt_pkrtz_f16_f32", n_wg, it);
  run<13, 8, 192>("v_pk_max_f16", n_wg, it);
  run<14, 8, 192>("v_dot2_f32_f16", n_wg, it);
  run<15, 8, 192>("v_dot2c_f32_f16", n_wg, it);
  return 0;
}
