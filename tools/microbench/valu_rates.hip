// Instruction-rate microbenchmark for the integer VALU ops the correlator kernels could be built on.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates ; run on the GPU box.
// Reports wave-instructions/clk/CU-equivalent as "Glane-ops/s" (64 lanes per wave-instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned int u32;
typedef unsigned long long u64;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

constexpr int ITERS = 4096;
constexpr int NACC = 8;

template <int OP>
__global__ __launch_bounds__(256) void k_rate(u32* out, u32 seed_a, u32 seed_b)
{
  u32 a[NACC], acc[NACC];
  u64 wide[NACC];
  u64 acc64[NACC];
  u32x4 acc128[NACC];
  u32 tid = threadIdx.x + blockIdx.x * blockDim.x;
  for (int i = 0; i < NACC; i++) {
    a[i] = seed_a * (tid + i * 7919u) + 12345u;
    acc[i] = i;
    wide[i] = ((u64)a[i] << 32) | (a[i] ^ 0x5a5a5a5au);
    acc64[i] = i;
    acc128[i] = u32x4{0, 1, 2, 3};
  }
  u32 b = seed_b;
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      if (OP == 0) { acc[i] = __builtin_popcount(a[i] ^ acc[i]) + acc[i]; }              // xor + bcnt(dependent)
      if (OP == 1) { acc[i] = __builtin_amdgcn_sad_u8(a[i], b, acc[i]); }
      if (OP == 2) { acc[i] = __builtin_amdgcn_msad_u8(a[i], b, acc[i]); }
      if (OP == 3) { acc64[i] = __builtin_amdgcn_qsad_pk_u16_u8(wide[i], b, acc64[i]); }
      if (OP == 4) { acc128[i] = __builtin_amdgcn_mqsad_u32_u8(wide[i], b, acc128[i]); }
      if (OP == 5) { acc[i] = __builtin_amdgcn_sdot4((int)a[i], (int)b, (int)acc[i], false); }
      if (OP == 6) { acc[i] = __builtin_amdgcn_sdot8((int)a[i], (int)b, (int)acc[i], false); }
      if (OP == 7) { acc[i] = __builtin_amdgcn_alignbit(a[i], acc[i], b); }
      if (OP == 8) { acc[i] = __builtin_amdgcn_alignbyte(a[i], acc[i], b); }
      if (OP == 9) { acc[i] = __builtin_amdgcn_perm(a[i], acc[i], b); }
      if (OP == 10) { acc[i] = __builtin_popcount(a[i]) + acc[i]; a[i] ^= b; }           // bcnt + xor independent pair
      if (OP == 11) { acc64[i] = __builtin_amdgcn_mqsad_pk_u16_u8(wide[i], b, acc64[i]); }
      if (OP == 12) { acc[i] = __builtin_amdgcn_udot4(a[i], b, acc[i], false); }
      if (OP == 13) { acc[i] = __builtin_amdgcn_udot8(a[i], b, acc[i], false); }
      if (OP == 14) { acc[i] = acc[i] * a[i] + b; }                                          // v_mad_u32_u24 / mul_lo
      if (OP == 15) { acc[i] = __builtin_amdgcn_sad_u16(a[i], b, acc[i]); }
      if (OP == 16) { acc[i] = (acc[i] + a[i]) ^ b; }                                        // add + xor (2 ops) or v_add3/xad
      // round 2: exact instructions by inline asm, to price the grid kernels' instruction mix (which ops are 2-cycle?)
      if (OP == 17) { u32 t; asm volatile("v_and_b32 %0, %1, %2" : "=v"(t) : "s"(b), "v"(a[i]));
                      asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(t)); }   // the polyphase main-loop pair
      if (OP == 18) { asm volatile("v_and_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 19) { asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 20) { asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 21) { asm volatile("v_mad_i32_i24 %0, %1, -2, %0" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 22) { asm volatile("v_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 23) { asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 24) { asm volatile("v_sqrt_f32 %0, %0" : "+v"(acc[i])); }
      if (OP == 25) { asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(acc[i])); }
      if (OP == 26) { asm volatile("v_max_u32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 27) { asm volatile("v_lshl_or_b32 %0, %0, 11, %1" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 28) { asm volatile("v_add3_u32 %0, %1, %1, %0" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 29) { asm volatile("v_bfe_u32 %0, %0, %1, 8" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 30) { asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 31) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 32) { asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(acc[i])); }
      if (OP == 33) { asm volatile("v_sub_u32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));
                      asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i])); }  // simple op beside bcnt
      if (OP == 34) { asm volatile("v_alignbit_b32 %0, %1, %0, %2" : "+v"(acc[i]) : "v"(a[i]), "s"(b)); }  // SGPR shift
      if (OP == 35) { asm volatile("v_max_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i])); }
      if (OP == 36) { asm volatile("v_pk_add_u16 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i])); }
    }
    b += 0x01010101u;
  }
  u32 r = 0;
  for (int i = 0; i < NACC; i++) {
    r ^= acc[i] ^ (u32)acc64[i] ^ (u32)(acc64[i] >> 32) ^ acc128[i].x ^ acc128[i].y ^ acc128[i].z ^ acc128[i].w ^ a[i];
  }
  out[tid] = r;
}

template <int OP>
int run(const char* name, int ops_per_stmt, u32* d_out)
{
  const int blocks = 256 * 8, threads = 256;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 0x9e3779b9u, 0x01020304u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 5;
  for (int r = 0; r < reps; r++)
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(threads), 0, 0, d_out, 0x9e3779b9u, 0x01020304u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  double stmts = (double)reps * blocks * threads * (double)ITERS * NACC;
  double per_s = stmts / (ms * 1e-3);
  printf("%-28s %8.3f ms  %9.2f Gstmt-lanes/s  (x%d ops => %9.2f Glane-ops/s; frac of 78.6T = %.3f per stmt)\n",
         name, ms / reps, per_s * 1e-9, ops_per_stmt, per_s * ops_per_stmt * 1e-9, per_s / 78.6e12);
  return 0;
}

int main()
{
  u32* d_out; CHECK(hipMalloc(&d_out, 256 * 8 * 256 * 4));
  hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  run<0>("xor+bcnt (dep chain)", 2, d_out);
  run<10>("bcnt+xor (indep)", 2, d_out);
  run<1>("v_sad_u8", 1, d_out);
  run<2>("v_msad_u8", 1, d_out);
  run<15>("v_sad_u16", 1, d_out);
  run<3>("v_qsad_pk_u16_u8", 1, d_out);
  run<11>("v_mqsad_pk_u16_u8", 1, d_out);
  run<4>("v_mqsad_u32_u8", 1, d_out);
  run<5>("v_dot4_i32_i8", 1, d_out);
  run<6>("v_dot8_i32_i4", 1, d_out);
  run<12>("v_dot4_u32_u8", 1, d_out);
  run<13>("v_dot8_u32_u4", 1, d_out);
  run<7>("v_alignbit_b32", 1, d_out);
  run<8>("v_alignbyte_b32", 1, d_out);
  run<9>("v_perm_b32", 1, d_out);
  run<14>("mul+add u32", 1, d_out);
  run<16>("add+xor", 2, d_out);
  run<17>("v_and(sgpr)+v_bcnt(acc)", 2, d_out);
  run<18>("v_and_b32", 1, d_out);
  run<19>("v_bcnt_u32_b32 (acc)", 1, d_out);
  run<33>("v_sub_u32+v_bcnt", 2, d_out);
  run<31>("v_add_u32", 1, d_out);
  run<20>("v_cndmask_b32 (vcc)", 1, d_out);
  run<21>("v_mad_i32_i24", 1, d_out);
  run<28>("v_add3_u32", 1, d_out);
  run<27>("v_lshl_or_b32", 1, d_out);
  run<29>("v_bfe_u32", 1, d_out);
  run<26>("v_max_u32", 1, d_out);
  run<34>("v_alignbit_b32 (sgpr shift)", 1, d_out);
  run<30>("v_mov_b32_dpp row_shr", 1, d_out);
  run<22>("v_mul_f32", 1, d_out);
  run<23>("v_fma_f32", 1, d_out);
  run<35>("v_max_f32", 1, d_out);
  run<24>("v_sqrt_f32", 1, d_out);
  run<25>("v_cvt_f32_i32", 1, d_out);
  run<32>("v_cvt_u32_f32", 1, d_out);
  run<36>("v_pk_add_u16", 1, d_out);
  return 0;
}
