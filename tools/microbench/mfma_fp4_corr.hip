// Feasibility experiment: the polyphase bit-plane correlation
//      X(q) = sum_c chip_p[c] * e[(q + c) mod 1023]        (32 PRNs p, 1023 chip offsets q, 1023 chips c)
// as a Toeplitz GEMM on the matrix cores -- MX-FP4 operands (0, +-1 exactly representable, E8M0 scale 2^0), f32
// accumulation (exact for integers < 2^24) -- v_mfma_scale_f32_32x32x64_f8f6f4:
//   A (M x K, rows = PRN)  = chips as FP4 nibbles, 16 bytes per lane from an LDS table
//   B (K x N, cols = q)    = Toeplitz view of ONE nibble vector in LDS: lane (n, h) reads the 32 nibbles starting at
//                            q + 64 kappa + 32 h -- an UNALIGNED 16-byte LDS read from one of two copies (even / odd start)
//   fragment reuse: tile (Q, kappa) reads the same bytes as (Q - 2, kappa + 1): a wave owning q-tiles Q0, Q0+2, Q0+4, Q0+6
//   walks anti-diagonals f = Q + 2 kappa, 19 fragment loads per 64 MFMAs.
// Part 1 checks operand / result layouts and exactness against the CPU; part 2 times the MFMA part of one bench step
// (1344 (search, Doppler) pairs x 17 passes x 2 streams) with 8 waves per workgroup.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_fp4_corr.hip -o mfma_fp4_corr
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned int u32;

constexpr int kVecBytes = 1024 + 32;      // one nibble vector copy: 2048 nibbles + slack for the last unaligned read
constexpr int kWaves = 8;
constexpr int kTiles = 4;                 // q-tiles per wave
constexpr u32 kScaleOne = 0x7F7F7F7Fu;    // E8M0 127 = 2^0 in every byte

struct Lds {
  uint8_t e[2][2][kVecBytes];             // [stream][copy (even / odd nibble start)][bytes]
  v4i chips[16][2][32];                   // [kappa][h][PRN]: 32 FP4 nibbles = chips 64 kappa + 32 h ..
  u32 e8[2][8][264];                      // MODE 1 / 2: eight copies, copy c starts at nibble c: every lane's window is dword aligned
};
// B-fragment load variants timed in part 2:
//   0  two copies, one byte-unaligned 16-byte read per lane          1  eight copies, one dword-aligned 16-byte read
//   2  eight copies, two ds_read2_b32 (dword-aligned by construction)  3  no data reads at all (MFMA-only ceiling of the loop)

__device__ __forceinline__ v8i widen(v4i x) { return v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0}; }

__device__ __forceinline__ v4i lds_read16_unaligned(const uint8_t *p)
{
  v4i v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

// One pass: acc[stream][tile] += chips x Toeplitz(e[stream]).  scale_b: E8M0 scale of the data operand.
template <int MODE>
__device__ __forceinline__ void toeplitz_pass(const Lds &L, int lane, int q0_tile, v16f (&acc)[2][kTiles], u32 scale_b)
{
  const int n = lane & 31, h = lane >> 5;
  const u32 *wi = &L.e8[0][n & 7][4 * (q0_tile + h) + (n >> 3)];
  const u32 *wq = &L.e8[1][n & 7][4 * (q0_tile + h) + (n >> 3)];
  v4i keep_i = *reinterpret_cast<const v4i *>(&L.e8[0][0][4 * lane]), keep_q = *reinterpret_cast<const v4i *>(&L.e8[1][0][4 * lane]);
  const uint8_t *bi = &L.e[0][n & 1][16 * (q0_tile + h) + (n >> 1)];
  const uint8_t *bq = &L.e[1][n & 1][16 * (q0_tile + h) + (n >> 1)];
  const v4i *ca = &L.chips[0][h][n];
  v4i a[16];
#pragma unroll
  for (int s = 0; s < 16 + kTiles - 1; s++) {
    if (s < 16)
      a[s] = ca[s * 64];                                  // chips[s][h][n]
    v4i fi, fq;
    if (MODE == 0) {
      fi = lds_read16_unaligned(bi + 32 * s);     // fragment f = Q0 + 2 s
      fq = lds_read16_unaligned(bq + 32 * s);
    } else if (MODE == 1) {
      fi = lds_read16_unaligned(reinterpret_cast<const uint8_t *>(wi + 8 * s));
      fq = lds_read16_unaligned(reinterpret_cast<const uint8_t *>(wq + 8 * s));
    } else if (MODE == 2) {
      fi = v4i{(int)wi[8 * s], (int)wi[8 * s + 1], (int)wi[8 * s + 2], (int)wi[8 * s + 3]};
      fq = v4i{(int)wq[8 * s], (int)wq[8 * s + 1], (int)wq[8 * s + 2], (int)wq[8 * s + 3]};
    } else {
      fi = keep_i;
      fq = keep_q;
      asm volatile("" : "+v"(fi), "+v"(fq));
    }
#pragma unroll
    for (int j = 0; j < kTiles; j++) {
      const int kappa = s - j;
      if (kappa < 0 || kappa >= 16)
        continue;
      acc[0][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(a[kappa]), widen(fi), acc[0][j], 4, 4, 0, kScaleOne, 0, scale_b);
      acc[1][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(widen(a[kappa]), widen(fq), acc[1][j], 4, 4, 0, kScaleOne, 0, scale_b);
    }
  }
}

// vec[stream][2048 nibbles] (global, one nibble per byte, values 0..15 = FP4 codes) -> the two LDS copies
__device__ void stage_vectors(Lds &L, const uint8_t *vec, int tid, int nthreads)
{
  for (int i = tid; i < 2 * 2 * 1024; i += nthreads) {
    const int st = i >> 11, copy = (i >> 10) & 1, byte = i & 1023;
    const uint8_t *v = vec + st * 2048;
    const int k0 = 2 * byte + copy, k1 = k0 + 1;
    const uint8_t lo = k0 < 2048 ? v[k0] : 0, hi = k1 < 2048 ? v[k1] : 0;
    L.e[st][copy][byte] = (uint8_t)(lo | (hi << 4));
  }
  for (int i = tid; i < 2 * 2 * 32; i += nthreads) {
    const int st = i >> 6, copy = (i >> 5) & 1, byte = 1024 + (i & 31);
    L.e[st][copy][byte] = 0;
  }
}

// chipnib[PRN][1024] (one FP4 code per byte) -> L.chips
__device__ void stage_chips(Lds &L, const uint8_t *chipnib, int tid, int nthreads)
{
  for (int i = tid; i < 16 * 2 * 32 * 4; i += nthreads) {
    const int dw = i & 3, prn = (i >> 2) & 31, h = (i >> 7) & 1, kappa = i >> 8;
    const uint8_t *c = chipnib + prn * 1024 + 64 * kappa + 32 * h + 8 * dw;
    u32 w = 0;
    for (int e = 0; e < 8; e++)
      w |= (u32)(c[e] & 15) << (4 * e);
    reinterpret_cast<u32 *>(&L.chips[kappa][h][prn])[dw] = w;
  }
}

// Part 1: one workgroup, n_pass passes over the same vectors (scale alternates to test the x4 path), full result out.
__global__ __launch_bounds__(64 * kWaves, 1) void k_check(const uint8_t *vec, const uint8_t *chipnib, float *out /* [2][1024 q][32 prn] */,
                                                          int n_pass, u32 scale_b)
{
  __shared__ Lds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stage_vectors(L, vec, tid, blockDim.x);
  stage_chips(L, chipnib, tid, blockDim.x);
  __syncthreads();
  const int q0_tile = 8 * (wave >> 1) + (wave & 1);   // this wave owns q-tiles q0_tile + 2 j
  v16f acc[2][kTiles];
  for (int st = 0; st < 2; st++)
    for (int j = 0; j < kTiles; j++)
      for (int r = 0; r < 16; r++)
        acc[st][j][r] = 0.f;
  for (int p = 0; p < n_pass; p++)
    toeplitz_pass<0>(L, lane, q0_tile, acc, scale_b);
  const int n = lane & 31, h = lane >> 5;
  for (int st = 0; st < 2; st++)
    for (int j = 0; j < kTiles; j++)
      for (int r = 0; r < 16; r++) {
        const int q = 32 * (q0_tile + 2 * j) + n;
        const int prn = (r & 3) + 8 * (r >> 2) + 4 * h;
        out[(st * 1024 + q) * 32 + prn] = acc[st][j][r];
      }
}

// Part 2: throughput.  Every workgroup = one (search, Doppler) pair: n_pass passes, a barrier + a cheap LDS rewrite between
// passes (stands for the next sample offset's vectors), accumulators folded into one float per lane at the end.
template <int MODE>
__global__ __launch_bounds__(64 * kWaves, 1) void k_rate(const uint8_t *vec, const uint8_t *chipnib, float *out, int n_pass)
{
  __shared__ Lds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stage_vectors(L, vec, tid, blockDim.x);
  stage_chips(L, chipnib, tid, blockDim.x);
  for (int i = tid; i < 2 * 8 * 264; i += blockDim.x)
    (&L.e8[0][0][0])[i] = 0x02A0020Au * (u32)(i + 1);
  __syncthreads();
  const int q0_tile = 8 * (wave >> 1) + (wave & 1);
  v16f acc[2][kTiles];
  for (int st = 0; st < 2; st++)
    for (int j = 0; j < kTiles; j++)
      for (int r = 0; r < 16; r++)
        acc[st][j][r] = 0.f;
#pragma unroll 1
  for (int p = 0; p < n_pass; p++) {
    toeplitz_pass<MODE>(L, lane, q0_tile, acc, kScaleOne);
    __syncthreads();
    reinterpret_cast<u32 *>(&L.e[0][0][0])[tid] ^= (u32)p;   // someone changes the vectors between passes
    __syncthreads();
  }
  float s = 0.f;
  for (int st = 0; st < 2; st++)
    for (int j = 0; j < kTiles; j++)
      for (int r = 0; r < 16; r++)
        s += acc[st][j][r];
  out[(size_t)blockIdx.x * blockDim.x + tid] = s;
}

static float fp4_value(int code)
{
  static const float mag[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
  return (code & 8) ? -mag[code & 7] : mag[code & 7];
}

int main()
{
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d clock=%d kHz, LDS struct %zu bytes\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, sizeof(Lds));
  // data: e vectors with values in {0, +1, -1} (codes 0, 2, 10) extended circularly (period 1023); chips 0/1 (codes 0 / 2)
  std::vector<uint8_t> vec(2 * 2048), chipnib(32 * 1024);
  srand(12345);
  for (int st = 0; st < 2; st++) {
    uint8_t base[1023];
    for (int k = 0; k < 1023; k++) {
      const int r = rand() % 4;
      base[k] = r == 0 ? 2 : (r == 1 ? 10 : 0);
    }
    for (int k = 0; k < 2048; k++)
      vec[st * 2048 + k] = base[k % 1023];
  }
  for (int p = 0; p < 32; p++)
    for (int c = 0; c < 1024; c++)
      chipnib[p * 1024 + c] = (c < 1023 && (rand() & 1)) ? 2 : 0;
  uint8_t *d_vec, *d_chips;
  float *d_out;
  CHECK(hipMalloc(&d_vec, vec.size()));
  CHECK(hipMalloc(&d_chips, chipnib.size()));
  CHECK(hipMalloc(&d_out, (size_t)4096 * 512 * 4));
  CHECK(hipMemcpy(d_vec, vec.data(), vec.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_chips, chipnib.data(), chipnib.size(), hipMemcpyHostToDevice));

  // ---- part 1: layout + exactness ------------------------------------------------------------------------------
  for (int variant = 0; variant < 3; variant++) {
    const int n_pass = variant == 0 ? 1 : 17;
    const u32 scale_b = variant == 2 ? 0x81818181u : kScaleOne;   // E8M0 129 = 2^2
    const float mult = (variant == 2 ? 4.f : 1.f) * n_pass;
    CHECK(hipMemset(d_out, 0, 2 * 1024 * 32 * 4));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64 * kWaves), 0, 0, d_vec, d_chips, d_out, n_pass, scale_b);
    CHECK(hipDeviceSynchronize());
    std::vector<float> out(2 * 1024 * 32);
    CHECK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    double maxabs = 0;
    for (int st = 0; st < 2; st++)
      for (int q = 0; q < 1024; q++)
        for (int p = 0; p < 32; p++) {
          double want = 0;
          for (int c = 0; c < 1024; c++)
            want += fp4_value(chipnib[p * 1024 + c]) * fp4_value(vec[st * 2048 + q + c]);
          want *= mult;
          const float got = out[(st * 1024 + q) * 32 + p];
          if (got != (float)want) {
            if (bad < 5)
              printf("  mismatch st %d q %d prn %d: got %g want %g\n", st, q, p, got, want);
            bad++;
          }
          if (fabs(want) > maxabs) maxabs = fabs(want);
        }
    printf("part 1 variant %d (%d passes, data scale %s): %ld mismatches of %d, max |value| %.0f\n", variant, n_pass,
           variant == 2 ? "2^2" : "2^0", bad, 2 * 1024 * 32, maxabs);
  }

  // ---- part 2: throughput ----------------------------------------------------------------------------------------
  auto time_mode = [&](int mode, int n_wg, int n_pass) {
    auto launch = [&]() {
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_rate<0>, dim3(n_wg), dim3(64 * kWaves), 0, 0, d_vec, d_chips, d_out, n_pass); break;
        case 1: hipLaunchKernelGGL(k_rate<1>, dim3(n_wg), dim3(64 * kWaves), 0, 0, d_vec, d_chips, d_out, n_pass); break;
        case 2: hipLaunchKernelGGL(k_rate<2>, dim3(n_wg), dim3(64 * kWaves), 0, 0, d_vec, d_chips, d_out, n_pass); break;
        default: hipLaunchKernelGGL(k_rate<3>, dim3(n_wg), dim3(64 * kWaves), 0, 0, d_vec, d_chips, d_out, n_pass); break;
      }
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    launch();
    CHECK(hipDeviceSynchronize());
    const int reps = 5;
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++)
      launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double mfmas = (double)n_wg * n_pass * 2 * 32 * 16;      // streams x q-tiles x k-steps
    const double flops = mfmas * 2.0 * 32 * 32 * 64;
    printf("part 2 mode %d: %5d workgroups x %2d passes: %.3f ms  -> %.1f TFLOP/s (FP4 dense peak ~10000), %.1f cycles per MFMA per SIMD\n",
           mode, n_wg, n_pass, ms, flops / (ms * 1e-3) / 1e12,
           ms * 1e-3 * prop.clockRate * 1e3 / (mfmas / (prop.multiProcessorCount * 4.0)));
  };
  for (int mode = 0; mode < 4; mode++)
    for (int n_wg : {256, 1344})
      time_mode(mode, n_wg, 17);
  time_mode(2, 2688, 9);
  return 0;
}
