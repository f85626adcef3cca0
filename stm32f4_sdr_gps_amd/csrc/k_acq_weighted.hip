// k_acq_weighted.hip -- EXTENSION, not in the reference: the acquisition grid on weighted two-bit samples.
//
// The reference correlates the MAX2769's sign bit only (PM/config.h:16) and every other kernel of this library computes what
// the reference computes.  This one uses both bits of a GPSX_IF_2BIT_SM capture (include/gpsx.h gpsx_acq_grid_weighted):
//   v[n]   = (sign ? +1 : -1) * (magnitude ? 3 : 1)                     (or +-1 in the sign-only mode, the comparison point)
//   vI, vQ = v with its sign flipped by the reference's carrier NCO (Fs/4 pattern of the accumulator's quadrant per 32-sample
//            word, phase 0 at the block's start, PM/GPS/gps_misc.c:211-240); the sixteen samples the NCO never mixes: weight 0
//   I(tau) = sum_n vI[n] c[((n - tau) mod 16368) / 16],  c = +1 / -1 for chip 0 / 1;  Q likewise
//   per (search, PRN, Doppler): max over the 16368 fine phases tau of floor(sqrt(I^2 + Q^2)), the first tau reaching it, the sum.
// Checked against its own CPU restatement (tests/test_gpu_weighted.py), itself pinned to this definition sample by sample.
//
// Formulation: with tau = 16 q + t0 the replica's chip c lies on samples 16 (q + c) + t0 .. + 15, so
//   I(16 q + t0) = sum_c c[c] S_t0[(q + c) mod 1023],   S_t0[k] = sum_{j < 16} vI[(16 k + t0 + j) mod 16368]   (|S| <= 48: int8)
// -- per sample offset t0 a circular correlation of the +-1 code with 1023 chip sums: v_dot4_i32_i8, four chips per
// instruction, integer multiply-accumulate on the vector ALU (no matrix cores: this is north_star's letter).
// A workgroup = one (search, Doppler bin, 8 PRNs): the capture's two bit planes -> LDS, wipe-off, the 2 x 16 x 1023 chip sums
// as int8 in LDS (doubled: q + c needs no wrap), then every thread owns four chip offsets q and walks the chips four at a time:
// per step 8 LDS dwords of sums (two streams x four q), 8 v_alignbyte, 8 wave-uniform chip words, 64 dot products.
#include "gpsx_device.hpp"
#include "gpsx_kernels.hpp"

namespace gpsx {

namespace {

constexpr int kWThreads = 256;
constexpr int kWG = 8;              // PRNs per workgroup
constexpr int kSumDwords = 512;     // one row of chip sums: 2046 int8 (1023 doubled) + 2 pad

struct WShared {
  u32 sign[512], mag[512];          // the block's bit planes (word 511: the last sixteen samples in its low half)
  u32 d[2][512];                    // wiped sign planes I / Q
  u32 sums[2][16][kSumDwords];      // [stream][t0][k]: int8 chip sums, k = 0 .. 2045
  u32 chips[kWG][256];              // per PRN: int8 x 4 chip signs, chips 4 i .. 4 i + 3 (chip 1023 = 0)
  u32 best[kWG], total[kWG];
};

// sixteen bits of a 16368-bit circular plane from bit p (p < 16368)
__device__ __forceinline__ u32 win16(const u32 *pl, int p)
{
  const u32 lo = __builtin_amdgcn_alignbit(pl[(p >> 5) + 1 < 512 ? (p >> 5) + 1 : 0], pl[p >> 5], (u32)(p & 31)) & 0xFFFFu;
  if (p <= kSamples - 16)
    return lo;
  const int n1 = kSamples - p;      // 1 .. 15 bits left before the wrap
  return ((lo & ((1u << n1) - 1u)) | (pl[0] << n1)) & 0xFFFFu;
}

__device__ __forceinline__ u32 isqrt_u64(u64 e)
{
  u64 r = (u64)__builtin_sqrt((double)e);
  r = r * r > e ? r - 1 : r;
  r = (r + 1) * (r + 1) <= e ? r + 1 : r;
  return (u32)r;
}

}  // namespace

__global__ __launch_bounds__(kWThreads) void k_acq_weighted(const uint8_t *__restrict__ if_blocks, int stride_blocks, int n_prn,
                                                            const uint8_t *__restrict__ chips_all, const uint8_t *__restrict__ prns,
                                                            int if_hz, int dopp_min_hz, int dopp_step_hz, int n_dopp,
                                                            int use_magnitude, gpsx_peak_t *__restrict__ peaks)
{
  extern __shared__ u32 w_smem[];
  WShared &sh = *reinterpret_cast<WShared *>(w_smem);
  const int tid = threadIdx.x;
  const int n_groups = (n_prn + kWG - 1) / kWG;
  const int group = (int)blockIdx.x % n_groups, dopp = ((int)blockIdx.x / n_groups) % n_dopp, search = (int)blockIdx.x / (n_groups * n_dopp);
  const uint8_t *blk = if_blocks + (size_t)search * stride_blocks * GPSX_BYTES_PER_MS_2BIT;

  // ---- the capture's two bit planes; the PRNs' chip signs ---------------------------------------------------------------
  for (int w = tid; w < 512; w += kWThreads) {
    u32 s = 0, m = 0;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int w16 = 2 * w + h;
      if (w16 < kWords16) {
        const uint16_t *p = reinterpret_cast<const uint16_t *>(blk) + 2 * w16;
        const u32 pairs = (u32)p[0] | ((u32)p[1] << 16);
        s |= even_bits16(pairs) << (16 * h);
        m |= even_bits16(pairs >> 1) << (16 * h);
      }
    }
    sh.sign[w] = s;
    sh.mag[w] = use_magnitude ? m : 0u;
  }
  for (int i = tid; i < kWG * 256; i += kWThreads) {
    const int g = i >> 8, c4 = i & 255, p = group * kWG + g;
    u32 word = 0;
    if (p < n_prn) {
      const uint8_t *ch = chips_all + (size_t)prns[p] * 1024;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int c = 4 * c4 + k;
        const u32 v = c < kChips ? (ch[c] ? 0xFFu : 0x01u) : 0u;   // -1 / +1 as int8; chip 1023 does not exist
        word |= v << (8 * k);
      }
    }
    sh.chips[g][c4] = word;
  }
  if (tid < kWG) {
    sh.best[tid] = 0;
    sh.total[tid] = 0;
  }
  __syncthreads();
  // ---- wipe-off: gps_shift_to_zero_freq on the sign plane (511 words; the last sixteen samples are never mixed) ----------
  const u32 step_word = nco_step_per_word((float)(if_hz + dopp_min_hz + dopp * dopp_step_hz));
  for (int w = tid; w < 512; w += kWThreads) {
    const u32 quad = (step_word * (u32)w) >> 30;
    sh.d[0][w] = w < kWords32 ? sh.sign[w] ^ carrier_i(quad) : 0u;
    sh.d[1][w] = w < kWords32 ? sh.sign[w] ^ carrier_q(quad) : 0u;
  }
  __syncthreads();
  // ---- chip sums: S = sum of the window's sixteen values = (2 pop(X & V) - pop(V)) + 2 (2 pop(X & M & V) - pop(M & V)),
  //      V = the samples the NCO mixed (bit positions below 16352) ---------------------------------------------------------
  for (int i = tid; i < 2 * 16 * kChips; i += kWThreads) {
    const int stream = i / (16 * kChips), rest = i % (16 * kChips), t0 = rest / kChips, k = rest % kChips;
    const int p = 16 * k + t0;                              // < 16368
    const u32 x = win16(sh.d[stream], p), m = win16(sh.mag, p);
    // valid bits of the window: positions (p + j) mod 16368 < 16352
    u32 v = 0xFFFFu;
    if (p + 16 > kSamples - 16) {
      v = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) {
        int n = p + j;
        n = n >= kSamples ? n - kSamples : n;
        v |= (n < kSamples - 16 ? 1u : 0u) << j;
      }
    }
    const int s = (2 * (int)__popc(x & v) - (int)__popc(v)) + 2 * (2 * (int)__popc(x & m & v) - (int)__popc(m & v));
    uint8_t *row = reinterpret_cast<uint8_t *>(sh.sums[stream][t0]);
    row[k] = (uint8_t)(int8_t)s;
    row[k + kChips] = (uint8_t)(int8_t)s;
  }
  for (int i = tid; i < 2 * 16; i += kWThreads)              // the two pad bytes of every row
    reinterpret_cast<uint8_t *>(sh.sums[i >> 4][i & 15])[2 * kChips] = reinterpret_cast<uint8_t *>(sh.sums[i >> 4][i & 15])[2 * kChips + 1] = 0;
  __syncthreads();

  // ---- the correlations: thread tid owns chip offsets q = tid + 256 j -----------------------------------------------------
  u32 best[kWG], total[kWG];
#pragma unroll
  for (int g = 0; g < kWG; g++) {
    best[g] = 0;
    total[g] = 0;
  }
#pragma unroll 1
  for (int t0 = 0; t0 < 16; t0++) {
    int acc[4][2][kWG];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int st = 0; st < 2; st++)
#pragma unroll
        for (int g = 0; g < kWG; g++)
          acc[j][st][g] = 0;
    const u32 *row_i = sh.sums[0][t0], *row_q = sh.sums[1][t0];
    u32 prev[4][2];                 // the dword below the window's upper one, per q and stream
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int q = min(tid + 256 * j, kChips - 1);
      prev[j][0] = row_i[q >> 2];
      prev[j][1] = row_q[q >> 2];
    }
#pragma unroll 2
    for (int c4 = 0; c4 < 256; c4++) {
      u32 win[4][2];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int q = min(tid + 256 * j, kChips - 1);
        const u32 nxt_i = row_i[(q >> 2) + c4 + 1], nxt_q = row_q[(q >> 2) + c4 + 1];
        win[j][0] = __builtin_amdgcn_alignbyte(nxt_i, prev[j][0], (u32)(q & 3));
        win[j][1] = __builtin_amdgcn_alignbyte(nxt_q, prev[j][1], (u32)(q & 3));
        prev[j][0] = nxt_i;
        prev[j][1] = nxt_q;
      }
#pragma unroll
      for (int g = 0; g < kWG; g++) {
        const u32 cw = sh.chips[g][c4];   // (wave-uniform address: one broadcast read)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          acc[j][0][g] = __builtin_amdgcn_sdot4((int)win[j][0], (int)cw, acc[j][0][g], false);
          acc[j][1][g] = __builtin_amdgcn_sdot4((int)win[j][1], (int)cw, acc[j][1][g], false);
        }
      }
    }
    // ---- this offset's magnitudes into the PRNs' running best / sum (tau = 16 q + t0; key = magnitude << 14 | 16383 - tau) --
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int q = tid + 256 * j;
      if (q >= kChips)
        continue;
      const u32 low = 16383u - (u32)(16 * q + t0);
#pragma unroll
      for (int g = 0; g < kWG; g++) {
        const long long ai = acc[j][0][g], aq = acc[j][1][g];
        const u32 m = isqrt_u64((u64)(ai * ai) + (u64)(aq * aq));
        const u32 key = (m << 14) | low;
        best[g] = key > best[g] ? key : best[g];
        total[g] += m;
      }
    }
  }
#pragma unroll
  for (int g = 0; g < kWG; g++) {
    const u32 b = wave_max_to_lane63(best[g]), t = wave_sum_to_lane63(total[g]);
    if ((tid & 63) == 63) {
      atomicMax(&sh.best[g], b);
      atomicAdd(&sh.total[g], t);
    }
  }
  __syncthreads();
  if (tid < kWG && group * kWG + tid < n_prn) {
    const u32 key = sh.best[tid], sum = sh.total[tid];
    gpsx_peak_t pk;
    pk.max_val = key >> 14;
    pk.phase = pk.max_val ? 16383u - (key & 0x3FFFu) : 0u;
    pk.sum = sum;
    pk.avr = sum / (u32)kSamples;
    peaks[((size_t)search * n_prn + group * kWG + tid) * n_dopp + dopp] = pk;
  }
}

int launch_acq_weighted(hipStream_t s, const uint8_t *d_if_blocks, int n_search, int stride_blocks, int n_prn,
                        const uint8_t *d_chips_all, const uint8_t *d_prns, int if_hz, int dopp_min_hz, int dopp_step_hz, int n_dopp,
                        int use_magnitude, gpsx_peak_t *d_peaks)
{
  // the attribute belongs to the CURRENT device's copy of the function: set before every launch (a table write in the runtime, no
  // device work) -- a process-wide "done" flag would leave the second device of a group without it, and would be a data race
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_acq_weighted), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)sizeof(WShared)) != hipSuccess)
    return -1;
  const int n_groups = (n_prn + kWG - 1) / kWG;
  hipLaunchKernelGGL(k_acq_weighted, dim3((unsigned)(n_search * n_dopp * n_groups)), dim3(kWThreads), sizeof(WShared), s, d_if_blocks,
                     stride_blocks, n_prn, d_chips_all, d_prns, if_hz, dopp_min_hz, dopp_step_hz, n_dopp, use_magnitude, d_peaks);
  return 0;
}

}  // namespace gpsx
