// k_codes.hip -- K1: C/A Gold code generation on the device and the tables derived from it.
//
// Replaces gps_generate_prn / gps_channell_prepare (PM/GPS/gps_misc.c:306-372): G1 = 1 + x^3 + x^10,
// G2 = 1 + x^2 + x^3 + x^6 + x^8 + x^9 + x^10, both registers all ones at the epoch, chip i = G1[i] ^ G2[i - delay].
// Each code-table slot is produced by one thread (1023 serial LFSR steps; this runs once per PRN list).
#include "gpsx_device.hpp"
#include "gpsx_kernels.hpp"

namespace gpsx {

__global__ void k_build_codes(const uint8_t *__restrict__ prns, int n_slots, int group, uint8_t *__restrict__ chips,
                              u32 *__restrict__ chipbits, u32 *__restrict__ cw, u32 *__restrict__ cw8)
{
  const int slot = blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= n_slots)
    return;
  const int prn = prns[slot];
  u32 bits[32];
#pragma unroll
  for (int i = 0; i < 32; i++)
    bits[i] = 0;

  if (prn >= 1 && prn <= GPSX_MAX_PRN) {
    // G2 output sequence first (packed), then G1 combined with the delayed G2
    u32 g2seq[32];
#pragma unroll
    for (int i = 0; i < 32; i++)
      g2seq[i] = 0;
    u32 g1 = 0x3FFu, g2 = 0x3FFu;  // bit k-1 = stage k
    for (int i = 0; i < kChips; i++) {
      g2seq[i >> 5] |= ((g2 >> 9) & 1u) << (i & 31);
      const u32 f2 = ((g2 >> 1) ^ (g2 >> 2) ^ (g2 >> 5) ^ (g2 >> 7) ^ (g2 >> 8) ^ (g2 >> 9)) & 1u;
      g2 = ((g2 << 1) | f2) & 0x3FFu;
    }
    const int delay = kG2Delay[prn - 1];
    int j = kChips - delay;  // index of the G2 chip aligned with G1 chip 0
    for (int i = 0; i < kChips; i++) {
      const u32 c = ((g1 >> 9) & 1u) ^ ((g2seq[j >> 5] >> (j & 31)) & 1u);
      bits[i >> 5] |= c << (i & 31);
      const u32 f1 = ((g1 >> 2) ^ (g1 >> 9)) & 1u;
      g1 = ((g1 << 1) | f1) & 0x3FFu;
      j = j + 1 == kChips ? 0 : j + 1;
    }
  }

  for (int i = 0; i < 32; i++)
    chipbits[(size_t)slot * 32 + i] = bits[i];
  uint8_t *row = chips + (size_t)slot * 1024;
  for (int i = 0; i < 1024; i++)
    row[i] = i < kChips ? (uint8_t)((bits[i >> 5] >> (i & 31)) & 1u) : (uint8_t)0;
  if (cw) {
    const bool live = prn >= 1 && prn <= GPSX_MAX_PRN;
    u32 *base = cw + (size_t)(slot / group) * kCodeWords * group + (slot % group);
    for (int jw = 0; jw < kCodeWords; jw++) {
      u32 word = 0;
      for (int e = 0; e < 4; e++) {
        const int c = 4 * jw + e;
        if (live && c < kChips)
          word |= (((bits[c >> 5] >> (c & 31)) & 1u) ? 17u : 1u) << (8 * e);
      }
      base[(size_t)jw * group] = word;
    }
  }
  if (cw8) {
    const bool live = prn >= 1 && prn <= GPSX_MAX_PRN;
    u32 *base = cw8 + (size_t)(slot / group) * (kCodeWords / 2) * group + (slot % group);
    for (int jw = 0; jw < kCodeWords / 2; jw++) {
      u32 word = 0;
      for (int e = 0; e < 8; e++) {
        const int c = 8 * jw + e;
        if (live && c < kChips)
          word |= ((bits[c >> 5] >> (c & 31)) & 1u) << (4 * e);
      }
      base[(size_t)jw * group] = word;
    }
  }
}

void launch_build_codes(hipStream_t s, const uint8_t *d_prns, int n_slots, int group, uint8_t *d_chips,
                        uint32_t *d_chipbits, uint32_t *d_cw, uint32_t *d_cw8)
{
  if (n_slots <= 0)
    return;
  hipLaunchKernelGGL(k_build_codes, dim3((n_slots + 63) / 64), dim3(64), 0, s, d_prns, n_slots, group, d_chips,
                     d_chipbits, d_cw, d_cw8);
}

}  // namespace gpsx
