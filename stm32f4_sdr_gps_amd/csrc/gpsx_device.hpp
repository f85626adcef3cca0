// gpsx_device.hpp -- shared device-side definitions for the gfx950 correlator kernels.
//
// Signal geometry and arithmetic follow the reference firmware (PM = Firmware/project_main):
//   PM/config.h:23-28   fs 16.368 MHz, IF 4.092 MHz, 16368 one-bit samples (2046 bytes) per C/A code period
//   PM/config.h:50      NCO resolution constant 0.003810972f (Hz per accumulator LSB, float32)
//   PM/GPS/gps_misc.c   the nine correlator primitives whose results these kernels reproduce bit for bit
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpsx.h"

namespace gpsx {

constexpr int kChips = 1023;          // C/A code length
constexpr int kSamples = 16368;       // samples per code period
constexpr int kBytes = 2046;          // packed bytes per code period
constexpr int kWords16 = 1023;        // 16-bit words per code period (one chip each at zero shift)
constexpr int kWords32 = 511;         // whole 32-sample words the carrier NCO mixes (the last 16 samples are not)
constexpr int kHalf = kSamples / 2;   // 8184: popcount centre

typedef unsigned int u32;
typedef unsigned long long u64;

// Fs/4 square wave, 32 samples per word, indexed by the NCO's top two bits (its quadrant).  The 7-nibble literal
// 0x09999999 in quadrant 0 (in-phase) / 1 (quadrature) is the reference's, PM/GPS/gps_misc.c:216-217.
__device__ __forceinline__ u32 carrier_i(u32 quadrant)
{
  const u32 t[4] = {0x09999999u, 0xCCCCCCCCu, 0x66666666u, 0x33333333u};
  return t[quadrant & 3u];
}
__device__ __forceinline__ u32 carrier_q(u32 quadrant)
{
  const u32 t[4] = {0x33333333u, 0x09999999u, 0xCCCCCCCCu, 0x66666666u};
  return t[quadrant & 3u];
}

// NCO accumulator increment per SAMPLE: (uint32)(freq_hz / 0.003810972f) evaluated in IEEE binary32
// (PM/GPS/gps_misc.c:219,250).  The library is compiled with -fhip-fp32-correctly-rounded-divide-sqrt.
__device__ __forceinline__ u32 nco_step_per_sample(float freq_hz)
{
  const float q = __fdiv_rn(freq_hz, 0.003810972f);
  return (u32)q;
}
// ... per 32-sample word (the NCO is only advanced once per word, PM/GPS/gps_misc.c:220-221,234)
__device__ __forceinline__ u32 nco_step_per_word(float freq_hz)
{
  return (u32)((u64)nco_step_per_sample(freq_hz) * 32ull);
}

// gps_correlation8's magnitude (PM/GPS/gps_misc.c:106-118): centre, one-sided clip, sqrtf of float32 squares,
// truncation.  sqrtf must be correctly rounded for the truncation to agree with the host libm: __builtin_sqrtf under
// -fhip-fp32-correctly-rounded-divide-sqrt expands to v_sqrt_f32 plus the two-FMA fix-up; HIP's __fsqrt_rn does NOT
// (it lowers to the bare 1-ulp v_sqrt_f32 on gfx950, checked in the ISA), so it must not be used here.
__device__ __forceinline__ int mag8(int cnt_i, int cnt_q)
{
  int i = cnt_i - kHalf;
  int q = cnt_q - kHalf;
  i = i < 0 ? 0 : i;
  q = q < 0 ? 0 : q;
  const float e = (float)(i * i) + (float)(q * q);
  return (int)__builtin_sqrtf(e);
}

// The same value for the hot epilogue of the grid kernel, with the instruction count trimmed:
//  * i, q <= 8184 < 2^24, so the squares are one full-rate v_mul_u32_u24 each instead of the quarter-rate v_mul_lo_u32;
//  * e is 0 or >= 1 and finite, so the denormal pre-scaling and the inf/zero class check of the generic sqrtf expansion
//    are dead weight; what remains is v_sqrt_f32 (<= 1 ulp) and the standard one-step correction that makes it the
//    correctly rounded root: with r- / r+ the floats next to r, pick r- if e - r-*r <= 0, r+ if e - r+*r > 0 (fused
//    residuals are exact enough to decide, this is the fix-up LLVM itself emits for a correctly rounded f32 sqrt).
//    e == 0: r = 0, r- is a NaN pattern, both tests fail, r stays 0.
template <bool SHORTCUT = true>
__device__ __forceinline__ int mag8_fast(int cnt_i, int cnt_q)
{
  int i = cnt_i - kHalf;
  int q = cnt_q - kHalf;
  i = i < 0 ? 0 : i;
  q = q < 0 ? 0 : q;
  const u32 ii = __umul24((u32)i, (u32)i), qq = __umul24((u32)q, (u32)q);
  // Whole wave below radius 1024 (the usual case: noise hypotheses sit at a few hundred): e = ii + qq < 2^20 is exact in
  // float, sqrtf(e) truncates to isqrt(e), and isqrt(e) = trunc(s) for ANY s within 1 ulp of sqrt(e + 1/2): the true
  // root of n^2 + r + 1/2 (0 <= r <= 2 n) keeps 1 / (4 (n + 1)) >= 2.4e-4 away from both integers around it, one ulp
  // below 1024 is 1.2e-4 at most.  So the bare v_sqrt_f32 does, with no fix-up.  (gpsx_mag8 checks both paths against
  // the generic one; tests/test_gpu_parity.py sweeps every pair of this domain.)
  if (SHORTCUT && __builtin_amdgcn_ballot_w64((ii + qq) >= (1u << 20)) == 0)
    return (int)__builtin_amdgcn_sqrtf((float)(ii + qq) + 0.5f);
  const float e = (float)ii + (float)qq;
  float r = __builtin_amdgcn_sqrtf(e);
  const float r_dn = __uint_as_float(__float_as_uint(r) - 1u);
  const float r_up = __uint_as_float(__float_as_uint(r) + 1u);
  const float res_dn = __builtin_fmaf(-r_dn, r, e);
  const float res_up = __builtin_fmaf(-r_up, r, e);
  r = res_dn <= 0.0f ? r_dn : r;
  r = res_up > 0.0f ? r_up : r;
  return (int)r;
}

// 16 even-position bits of a word gathered into the low half (bit 2k -> bit k): the sign plane of 16 two-bit samples
__device__ __forceinline__ u32 even_bits16(u32 x)
{
  x &= 0x55555555u;
  x = (x | (x >> 1)) & 0x33333333u;
  x = (x | (x >> 2)) & 0x0F0F0F0Fu;
  x = (x | (x >> 4)) & 0x00FF00FFu;
  x = (x | (x >> 8)) & 0x0000FFFFu;
  return x;
}

// One 16-bit word (16 samples) of the sign plane of IF block `blk`: read straight from a 1-bit block, or unpacked from
// sixteen sign/magnitude pairs of a 2-bit block.  `blk` is 2-byte aligned in either format.
__device__ __forceinline__ uint16_t load_sign16(const uint8_t *blk, int word, int if_format)
{
  if (if_format == GPSX_IF_2BIT_SM) {
    const uint16_t *p = reinterpret_cast<const uint16_t *>(blk) + 2 * word;
    return (uint16_t)even_bits16((u32)p[0] | ((u32)p[1] << 16));
  }
  return reinterpret_cast<const uint16_t *>(blk)[word];
}

__device__ __forceinline__ u32 pop16(u32 v) { return (u32)__popc(v & 0xFFFFu); }

// wave64 all-lanes reductions by butterfly shuffles
__device__ __forceinline__ u32 wave_max_u32(u32 v)
{
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const u32 o = (u32)__shfl_xor((int)v, m, 64);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ u32 wave_sum_u32(u32 v)
{
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1)
    v += (u32)__shfl_xor((int)v, m, 64);
  return v;
}

// wave64 reductions on the DPP network (6 VALU ops each); the result is valid in lane 63
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ u32 dpp(u32 old, u32 v)
{
  return (u32)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ u32 wave_max_to_lane63(u32 v)
{
  u32 o;
  o = dpp<0xB1>(v, v); v = o > v ? o : v;            // quad_perm [1,0,3,2]
  o = dpp<0x4E>(v, v); v = o > v ? o : v;            // quad_perm [2,3,0,1]
  o = dpp<0x141>(v, v); v = o > v ? o : v;           // row_half_mirror
  o = dpp<0x140>(v, v); v = o > v ? o : v;           // row_mirror: every lane holds its row's maximum
  o = dpp<0x142, 0xA>(v, v); v = o > v ? o : v;      // row_bcast15 into rows 1, 3
  o = dpp<0x143, 0xC>(v, v); v = o > v ? o : v;      // row_bcast31 into rows 2, 3
  return v;
}
__device__ __forceinline__ u32 wave_sum_to_lane63(u32 v)
{
  v += dpp<0xB1>(0u, v);
  v += dpp<0x4E>(0u, v);
  v += dpp<0x141>(0u, v);
  v += dpp<0x140>(0u, v);
  v += dpp<0x142, 0xA>(0u, v);   // rows 1, 3 += lane 15 of the row before (masked rows read the old value, 0)
  v += dpp<0x143, 0xC>(0u, v);   // rows 2, 3 += lane 31
  return v;
}

// IS-GPS-200 G2 register delay in chips for PRN 1..210 (PM/GPS/gps_misc.c:319-341 uses the same assignment,
// including the PRN 34 / PRN 37 duplicate).
__device__ const uint16_t kG2Delay[210] = {
  5, 6, 7, 8, 17, 18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258, 469, 470, 471, 472, 473, 474, 509, 512, 513, 514,
  515, 516, 859, 860, 861, 862, 863, 950, 947, 948, 950, 67, 103, 91, 19, 679, 225, 625, 946, 638, 161, 1001, 554, 280,
  710, 709, 775, 864, 558, 220, 397, 55, 898, 759, 367, 299, 1018, 729, 695, 780, 801, 788, 732, 34, 320, 327, 389, 407,
  525, 405, 221, 761, 260, 326, 955, 653, 699, 422, 188, 438, 959, 539, 879, 677, 586, 153, 792, 814, 446, 264, 1015, 278,
  536, 819, 156, 957, 159, 712, 885, 461, 248, 713, 126, 807, 279, 122, 197, 693, 632, 771, 467, 647, 203, 145, 175, 52,
  21, 237, 235, 886, 657, 634, 762, 355, 1012, 176, 603, 130, 359, 595, 68, 386, 797, 456, 499, 883, 307, 127, 211, 121,
  118, 163, 628, 853, 484, 289, 811, 202, 1021, 463, 568, 904, 670, 230, 911, 684, 309, 644, 932, 12, 314, 891, 212, 185,
  675, 503, 150, 395, 345, 846, 798, 992, 357, 995, 877, 112, 144, 476, 193, 109, 445, 291, 87, 399, 292, 901, 339, 208,
  711, 189, 263, 537, 663, 942, 173, 900, 30, 500, 935, 556, 373, 85, 652, 310
};

}  // namespace gpsx
